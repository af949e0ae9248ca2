/* mst_amd.h — C ABI of libmst_amd.so, the MI355X (gfx950) hot path of music-style-transfer.
 *
 * The reference (marcinp7/music-style-transfer) is pure Python and has no FFI; the boundary
 * this library sits behind is the Python surface of style/model.py.  Each entry point names
 * the reference interface it replaces (paths relative to the reference root).  A maintainer
 * binds these with ctypes; no torch types cross the boundary — only raw
 * device pointers, sizes and a hipStream_t.
 *
 * Conventions: every function returns 0 on success and a negative mst_status otherwise and
 * never throws; all work is enqueued on `stream` (no host synchronisation, no allocation in
 * the launch path — graph-capturable); all tensors are fp32, contiguous, device-resident.
 */
#ifndef MST_AMD_H
#define MST_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MST_OK = 0,
    MST_ERR_ARG = -1,          /* null pointer / bad size */
    MST_ERR_UNSUPPORTED = -2,  /* layer width outside the instantiated kernels */
    MST_ERR_LAUNCH = -3,       /* hipGetLastError() after a launch */
    MST_ERR_ALLOC = -4,
} mst_status;

/* Layer widths = the constructor arguments at train-model.py:54-60 + style/data.py:19-31,
 * clip shape = (channel, bar, beat) of prepare_input's tensors (style/data.py:130-156). */
typedef struct {
    int32_t C, R, T;            /* pitched channels, bars, beats per bar (batch is always 1) */
    int32_t beat, bar, nrf;     /* beat_size, bar_size, n_rhythm_features */
    int32_t style, melody, rhythm;
    int32_t instr;              /* instrument_size (51) */
    int32_t n_instruments;      /* 41 */
    int32_t has_unpitched;      /* unpitched_channels is not None */
    int32_t clips;              /* independent clips of this shape carried by every launch (0 or 1 = one clip).
                                 * The reference is strictly batch 1 and sums gradients over iter_size songs
                                 * (train-model.py:95,126,151-153); K clips here = K such iterations at the same
                                 * parameters: per-clip combine / LSTM chains / loss tree, parameter gradients summed. */
} mst_dims;

enum { MST_STAGE_EXTRACT = 1, MST_STAGE_INFO = 2, MST_STAGE_APPLY = 4, MST_STAGE_ALL = 7 };

/* Loss leaves, in the key order of get_total_loss's nested dict flattened with '_'
 * (style/model.py:944-996, train-model.py:148). Unpitched leaves are NaN when absent. */
enum {
    MST_L_TOTAL = 0,
    MST_L_CH_TOTAL, MST_L_P_TOTAL, MST_L_P_NOTES, MST_L_P_VELOCITY, MST_L_P_DURATION, MST_L_P_ACCIDENTALS,
    MST_L_U_TOTAL, MST_L_U_NOTES, MST_L_U_VELOCITY, MST_L_U_DURATION,
    MST_L_SI_TOTAL, MST_L_SI_INSTRUMENTS, MST_L_SI_MODE, MST_L_SI_BPM,
    MST_N_LOSSES
};

typedef struct mst_plan mst_plan;
typedef void* mst_stream;       /* hipStream_t */

/* ---- parameter layout: replaces model.parameters()/state_dict() ordering
 * (style/model.py:36-75,102-126,144-177,203-250,301-344,384-416,446-511,582-622,678-701,727-749).
 * All parameters live in ONE flat fp32 buffer in model.parameters() order. */
int32_t mst_param_count(const mst_dims* d);
int64_t mst_param_floats(const mst_dims* d);
/* i-th tensor: state_dict name, flat offset, shape (up to 3 dims). */
int32_t mst_param_info(const mst_dims* d, int32_t i, char* name, int32_t name_cap,
                       int64_t* offset, int32_t* ndim, int32_t shape[3]);

/* MST_OK when the HIP kernels are instantiated for the layer widths in `d` (the note-level kernels exist for
 * melody_size 8 and 4 with the reference's derived widths), MST_ERR_UNSUPPORTED otherwise: lets the Python
 * constructors fail early instead of at the first forward. */
int32_t mst_widths_supported(const mst_dims* d);

/* ---- plan: static launch schedule + device-side descriptors for one mst_dims. */
typedef struct {
    int32_t gemm_tile;          /* 0 = choose from the clip count (64x64 tiles from 6 clips per launch on, else 32x32
                                 * split-K tiles); 32 / 64 = force that tiling (experiments, parity tests) */
    int32_t no_merge;           /* 1 = one launch per scheduled member (profiling aid); 0 = merge a dependency level's
                                 * launches of one kernel */
    int32_t gemm_run;           /* 64x64 tiling: output tiles one workgroup walks back to back; 0 = choose (1..8, keeping >= ~4096
                                 * workgroups per launch) */
    int32_t tile_r0, tile_rows; /* bar tiling of ONE clip over several ranks (BASELINE.json configs[4]): this plan computes
                                 * the per-position work of bars [tile_r0, tile_r0 + tile_rows) of the mst_dims.R bars; the bar-level
                                 * chains run replicated.  tile_rows = 0: not tiled.  Run with mst_tiled_phase (clips must be 1);
                                 * `pitched` / `unpitched` are then the tile's bars only: (1,C,tile_rows,T,10,56,5), (1,1,tile_rows,T,10,47,2) */
    int32_t lstm_flavour;       /* StyleEncoder.bars_lstm (H = 192, style/model.py:152,179-181): 0 = choose (the 12-workgroup-per-clip
                                 * kernels when all of a launch's workgroups fit the device at once with one workgroup slot per CU to spare,
                                 * else one workgroup per sequence); 1 = always one workgroup per sequence (use it when other kernels are
                                 * meant to run beside a batched plan); 2 = the 12-workgroup kernels with an injected exchange fault
                                 * (tests of the mst_plan_status path only) */
    int32_t dense_flavour;      /* large dense nn.Linear layers of plans on the 64x64 tiling (lin.hip: all clips as rows of one launch, 2 x 2-blocked
                                 * MFMA tiles): 0 = choose (layers of >= 512 rows and >= 4 MFLOP per clip with more than 32 outputs), 1 = never, 2 = every eligible layer (parity tests
                                 * at small sizes) */
    int32_t branches;           /* 1 = spread the launches of the whole-model passes over the caller's stream + 3 side streams along the dependency
                                 * DAG (event waits where a dependency crosses streams); 0 = one stream (default).  An experiment that LOST on
                                 * MI355X (ROCm 7.2), kept for the record and for its parity test: one bench clip, eager launches 0.85 ms per
                                 * step against 0.49; a replayed hipGraph 875 us per iteration against 812 (a graph with parallel branches pays
                                 * ~5 us per node boundary and per cross edge, a single chain ~2 us per boundary: tools/probe/capture_probe.cpp);
                                 * torch.cuda.graph's capture_end crashes on such captures, so the Python binding refuses them. */
} mst_plan_options;
mst_plan* mst_plan_create(const mst_dims* d, int32_t* status);                       /* default options */
mst_plan* mst_plan_create_ex(const mst_dims* d, const mst_plan_options* opt, int32_t* status);
int32_t mst_plan_gemm_tile(const mst_plan* p);                                       /* 32 or 64 */
void mst_plan_destroy(mst_plan* p);
int64_t mst_plan_workspace_floats(const mst_plan* p);
/* Named tensor inside the workspace: activation offset, gradient offset, element count.
 * Names: "instr","mode","bpm","style","melody","rhythm","instruments_pred","mode_pred",
 * "bpm_pred","pitched_pred","unpitched_pred","pitched_beats","pitched_bars",... */
int32_t mst_plan_tensor(const mst_plan* p, const char* name, int64_t* off, int64_t* goff, int64_t* numel);
int32_t mst_plan_launch_count(const mst_plan* p, int32_t stage_mask, int32_t backward);
/* out = {clips, activation floats per clip, scratch floats per clip, offset of the gradient arena}.
 * Clip k's copy of a named tensor sits k * out[1] floats after clip 0's (mst_plan_tensor offsets). */
int32_t mst_plan_layout(const mst_plan* p, int64_t out[4]);

/* ---- device health.  Kernels whose workgroups wait for each other (the 12-workgroup StyleEncoder LSTM) cannot report a
 * failure through a return code: the launch has long returned.  They OR a bit into a status word inside the workspace
 * (named tensor "device_status", first element, int32) and produce NaNs from then on; the word is sticky.  mst_plan_status
 * synchronises `stream`, copies the word to *status and, with clear != 0, resets it.  A training loop that reads its losses back
 * every N iterations checks it at the same moment (style/train.py LossLog.flush).  No reference counterpart: the reference has
 * no device-side synchronisation of its own. */
enum { MST_DEV_OK = 0, MST_DEV_LSTM_TIMEOUT = 1 /* a granule of the LSTM exchange did not arrive within 0.2 s */ };
int32_t mst_plan_status(const mst_plan* p, float* ws, int32_t clear, int32_t* status, mst_stream stream);

/* ---- forward: StyleTransferModel.extract_style / predict_song_info / apply_style / forward
 * (style/model.py:751-793), selected by stage_mask. `pitched` (1,C,R,T,10,56,5) and
 * `unpitched` (1,1,R,T,10,47,2) are borrowed for the call; small inputs and stage-boundary
 * tensors are read from / written to their workspace slots (mst_plan_tensor). */
int32_t mst_forward(const mst_plan* p, int32_t stage_mask, const float* params, float* ws,
                    const float* pitched, const float* unpitched, mst_stream stream);

/* ---- backward of the same stages (what loss.backward() does, train-model.py:126).
 * Reads output gradients from the workspace gradient slots, accumulates parameter gradients
 * into gparams (+=, sum semantics — train-model.py:126,151-153).  Call mst_zero_grads(stage_mask) before it;
 * when a stage runs on its own the caller also writes (or clears) the gradient slots of that stage's outputs and
 * of the stage-boundary tensors "style", "melody", "rhythm". */
int32_t mst_backward(const mst_plan* p, int32_t stage_mask, const float* params, float* gparams,
                     float* ws, const float* pitched, const float* unpitched, mst_stream stream);
int32_t mst_zero_grads(const mst_plan* p, int32_t stage_mask, float* ws, mst_stream stream);
/* Gradient floats per clip that mst_zero_grads(stage_mask) clears: only ranges the plan's first-writer analysis could not
 * prove written before they are read or accumulated into (every other gradient slot's first writer stores). */
int64_t mst_plan_zero_floats(const mst_plan* p, int32_t stage_mask);

/* ---- get_total_loss (style/model.py:935-997) with normalize flag; pointer based so that it
 * also serves the stand-alone Python get_total_loss. n_*_pos = number of note positions
 * (product of all dims but the last). losses: MST_N_LOSSES floats. saved: MST_LOSS_SAVED floats
 * carried to the backward call. partials: scratch of mst_loss_scratch_floats(). */
#define MST_LOSS_SAVED 512
int64_t mst_loss_scratch_floats(void);
int32_t mst_total_loss_fwd(const float* pitched_pred, const float* pitched_target, int64_t n_pitched_pos,
                           const float* unpitched_pred, const float* unpitched_target, int64_t n_unpitched_pos,
                           const float* instr_logits, const float* instr_target, int32_t n_instr,
                           const float* mode_logits, const float* mode_target,
                           const float* bpm_pred, const float* bpm_target,
                           int32_t normalize, float* losses, float* saved, float* scratch, mst_stream stream);
/* grad_losses: MST_N_LOSSES upstream gradients (one-hot on MST_L_TOTAL for loss.backward()).
 * Writes (=) the gradients of every prediction. */
int32_t mst_total_loss_bwd(const float* pitched_pred, const float* pitched_target, int64_t n_pitched_pos,
                           const float* unpitched_pred, const float* unpitched_target, int64_t n_unpitched_pos,
                           const float* instr_logits, const float* instr_target, int32_t n_instr,
                           const float* mode_logits, const float* mode_target,
                           const float* bpm_pred, const float* bpm_target,
                           const float* saved, const float* grad_losses,
                           float* g_pitched, float* g_unpitched, float* g_instr, float* g_mode, float* g_bpm,
                           mst_stream stream);

/* ---- one train-model.py loop body (train-model.py:113-126): forward, total loss,
 * backward; gradients accumulate in gparams. Targets are the inputs (auto-encoder), the
 * instrument target is `used` (n_instruments), bpm target a device float. losses may be null.
 * With mst_dims.clips = K: `pitched` is (K,C,R,T,10,56,5), `unpitched` (K,1,R,T,10,47,2), `losses`
 * K x MST_N_LOSSES, the per-clip small inputs sit in each clip's workspace slice, and gparams
 * receives the SUM over the K clips (= K loop bodies at the same parameters). */
int32_t mst_train_iteration(const mst_plan* p, const float* params, float* gparams, float* ws,
                            const float* pitched, const float* unpitched, float* losses, mst_stream stream);

/* ---- one loop body (train-model.py:113-126) of a clip whose bars are tiled over ranks.  Every rank calls phase 0, 1, ...
 * mst_tiled_phase_count() - 1 on its own plan / workspace / gparams; a phase returns *nx <= MST_MAX_XCHG workspace ranges
 * ws[xoff[q], xoff[q] + xlen[q]) that the host all-reduces (SUM) over the ranks before the next phase — as ONE collective (pack the
 * ranges, reduce, unpack: Plan.tiled_train_iteration); exchanges of one dependency level end the same phase.  Afterwards gparams
 * holds this rank's share of the clip's gradient: all-reduce (SUM) it like in data parallelism (train-model.py:126,151-153), then
 * mst_adam_step.  is_root: exactly one rank passes 1 (it contributes the replicated song-info loss gradients).  losses:
 * MST_N_LOSSES floats, identical on every rank. */
#define MST_MAX_XCHG 8
int32_t mst_tiled_phase_count(const mst_plan* p);
int32_t mst_tiled_phase(const mst_plan* p, int32_t phase, const float* params, float* gparams, float* ws,
                        const float* pitched, const float* unpitched, float* losses, int32_t is_root,
                        mst_stream stream, int64_t xoff[MST_MAX_XCHG], int64_t xlen[MST_MAX_XCHG], int32_t* nx);

/* ---- torch.optim.Adam(lr=.01) + StepLR(200,.9) + zero_grad (train-model.py:89-90,151-154)
 * over the flat buffers. state: 4 floats {step count t, lr_t/(1-b1^t), sqrt(1-b2^t), reserved} kept on the device so that
 * the launch is graph-replayable; lr = lr0 * gamma^(t / step_size). */
int32_t mst_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                      float* state, double lr0, double beta1, double beta2, double eps,
                      int32_t step_size, double gamma, int32_t zero_grad, mst_stream stream);

/* Same step over the SUM of two gradient buffers: the iter_size = 2 accumulation iterations of
 * train-model.py:126,151-153 are independent, so they may run concurrently (two streams, two workspaces,
 * two gradient buffers) and meet here; grads + grads2 equals accumulating in place, bit for bit. */
int32_t mst_adam_step2(float* params, float* grads, float* grads2, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float* state, double lr0, double beta1, double beta2, double eps,
                       int32_t step_size, double gamma, int32_t zero_grad, mst_stream stream);

/* ---- hard_output (style/model.py:818-832): n_pos positions x nfeat (5 or 2) features.
 * Like the reference it also zeroes sub-threshold velocities of `x` in place. */
int32_t mst_hard_output(float* x, float* out, int64_t n_pos, int32_t nfeat, mst_stream stream);

/* ---- instrumentation (bench.py only; synchronises on HIP events, never used for training):
 * average duration of every launch step of a pass, with its algorithmic FLOPs and bytes.
 * kind: 0 gemm, 1 gather, 2 segment-reduce, 3/4 lstm fwd/bwd, 5/6 combine fwd/bwd, 7/8 melody notes, 9/10 applier notes,
 * 11 lstm weight transpose, 12/13 row-wise tiny Linear fwd/bwd, ..., 26-28 conv.hip prep / forward / weight gradient,
 * 29-31 lin.hip forward / input gradient / weight gradient. */
int32_t mst_plan_step_count(const mst_plan* p, int32_t stage_mask, int32_t backward);
int32_t mst_plan_step_info(const mst_plan* p, int32_t stage_mask, int32_t backward, int32_t* info /* 8 ints per step: 4 shape values, member count, kind, dependency level (inside its chain), chain (-1 = none) */);
/* (new) instrumentation: the members of GEMM launch step `step` of a pass, one clip's worth, 6 values each:
 * {M, N, K, k-splits, folded rows per clip (0 = not folded), workgroups}.  Returns the member count (<= cap). */
int32_t mst_plan_step_gemms(const mst_plan* p, int32_t stage_mask, int32_t backward, int32_t step, int32_t* out, int32_t cap);
int32_t mst_plan_time_steps(const mst_plan* p, int32_t stage_mask, int32_t backward, const float* params,
                            float* gparams, float* ws, const float* pitched, const float* unpitched,
                            mst_stream stream, int32_t reps, float* ms, int32_t* kind, double* flops, double* bytes);

/* ---- AUDIO EXTENSION — NOT REFERENCE PARITY.  The reference has no audio path (latex/music-style-transfer.tex:79-80 names it as
 * future work; requirements.txt:1-8 has no audio library); BASELINE.json's metric text nevertheless speaks of 30 s @ 44.1 kHz clips,
 * an STFT(1024/256) featuriser and a feature-Gram style loss, and SURVEY.md 8(f4) keeps that as an optional extension with a
 * build-defined oracle (oracle/audio_oracle.py: torch.stft / matmul / autograd on the CPU; parity unpinned).  Nothing of the hot
 * path above uses these entry points.
 * Layout: a clip of n_samples mono float samples -> frames = 1 + n_samples / hop frames (centre-padded by reflection, periodic Hann
 * window: torch.stft's defaults) x bins = n_fft / 2 + 1; magnitude / feature matrices are (frames x ld) row-major with
 * ld = bins rounded up to a multiple of 8 and zero pad columns; Gram matrices are (ld x ld), pads zero. */
typedef struct mst_audio_plan mst_audio_plan;
mst_audio_plan* mst_audio_plan_create(int32_t n_fft /* 1024 | 2048 */, int32_t hop, int64_t n_samples, int32_t* status);
void mst_audio_plan_destroy(mst_audio_plan* p);
/* out = {frames, bins, ld, workspace floats of mst_audio_gram / mst_audio_style_iteration, k-splits of the Gram, Gram tiles} */
int32_t mst_audio_plan_info(const mst_audio_plan* p, int64_t out[6]);
/* spec: frames x bins complex64 (re, im interleaved) or NULL; mag: frames x ld or NULL (at least one) */
int32_t mst_audio_stft(const mst_audio_plan* p, const float* audio, float* spec, float* mag, mst_stream stream);
/* gram = feat^T feat / frames (ld x ld) on the f32 matrix cores; ws: workspace (mst_audio_plan_info) */
int32_t mst_audio_gram(const mst_audio_plan* p, const float* feat, float* gram, float* ws, mst_stream stream);
/* One optimisation iteration on x (frames x ld): loss = || x^T x / frames - gram_style ||_F^2 (written to *loss, a device float),
 * grad = (4 / frames) x (x^T x / frames - gram_style) (written to `grad`, frames x ld), then Adam(lr, .9, .999, 1e-8) on x with
 * state tensors exp_avg / exp_avg_sq (frames x ld) and `state` (4 floats, as mst_adam_step). */
int32_t mst_audio_style_iteration(const mst_audio_plan* p, float* x, const float* gram_style, float* grad, float* exp_avg,
                                  float* exp_avg_sq, float* state, float* ws, float* loss, double lr, mst_stream stream);

/* test hook of the plan builder's single-launch weight-gradient reduction guard: do two column blocks of one row-major matrix
 * (row pitch ld), given by their first elements' offsets from the matrix's element (0, 0), share no element?  1 = disjoint. */
int32_t mst_debug_slab_columns_disjoint(int64_t off_x, int32_t width_x, int64_t off_y, int32_t width_y, int32_t ld);

const char* mst_version(void);

#ifdef __cplusplus
}
#endif
#endif
