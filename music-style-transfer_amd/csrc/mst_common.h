// Internal shared definitions: derived layer sizes, device-side descriptors, launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mst_amd.h"

// global-address-space qualifier for hot loads (so they are global_load, never flat_load);
// the hipsim interpreter pre-defines it as nothing
#ifndef MST_GLOBAL_AS
#define MST_GLOBAL_AS __attribute__((address_space(1)))
#endif
// constant-address-space qualifier for WAVE-UNIFORM read-only data (the tiny weight matrices of the note kernels): loads through
// such a pointer are scalar loads (s_load_dword*) into SGPRs, which every lane's FMA reads as an operand — held in vector
// registers the same weights cost ~170 VGPRs per lane and the second wave per SIMD.  The interpreter pre-defines it as nothing.
#ifndef MST_CONST_AS
#define MST_CONST_AS __attribute__((address_space(4)))
#endif
// a value that is the same in every lane of the wave, as a scalar (SGPR): branches on it are uniform branches
#ifndef MST_UNIFORM
#define MST_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
// LDS-only workgroup barrier: waits for this wave's LDS traffic, then s_barrier.  Unlike
// __syncthreads() it does not drain vmcnt, so global prefetches / streamed stores issued around
// it stay in flight (on MI355X the vmcnt(0) of __syncthreads() cost ~1-4 us per LSTM step).
// Only for data exchanged through LDS; the interpreter maps it to its ordinary barrier.
// hardware exponential (v_exp_f32, ~1 ulp) for the strictly sequential LSTM steps, where the
// libm-accurate expf/tanhf bodies sit on the critical path; the interpreter maps it to expf
#ifndef MST_FAST_EXP
#define MST_FAST_EXP(x) __expf(x)
// v_rcp_f32 (1 ulp); __fdividef expands to the full IEEE division sequence (div_scale, rcp, 4 fma, div_fmas, div_fixup)
#define MST_FAST_RCP(x) __builtin_amdgcn_rcpf(x)
// makes a loaded value live HERE: the compiler waits for the load at this point instead of at its first use — inside a loop
// that also stores, that wait (vmcnt counts stores too on gfx9) would drain the previous iteration's stores every iteration
#define MST_PIN(x) asm volatile("" : "+v"(x))
#endif
#ifndef MST_LDS_BARRIER
#define MST_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// Exchange through a WAVE-PRIVATE LDS region: the DS operations of one wave execute in issue order, so a wave's reads see
// its own lanes' earlier writes without an s_barrier; this only keeps the compiler from moving LDS accesses across the
// point (no instruction is emitted).  The interpreter maps it to a real wave barrier (its lanes are separate fibers).
// scheduling fence: keeps hipcc from hoisting a fully unrolled loop's LDS reads far ahead of their use (it did: 512 VGPRs
// and scratch spills in the applier's backward); no instruction is emitted
#ifndef MST_SCHED_FENCE
#define MST_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef MST_WAVE_SYNC
#define MST_WAVE_SYNC()                                            \
    do {                                                           \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
        __builtin_amdgcn_wave_barrier();                           \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
    } while (0)
#endif

// ---- piano-roll constants (style/model.py:13-25)
#define NF 10       // beat fractions
#define NPF 5       // pitched note features
#define NUF 2       // unpitched note features
#define NOCT 8
#define NDEG 7
#define NPN 56      // pitched notes
#define NUN 47      // unpitched notes
#define CONV_K 14
#define CONV_PAD 4
#define LEAKY 0.01f

// address spaces a descriptor offset can live in: activations, flat params, flat param grads,
// the two borrowed note tensors, the gradient arena (mirrors SP_WS offsets), scratch
enum { SP_WS = 0, SP_PAR = 1, SP_GPAR = 2, SP_EXT0 = 3, SP_EXT1 = 4, SP_GRAD = 5, SP_TMP = 6, SP_COUNT = 7 };
// flags: MST_BF_LOSS_FUSED — this backward pass belongs to mst_train_iteration: the gradient of the pitched prediction is not in
// its gradient slot; the applier's backward kernel derives it from (prediction, target, the loss tail's saved Jacobian) itself
enum { MST_BF_LOSS_FUSED = 1,
       MST_BF_ALL_STAGES = 2 };    // the pass walks the whole-model launch list (selects LinDesc.first[1])
struct Bases { float* p[SP_COUNT]; int32_t flags; };

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SIGOUT = 2, ACT_BPM = 3 };

// ---- generic GEMM: C[m,n] = epilogue(sum_k A(m,k) * B(k,n))
enum { OPK_DENSE = 0, OPK_ACTGRAD = 2, OPK_IM2COL = 3, OPK_PERMW = 4, OPK_CONVGRAD = 5 };
#define MAX_SEG 6
struct Seg {
    int32_t space, ld, start, width;
    int64_t off;
    int32_t s[4];        // source-row stride for each of the 4 row-space dims (0 = broadcast)
};
struct Operand {
    int32_t kind;
    int32_t space;       // DENSE / ACTGRAD(dY) / IM2COL(x) / CONVW / CONVGRAD(dY)
    int64_t off;
    int64_t si, sj;      // DENSE: val = base[off + i*si + j*sj], (i,j) = (m,k) for A, (k,n) for B
    int32_t ones_at;     // second index == ones_at -> 1.0f (bias-gradient column); -1 = none
    int32_t kfast;       // tile loader walks the k index fastest (coalescing hint)
    // ACTGRAD: val(row,j) = dY[row*ld+j] * act'(Y[row*ld+j]); transposed: first index is j
    int32_t space2, ld, act, transposed;
    int64_t off2;
    int32_t oc;          // conv: number of output channels
    int32_t pb, pc;      // PERMW: k = (a, b, c) with sizes (., pb, pc) reads weight column a*pb*pc + c*pb + b
};
enum { OUT_STORE = 0, OUT_ACCUM = 1, OUT_CONV = 2, OUT_SLAB = 3, OUT_PERMW_SLAB = 4 };
struct OutSpec {
    int32_t kind, space, ldc, act;
    int64_t off;
    int32_t bias_space;  // -1 = no bias
    int64_t bias_off;
    int64_t slab_stride; // OUT_SLAB: floats between k-splits
    int32_t first;       // OUT_ACCUM: 1 = first writer of its (dense) target in this pass: store instead of += (no memset needed)
    int32_t wcols;       // OUT_SLAB: columns n < wcols are weights (row-major m*wcols+n); n == wcols is the bias column
    int32_t pb, pc;      // OUT_PERMW_SLAB: same column permutation as OPK_PERMW
    int32_t bias_div, bias_ld;   // OUT_STORE with an activation bias: bias_div > 0 -> row m adds bias row m / bias_div (stride bias_ld)
};
struct GemmDesc {
    int32_t M, N, K, ksplit;
    int32_t variant;     // GV_* instantiation (filled by the plan from the operand kinds)
    int32_t blk_begin;   // first workgroup of this member inside its (merged) launch
    int32_t kdsel;       // k-tile depth: 0 -> 32, 1 -> 64, 2 -> 128
    int32_t run;         // 64x64 tiling: consecutive output tiles one workgroup walks (same k-split); 0 / 1 = one tile
    // Weight-gradient GEMMs of a batched plan fold the clips into the reduction: K = fold_rows * clips, reduction index
    // k = (clip, row) reads A at A.off + clip * acs (+ acs2 for the activation) + row ..., B at B.off + clip * bcs + row ...
    // One GEMM instead of one per clip: no k-tile padding at short sequences, clips x fewer slab rows to reduce.
    int32_t fold_rows;   // 0 = not folded
    int64_t acs, acs2, bcs;
    // Linears over ONE row per clip (style / song-info heads) of a batched plan: the clips are the rows of one GEMM — M = clips,
    // row stride = the clip's arena stride (set at schedule time) — so the weights are read once instead of once per clip.
    int32_t clip_rows;   // 1 = built for one clip with M = 1; scheduled with M = clips
    Operand A, B;
    OutSpec out;
};

// ---- gather: out[row, :] = concat_s seg_s[index_s(row), :]   (cat_with_broadcast, materialised once)
struct GatherDesc {
    int32_t rows, K, nseg;
    int32_t sum;         // 1: the segments (all K wide, start 0) are ADDED instead of concatenated (a broadcast sum)
    int32_t d[4];
    int64_t out_off;     // [SP_WS] rows x K contiguous
    Seg seg[MAX_SEG];
};

// ---- segment reduce: dX_seg[idx, w] += sum_{rows -> idx} dAcat[row, start + w]
struct SegRedDesc {
    int64_t src_off; int32_t src_ld; int32_t start, width;   // gradient of the gathered tensor, SP_GRAD
    int64_t dst_off; int32_t dst_ld;                           // gradient slot in SP_GRAD
    int32_t d[4];        // row-space dims
    int32_t s[4];        // destination row stride per dim (0 = reduced)
    int32_t nidx;        // number of distinct destination rows
    int32_t kd[4];       // kept-dim sizes (1 where reduced)
    int32_t nchunk;      // chunks of 64 reduced rows; > 1 => partials at part_off, summed by stage 2
    int64_t part_off;    // [SP_TMP] nidx * nchunk * width
    int32_t blk_begin;   // first workgroup of this member inside its clip's block range of the (merged) launch
    int32_t first;       // 1 = first writer of dst in this pass: store instead of +=
    int32_t act;         // != ACT_NONE: the summed values are src x act'(y), y = the activation at y_off [SP_WS], laid out like src
    int64_t y_off;
};

// ---- LSTM recurrence
struct LstmDesc {
    int32_t B, S, H, reverse;
    int64_t zx_off;      // (B*S, 4H) input projection incl. b_ih          [SP_WS]
    int64_t whh_off, bhh_off;                                             // [SP_PAR]
    int64_t out_off; int32_t out_ld;                                      // h written at out_off + row*out_ld
    int64_t gates_off, c_off, hprev_off;                                  // saved (B*S,4H),(B*S,H),(B*S,H)
    int64_t tc_off;      // saved tanh(c_t) (B*S,H)                         [SP_TMP]
    int64_t gout_off;    // gradient of out (same ld)                      [SP_WS]
    int64_t gzx_off;     // gradient of zx, written (=)
    int64_t whht_off;    // [SP_TMP] W_hh transposed (H x 4H), built per forward when H > 64
    // multi-workgroup flavour (one-clip plans, H = 192, batch 1): the gate rows are split over LSTM_NB workgroups that exchange
    // h_t (forward) / dz_t (backward) through tagged 8-byte granules
    int32_t multi;       // 0 single-workgroup flavours, 1 multi-workgroup, 2 multi-workgroup with an injected exchange fault (tests)
    int64_t xch_off;     // [SP_TMP] forward 2 x H granules, then backward 2 x 4H granules (8 bytes each)
    int64_t status_off;  // [SP_WS, clip 0] the plan's device status word (int32): MST_DEV_* bits, sticky until the host clears it
};
#define LSTM_NB 12       // workgroups per sequence in the multi-workgroup flavour: 16 hidden units (64 gate rows) each at H = 192
#define LSTM_MH 192

// ---- combine (style/model.py:796-815): out = sum_c x_c n_c / sum_c n_c
#define COMBINE_MAXC 32
#define COMBINE_MAXBLK 256
#define COMBINE_SMALL 4096   // slices up to this many elements: reduction + elementwise pass in one single-workgroup launch
struct CombineDesc {
    int32_t Cn, rows, cols, ld;  // each slice: rows x cols, row stride ld
    int64_t x_off, cs;           // slice c at x_off + c*cs                [SP_WS]
    int64_t out_off;             // rows*cols contiguous
    int64_t stats_off;           // Cn + 1 floats: n_c..., S  (+ partial scratch behind it)
    int64_t part_off;            // COMBINE_MAXBLK * (Cn+1) partials
    int64_t gx_off, gout_off;    // gradient slots (gx has the x layout)
    int32_t nblk;
    int32_t first;               // backward: 1 = first writer of gx in this pass: store instead of +=
};

// ---- note-level fused stages
struct NotesDesc {
    int32_t C, Q;                // channels, R*T
    int32_t W, CW, ML;           // melody width, channels_linear width, melody_linear width
    int64_t oct_off, deg_off;    // ME: (P, 8W),(P,7W);  PSA: unused (the octave / degree rows are rt + it, below)
    // PSA: octave_linear / scale_degree_linear act on a broadcast-concat [style | rhythm(q,f) | instrument(c)], so their
    // pre-activations decompose: z[c, qf] = rt[qf] + it[c] (it carries the style part and the bias).  [SP_WS]
    int64_t rt_oct_off, rt_deg_off;      // (Q*F, 240), (Q*F, 210)
    int64_t it_oct_off, it_deg_off;      // (C, 240), (C, 210)
    // (the backward kernel sums dL/dz over the channels itself and writes the rt gradients at the same offsets in SP_GRAD)
    int64_t x_off; int32_t x_space;  // ME: pitched input
    int64_t mel_off, g_mel_off;  // PSA: melody (Q*F*56, W) [SP_WS] and its gradient [SP_GRAD]: melody_linear is applied inside the note kernels
    int64_t wm_off;              // PSA: melody_linear.weight (ML x W), immediately followed by its bias (ML) [SP_PAR]
    int64_t ml_off;              // (unused since round 3: the melody_linear activations are no longer materialised)
    int64_t wc_off, bc_off, wl_off, bl_off;   // params (ME: channels_linear, linear; PSA: linear only in wl/bl)
    int64_t out_off;             // ME: melody (Q,F,56,W), the channels already combined; PSA: (P,F,56,5)
    int64_t g_out_off, g_oct_off, g_deg_off, g_ml_off;   // PSA: g_oct / g_deg = gradient of the PRE-activations z, (P*F, 240 | 210)
    int64_t slab_off; int32_t slab_stride; int32_t nblk;    // one slab row per workgroup
    // PSA backward (psa_bwd2_kernel): per-workgroup partial sums over qf of dL/dz per channel = partial gradients of `it`
    // [SP_GRAD] (nblk, C, 240) and (nblk, C, 210); the loss tail's saved Jacobian and upstream loss gradients [SP_WS]
    int64_t itp_oct_off, itp_deg_off, loss_saved_off, loss_gl_off;
    // ME only: the channel combine (style/model.py:296,796-815) is fused in.  nwc waves per channel leave partial sums:
    int32_t fhn;                 // me_notes_fwd: waves per q, each taking NF / fhn fractions (a divisor of NF)
    int32_t nwc;                 // <= 64
    int64_t part_off;            // [SP_TMP] forward: C*nwc partial sums of squares; backward: C*nwc partial a_c, then nwc partial b
    int64_t stats_off;           // [SP_TMP] n_c (C floats), S
};

// ---- the note-axis convolution of batched plans on its own kernels (conv.hip): ONE descriptor for all clips
struct ConvDesc {
    int32_t P, clips, OC, splits;        // positions per clip, clips, out channels (57), k-splits of the weight gradient
    int64_t rows_per_split;              // rows (position, octave) per split: a multiple of 32
    int64_t x1_off, clip_stride;         // conv output (P, OC * 8) [SP_WS / SP_GRAD], floats between two clips' arenas
    int64_t wp_off;                      // [SP_TMP, clip 0] permuted, zero-padded weights W' (720 x 64)
    int64_t w_off, b_off;                // [SP_PAR] Conv1d weight (OC, 50, 14), bias
    int64_t slab_off, slab_stride;       // [SP_TMP, clip 0] one slab of OC * 700 + OC per split, parameter layout
};
int launch_conv_prep(const ConvDesc& d, Bases b, hipStream_t s);
int launch_conv_fwd(const ConvDesc& d, Bases b, hipStream_t s);
int launch_conv_dw(const ConvDesc& d, Bases b, hipStream_t s);

// The 2 x 2-blocked MFMA body shared by conv.hip / lin.hip: a wave's 64 x 64 outputs (four f32 32x32x2 accumulators) over one
// staged k-tile of KT rows, As[k][PA] / Bs[k][PB] k-major in LDS.  The operands of k-step s + 1 are read from LDS BEFORE the four
// MFMAs of step s are issued (two register sets): written the plain way (read, then use) the compiler waits lgkmcnt(0) in front of
// every group of four MFMAs and the LDS latency is exposed 16 times per k-tile (PMC: 60 % MFMA-busy at two waves per SIMD).
typedef float mst_f32x16 __attribute__((ext_vector_type(16)));
template <int KT, bool M1 = true, bool N1 = true, int PA = 0, int PB = 0>
__device__ __forceinline__ void mst_mfma_ktile_2x2(const float (*As)[PA], const float (*Bs)[PB], int arow, int bcol, int l31, int kh,
                                                   mst_f32x16 (&acc)[2][2]) {
    // M1 / N1: the wave's second 32-row / 32-column block is live (ragged tiles skip the dead blocks' MFMAs and reads)
    float a0[2], a1[2], b0[2], b1[2];
    a0[0] = As[kh][arow + l31]; b0[0] = Bs[kh][bcol + l31];
    if (M1) a1[0] = As[kh][arow + 32 + l31];
    if (N1) b1[0] = Bs[kh][bcol + 32 + l31];
#pragma unroll
    for (int s = 0; s < KT / 2; ++s) {
        const int c = s & 1, n = c ^ 1;
        if (s + 1 < KT / 2) {
            const int k = 2 * (s + 1) + kh;
            a0[n] = As[k][arow + l31]; b0[n] = Bs[k][bcol + l31];
            if (M1) a1[n] = As[k][arow + 32 + l31];
            if (N1) b1[n] = Bs[k][bcol + 32 + l31];
        }
        __builtin_amdgcn_sched_barrier(0);           // (the scheduler otherwise sinks the reads back behind the MFMAs to save four registers)
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc[0][0], 0, 0, 0);
        if (N1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc[0][1], 0, 0, 0);
        if (M1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc[1][0], 0, 0, 0);
        if (M1 && N1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}


// ---- dense nn.Linear of batched plans on its own kernels (lin.hip): ONE descriptor for all clips
struct LinDesc {
    int32_t rows, K, N, act, xgrad, clips;       // rows per clip; Linear K -> N
    int32_t x_space, x_ld; int64_t x_off, x_cs;  // input rows at base[x_space] + x_off + clip * x_cs + r * x_ld
    int32_t y_ld; int64_t y_off, y_cs;           // output rows [SP_WS] and their gradient [SP_GRAD]
    int64_t gx_cs;                               // clip stride of the input's gradient [SP_GRAD, at x_off / x_ld]
    int64_t w_off, b_off;                        // [SP_PAR] weight (N x K), bias (N)
    int64_t slab_off, slab_stride, rows_per_split; int32_t splits;   // dW | db slabs [SP_TMP, clip 0], parameter layout
    int32_t first[2];                            // dX stores instead of accumulating: in the per-stage / the whole-model backward list
};
int launch_lin_fwd(const LinDesc& d, Bases b, hipStream_t s);
int launch_lin_dx(const LinDesc& d, Bases b, int first, hipStream_t s);
int launch_lin_dw(const LinDesc& d, Bases b, hipStream_t s);
int lin_dw_tiles(const LinDesc& d);

// ---- row-wise tiny Linear: y[r, :] = act(W x[r, :] + b) with K_in <= 8 and N_out <= 20 over very many rows
// (PitchedStyleApplier.melody_linear 8->20 over positions x 56 notes, UnpitchedStyleApplier.linear 8->2 over
// positions x 47 notes; style/model.py:606-610,660-662,694-701,722-723).  As GEMMs these are all tile padding.
struct RowLinDesc {
    int32_t rows, kin, nout, act, xgrad;
    int64_t x_off, y_off;        // [SP_WS] rows x kin / rows x nout, contiguous; gradients at the same offsets in SP_GRAD
    int64_t w_off, b_off;        // [SP_PAR] weight (nout x kin) immediately followed by bias (nout)
    int64_t slab_off; int32_t slab_stride, nblk;   // [SP_TMP] one row of nout*kin + nout partial weight gradients per workgroup
    int32_t first;               // backward: 1 = first writer of dx in this pass: store instead of +=
};

// ---- bar tiling of one clip over several ranks (mst_plan_options.tile_rows > 0, plan.hip "tiled") ----------------
// strided row copy: dst[a * dst_sa + b * dst_sb + j] = src[a * src_sa + b * src_sb + j], j < cols (forward, activations);
// backward: g_src (+)= g_dst over the same index map
struct CopyDesc { int64_t src_off, dst_off; int32_t na, nb, cols, src_sa, src_sb, dst_sa, dst_sb, first; };
// fold: sum[c] = sum over rows of part[row * row_stride + c * col_stride] (ordered) — the rank-local partial sums of a
// global reduction; the host all-reduces `sum` over the ranks; spread writes it back as row 0 of the partials (other rows 0)
// so that the unchanged consumer kernel re-sums exactly the global value
struct FoldDesc { int32_t space, nrows, row_stride, ncols, col_stride, pad; int64_t part_off, sum_off /* SP_TMP */; };
int launch_copy_rows(const CopyDesc* dev, const CopyDesc& host, int backward, Bases b, hipStream_t s);
int launch_fold(const FoldDesc* dev, int spread, Bases b, hipStream_t s);
// one half of a two-launch combine: which = 0 sums of squares, 1 apply, 2 backward reduce, 3 backward apply
int launch_combine_phase(const CombineDesc* dev, int nblk, int which, Bases b, hipStream_t s);

// ---- deferred weight-gradient reduction: gpar[dst+i] += sum_s ws[src + s*stride + i]
// (reps = clips of a batched plan: clip r's slabs sit rep_stride floats after clip r-1's; summed clip-major, in order)
// width > 0: the entry is a (count / width) x width block of a wider parameter matrix: element i lands at
// dst + (i / width) * dst_ld + i % width (column blocks of a Linear whose input is a broadcast-concat, plan.hip linear_part)
struct SlabEntry { int64_t dst, src, stride; int32_t count, splits; int32_t reps; int64_t rep_stride; int32_t width, dst_ld; int32_t single; /* 1: one slab for all clips (folded GEMM) */
                   int64_t base; /* width > 0: offset of element (0, 0) of the parameter matrix the block belongs to (host-side disjointness check) */ };
struct SlabBlock { int32_t entry, start; };   // one workgroup's 64-element slice of an entry

// ---- derived layer sizes (style/model.py:31-33 and every ctor)
struct Sizes {
    // pce
    int OC, PCE_IL, H, HB;
    // se
    int SE_L, SE_IL, SE_ML, SE_BL;
    // me
    int ME_BL, ME_BRL, ME_IL, ME_CW, MEL;
    // pre / ure
    int RH, PRE_BL, PRE_BRL, PRE_CL, PRE_IL, PRE_ML, PRE_BPL, URE_CL;
    // sim
    int SIM_BL, NRF, SIM_SI, SIM_RI, SIM_SM, SIM_RM, SIM_SB, SIM_RB, NI;
    // psa / usa
    int PSA_SL, PSA_RL, PSA_IL, PSA_ML, USA_SL, USA_RL;
    int I, STYLE, BAR;
};
Sizes mst_sizes(const mst_dims& d);

// ---- launchers (each returns a hipError_t-style int; 0 = ok)
enum { GV_LIN_FWD, GV_LIN_FWD_PERM, GV_LIN_DW, GV_LIN_DW_PERM, GV_LIN_DA, GV_CONV_FWD, GV_CONV_DW, GV_HH_DW, GV_LIN_DA_RAW };
int launch_gather(const GatherDesc* dev, int count, int max_rows, Bases b, hipStream_t s);
int gemm_variant(const GemmDesc& g);
// descs = clips x members (clip-major); every clip's copy of a member has the same blk_begin inside the clip's block range
// mfma = 1: the 64x64-tile f32-MFMA kernel (throughput), 0: the 32x32 split-K kernel (latency); blk_begin must have been
// computed with the matching tile edge (gemm_tile_edge)
// dev_owner: member index of every workgroup of one clip's block range (the plan's block -> member table of the launch)
int launch_gemm(const GemmDesc* dev_descs, const int* dev_owner, int members, int blocks_per_clip, int clips, int mfma, Bases b, hipStream_t s);
int gemm_tile_edge(int mfma);
int gemm_blocks(const GemmDesc& g, int mfma);       // workgroups of one member (tiles / run x k-splits)
int launch_segred(const SegRedDesc* dev_descs, int members, int blocks_per_clip, int clips, int stage2_blocks, Bases b, hipStream_t s);
int launch_lstm_transpose(const LstmDesc* dev_descs, int count, int maxH, int multi, Bases b, hipStream_t s);
int launch_lstm_fwd(const LstmDesc* dev_descs, int count, int maxB, int maxH, int multi, Bases b, hipStream_t s);
int launch_lstm_bwd(const LstmDesc* dev_descs, int count, int maxB, int maxH, int multi, Bases b, hipStream_t s);
int lstm_multi_blocks_per_cu();      // workgroups of the multi-workgroup LSTM kernels one CU holds at once (occupancy API)
int launch_combine_fwd(const CombineDesc* dev_descs, int count, int max_nblk, int all_small, Bases b, hipStream_t s);
int launch_combine_bwd(const CombineDesc* dev_descs, int count, int max_nblk, int all_small, Bases b, hipStream_t s);
// `count` descriptors of identical shape (the clips of a batched plan), blockIdx.y = descriptor
int launch_me_sumsq(const NotesDesc* dev, const NotesDesc& host, int count, Bases b, hipStream_t s);
int launch_me_notes_fwd(const NotesDesc* dev, const NotesDesc& host, int count, Bases b, hipStream_t s);
int launch_me_bwd_reduce(const NotesDesc* dev, const NotesDesc& host, int count, Bases b, hipStream_t s);
int launch_me_notes_bwd(const NotesDesc* dev, const NotesDesc& host, int count, Bases b, hipStream_t s);
int launch_psa_notes_fwd(const NotesDesc* dev, const NotesDesc& host, int count, Bases b, hipStream_t s);
int launch_psa_notes_bwd(const NotesDesc* dev, const NotesDesc& host, int count, Bases b, hipStream_t s);
int psa_bwd_waves(int C);     // waves per workgroup of the applier's backward kernel (one per channel pair)
// per-clip strides of a batched loss evaluation (all 0 for the stand-alone single-clip entry points)
struct LossBatch { int32_t clips; int64_t ws, grad, tmp, ext0, ext1; };
// the two halves of loss_fwd_batched for tiled plans: partial sums, then (after the ranks' sums met) tail
int loss_fwd_partials(const float* pp, const float* pt, int64_t np, const float* up, const float* ut, int64_t nu, float* scratch,
                      hipStream_t s);
int loss_fwd_tail(int64_t np, int64_t nu, int has_u, const float* il, const float* it, int ni, const float* mlg, const float* mt,
                  const float* bp, const float* bt, int normalize, float* losses, float* saved, float* scratch, hipStream_t s,
                  float* gl_onehot, float* losses_out);
int loss_blocks(int64_t n);
int loss_fwd_batched(const float* pp, const float* pt, int64_t np, const float* up, const float* ut, int64_t nu, const float* il,
                     const float* it, int ni, const float* mlg, const float* mt, const float* bp, const float* bt, int normalize,
                     float* losses, float* saved, float* scratch, LossBatch lb, hipStream_t s,
                     float* gl_onehot /* nullable: [clips x ws stride] one-hot on the total */,
                     float* losses_out /* nullable: clips x MST_N_LOSSES dense copy */);
int loss_bwd_batched(const float* pp, const float* pt, int64_t np, const float* up, const float* ut, int64_t nu, const float* il,
                     const float* it, int ni, const float* mlg, const float* mt, const float* bp, const float* bt,
                     const float* saved, const float* gl, float* gp, float* gu, float* gi, float* gm, float* gb, LossBatch lb,
                     hipStream_t s, float info_scale = 1.f /* 0 on the non-root ranks of a tiled plan: the song-info gradients are
                                                              replicated, only one rank may contribute them to the summed gradient */);
bool rowlin_supported(int kin, int nout);
int launch_rowlin_fwd(const RowLinDesc* dev, const RowLinDesc& host, int count, Bases b, hipStream_t s);
int launch_rowlin_bwd(const RowLinDesc* dev, const RowLinDesc& host, int count, Bases b, hipStream_t s);
int launch_slab_reduce(const SlabEntry* dev, const SlabBlock* blocks, int nblocks, Bases b, hipStream_t s);
// gradient ranges still cleared before a backward pass (offsets into clip 0's slice of the gradient arena)
struct ZeroChunk { int64_t off; int32_t len; int32_t pad; };
int launch_zero(const ZeroChunk* dev, int nchunks, int clips, float* grad_base, int64_t clip_stride, hipStream_t s);
bool notes_widths_supported(int W, int CW, int ML);

#define GEMM_BM 32
#define GEMM_BN 32
#define GEMM_BK 128     // k-tile staged in LDS per step
#define GEMM_THREADS 1024 // lanes per workgroup: 16 waves split every k-tile
