// Plan builder + C ABI.  A plan is the whole StyleTransferModel (style/model.py:727-793) for one
// (layer widths, clip shape) unrolled into a static list of ops; every op carries its forward
// launch steps AND the backward steps that undo it, so loss.backward() is "walk the list in
// reverse".  All descriptors are uploaded once at plan creation; running a plan only enqueues
// kernels on the caller's stream (no allocation, no host sync => hipGraph-capturable).
//
// Workspace layout (floats):  [ activations | gradients (same offsets) | scratch ], each arena holding
// one slice per clip.  A plan for K clips (mst_dims.clips) is the one-clip plan with every launch
// widened: at schedule time each descriptor is replicated K times with its workspace / note-tensor
// offsets relocated to clip k's slices, so the launch count stays that of one clip while every
// launch carries K times the work.  Clips never mix: `combine`, the loss tree and the LSTM chains are
// per clip (B = 1 in the reference); only the parameter gradients are summed over clips, in order.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <array>
#include <vector>

#include "mst_common.h"

#define MST_ALIGN_UP(n) (((n) + 63) / 64 * 64)

// ------------------------------------------------------------------------------------------ sizes
static int mean_size(double a, double b, double factor = 1.0) { return (int)std::ceil(((a + b) / 2.0) * factor); }

Sizes mst_sizes(const mst_dims& d) {
    Sizes z{};
    z.I = d.instr; z.STYLE = d.style; z.BAR = d.bar; z.H = d.beat; z.HB = d.bar / 2; z.RH = d.rhythm;
    z.MEL = d.melody; z.NRF = d.nrf; z.NI = d.n_instruments;
    z.OC = mean_size(NF * NPF, d.beat);                       // style/model.py:41-42
    z.PCE_IL = mean_size(d.instr, d.beat);                    // :43
    z.SE_L = mean_size(d.bar, d.style);                       // :148-151
    z.SE_IL = mean_size(d.instr, d.style, .25);
    z.SE_ML = mean_size(2, d.style, .1);
    z.SE_BL = mean_size(d.style, 1, .05);
    z.ME_BL = mean_size(d.beat, d.melody);                    // :207-211
    z.ME_BRL = mean_size(d.bar, d.melody);
    z.ME_IL = mean_size(d.instr, d.melody, .25);
    z.ME_CW = mean_size(NPF, d.melody);
    z.PRE_BL = mean_size(d.beat, d.rhythm);                   // :305-311
    z.PRE_BRL = mean_size(d.bar, d.rhythm, .5);
    z.PRE_CL = mean_size(NPN * NPF, d.rhythm, .1);
    z.PRE_IL = mean_size(d.instr, d.rhythm, .5);
    z.PRE_ML = mean_size(2, d.rhythm, .25);
    z.PRE_BPL = mean_size(1, d.rhythm, .25);
    z.URE_CL = mean_size(NUN * NUF, d.rhythm, .25);           // :390-391
    z.SIM_BL = mean_size(NF * d.rhythm, d.nrf, .05);          // :450-460
    z.SIM_SI = mean_size(d.style, d.n_instruments, .05);
    z.SIM_RI = mean_size(d.rhythm, d.n_instruments, .25);
    z.SIM_SM = mean_size(d.style, 2, .01);
    z.SIM_RM = mean_size(d.rhythm, 2, .1);
    z.SIM_SB = mean_size(d.style, 1, .01);
    z.SIM_RB = mean_size(d.rhythm, 1, .1);
    z.PSA_SL = mean_size(d.style, NPF, .5);                   // :586-590
    z.PSA_RL = mean_size(d.rhythm, NPF, .5);
    z.PSA_IL = mean_size(d.instr, NPF, .4);
    z.PSA_ML = mean_size(d.melody, NPF, 3);
    z.USA_SL = mean_size(d.style, NUF, .5);                   // :682-684
    z.USA_RL = mean_size(d.rhythm, NUF, 1);
    return z;
}

// ------------------------------------------------------------------------------------------ params
struct PInfo { std::string name; int64_t off, numel; int ndim; int shape[3]; };

struct ParamTable {
    std::vector<PInfo> v;
    std::map<std::string, int> idx;
    int64_t top = 0;
    void add(const std::string& name, int a, int b = 0, int c = 0) {
        PInfo p; p.name = name; p.off = top; p.ndim = 1 + (b > 0) + (c > 0);
        p.shape[0] = a; p.shape[1] = b; p.shape[2] = c;
        p.numel = (int64_t)a * (b > 0 ? b : 1) * (c > 0 ? c : 1);
        top += p.numel;
        idx[name] = (int)v.size();
        v.push_back(p);
    }
    void lin(const std::string& pre, int out, int in) { add(pre + ".weight", out, in); add(pre + ".bias", out); }
    void lstm(const std::string& pre, int in, int hid, bool bi) {
        for (int r = 0; r < (bi ? 2 : 1); ++r) {
            const std::string sfx = r ? "_reverse" : "";
            add(pre + ".weight_ih_l0" + sfx, 4 * hid, in);
            add(pre + ".weight_hh_l0" + sfx, 4 * hid, hid);
            add(pre + ".bias_ih_l0" + sfx, 4 * hid);
            add(pre + ".bias_hh_l0" + sfx, 4 * hid);
        }
    }
    int64_t off(const std::string& name) const {
        auto it = idx.find(name);
        if (it == idx.end()) { fprintf(stderr, "mst: unknown parameter %s\n", name.c_str()); abort(); }
        return v[it->second].off;
    }
};

// registration order == model.parameters() order (style/model.py:737-749 and each ctor)
static void build_params(const mst_dims& d, const Sizes& z, ParamTable& t) {
    std::string m = "pitched_channels_encoder";
    t.add(m + ".beats_conv.module.weight", z.OC, NF * NPF, CONV_K);
    t.add(m + ".beats_conv.module.bias", z.OC);
    t.lin(m + ".instruments_linear", z.PCE_IL, z.I);
    t.lin(m + ".linear", z.H, z.OC * NOCT + z.PCE_IL);
    t.lstm(m + ".beats_lstm.module", z.H, z.H, false);
    t.lstm(m + ".bars_lstm", z.H, z.HB, true);
    m = "unpitched_channels_encoder";
    t.lin(m + ".linear", z.H, NF * NUN * NUF);
    t.lstm(m + ".beats_lstm.module", z.H, z.H, false);
    t.lstm(m + ".bars_lstm", z.H, z.HB, true);
    m = "style_encoder";
    t.lstm(m + ".bars_lstm", z.BAR, z.SE_L, false);
    t.lin(m + ".instruments_linear", z.SE_IL, z.I);
    t.lin(m + ".mode_linear", z.SE_ML, 2);
    t.lin(m + ".bpm_linear", z.SE_BL, 1);
    t.lin(m + ".linear", z.STYLE, z.SE_L + z.SE_IL + z.SE_ML + z.SE_BL);
    m = "melody_encoder";
    t.lin(m + ".beats_linear", z.ME_BL, z.H);
    t.lin(m + ".bars_linear", z.ME_BRL, z.BAR);
    t.lin(m + ".instruments_linear", z.ME_IL, z.I);
    t.lin(m + ".octave_linear", z.MEL * NOCT, z.ME_BL + z.ME_BRL + z.ME_IL);
    t.lin(m + ".scale_degree_linear", z.MEL * NDEG, z.ME_BL + z.ME_BRL + z.ME_IL);
    t.lin(m + ".channels_linear", z.ME_CW, NPF);
    t.lin(m + ".linear", z.MEL, z.MEL + z.ME_CW);
    m = "pitched_rhythm_encoder";
    t.lin(m + ".beats_linear", z.PRE_BL, z.H);
    t.lin(m + ".bars_linear", z.PRE_BRL, z.BAR);
    t.lin(m + ".channels_linear", z.PRE_CL, NPN * NPF);
    t.lin(m + ".instruments_linear", z.PRE_IL, z.I);
    t.lin(m + ".mode_linear", z.PRE_ML, 2);
    t.lin(m + ".bpm_linear", z.PRE_BPL, 1);
    t.lin(m + ".linear", z.RH, z.PRE_BL + z.PRE_BRL + z.PRE_CL + z.PRE_IL + z.PRE_ML + z.PRE_BPL);
    m = "unpitched_rhythm_encoder";
    t.lin(m + ".beats_linear", z.PRE_BL, z.H);
    t.lin(m + ".bars_linear", z.PRE_BRL, z.BAR);
    t.lin(m + ".channels_linear", z.URE_CL, NUN * NUF);
    t.lin(m + ".bpm_linear", z.PRE_BPL, 1);
    t.lin(m + ".linear", z.RH, z.PRE_BL + z.PRE_BRL + z.URE_CL + z.PRE_BPL);
    m = "song_info_model";
    t.lstm(m + ".beats_lstm.module", NF * z.RH, z.SIM_BL, false);
    t.lstm(m + ".bars_lstm", z.SIM_BL, z.NRF, false);
    t.lin(m + ".style_instruments_linear", z.SIM_SI, z.STYLE);
    t.lin(m + ".rhythm_instruments_linear", z.SIM_RI, z.NRF);
    t.lin(m + ".instruments_linear", z.NI, z.SIM_SI + z.SIM_RI);
    t.lin(m + ".style_mode_linear", z.SIM_SM, z.STYLE);
    t.lin(m + ".rhythm_mode_linear", z.SIM_RM, z.NRF);
    t.lin(m + ".mode_linear", 2, z.SIM_SM + z.SIM_RM);
    t.lin(m + ".style_bpm_linear", z.SIM_SB, z.STYLE);
    t.lin(m + ".rhythm_bpm_linear", z.SIM_RB, z.NRF);
    t.lin(m + ".bpm_linear", 1, z.SIM_SB + z.SIM_RB);
    m = "pitched_style_applier";
    t.lin(m + ".style_linear", z.PSA_SL, z.STYLE);
    t.lin(m + ".rhythm_linear", z.PSA_RL, z.RH);
    t.lin(m + ".instruments_linear", z.PSA_IL, z.I);
    t.lin(m + ".octave_linear", NPF * 6 * NOCT, z.PSA_SL + z.PSA_RL + z.PSA_IL);
    t.lin(m + ".scale_degree_linear", NPF * 6 * NDEG, z.PSA_SL + z.PSA_RL + z.PSA_IL);
    t.lin(m + ".melody_linear", z.PSA_ML, z.MEL);
    t.lin(m + ".linear", NPF, NPF * 6 + z.PSA_ML);
    m = "unpitched_style_applier";
    t.lin(m + ".style_linear", NF * z.USA_SL, z.STYLE);
    t.lin(m + ".rhythm_linear", z.USA_RL, z.RH);
    t.lin(m + ".notes_linear", NUN * NUF * 4, z.USA_SL + z.USA_RL);
    t.lin(m + ".linear", NUF, NUF * 4);
}

static bool dims_ok(const mst_dims* d) {
    // C <= 24: the applier's backward kernel runs one wave per channel pair of a qf in ONE workgroup (LDS of 12 waves); a MIDI file
    // has 16 channels, so the reference's own inputs stop at 15 pitched channels
    return d && d->C >= 1 && d->C <= COMBINE_MAXC && d->C <= 24 && d->R >= 1 && d->T >= 1 && d->beat >= 1 && d->bar >= 2 &&
           d->bar % 2 == 0 && d->nrf >= 1 && d->style >= 1 && d->melody >= 1 && d->rhythm >= 1 && d->instr >= 1 &&
           d->n_instruments >= 1 && d->clips >= 0 && d->clips <= 4096;
}

extern "C" int32_t mst_param_count(const mst_dims* d) {
    if (!dims_ok(d)) return MST_ERR_ARG;
    ParamTable t; build_params(*d, mst_sizes(*d), t);
    return (int32_t)t.v.size();
}
extern "C" int64_t mst_param_floats(const mst_dims* d) {
    if (!dims_ok(d)) return MST_ERR_ARG;
    ParamTable t; build_params(*d, mst_sizes(*d), t);
    return t.top;
}
extern "C" int32_t mst_param_info(const mst_dims* d, int32_t i, char* name, int32_t cap, int64_t* off, int32_t* ndim,
                                  int32_t shape[3]) {
    if (!dims_ok(d) || !name || !off || !ndim || !shape) return MST_ERR_ARG;
    ParamTable t; build_params(*d, mst_sizes(*d), t);
    if (i < 0 || i >= (int)t.v.size()) return MST_ERR_ARG;
    const PInfo& p = t.v[i];
    snprintf(name, cap, "%s", p.name.c_str());
    *off = p.off; *ndim = p.ndim;
    for (int k = 0; k < 3; ++k) shape[k] = p.shape[k];
    return MST_OK;
}

// ------------------------------------------------------------------------------------------ plan
struct T { int64_t off; int rows, cols, ld; };
struct SegIn { int space; int64_t off; int ld, width; int s[4]; bool grad; };
enum { K_GEMM, K_GATHER, K_SEGRED, K_LSTM_F, K_LSTM_B, K_COMB_F, K_COMB_B, K_ME_F, K_ME_B, K_PSA_F, K_PSA_B, K_LSTM_T, K_ROW_F, K_ROW_B,
       K_ME_SQ, K_ME_RED,
       // tiled plans only (never merged): halves of a combine, strided copies, fold / spread around an exchange, the exchange
       K_COMB_F1, K_COMB_F2, K_COMB_B1, K_COMB_B2, K_COPY_F, K_COPY_B, K_FOLD, K_SPREAD, K_XCHG,
       K_GEMM_FOLD,        // weight-gradient GEMMs of a batched plan with the clips folded into K (one descriptor for all clips)
       K_CONV_P, K_CONV_F, K_CONV_W,
       K_LIN_F, K_LIN_A, K_LIN_W };     // ... and its large dense nn.Linear layers (lin.hip): forward, input gradient, weight gradient      // batched plans on the 64x64 tiling: the note-axis convolution on its own kernels (conv.hip), one descriptor for all clips
struct Step { int kind, first, count, a, b, stage; int c = 0; int lvl = 0; int chain = -1; int signal = -1; int wait[3] = {-1, -1, -1}; };
// c: GEMM steps — offset of the step's block -> member table; lvl: dependency level in its scheduled pass; chain / signal / wait: the stream the
// launch goes to, the event slot recorded behind it and the slots its stream waits for in front of it (assign_streams; chain -1 = caller's stream)
struct Acc { int space; int64_t lo, hi; bool w; bool accum = false, dense = true; };   // accum: a += writer; dense: covers [lo, hi) fully
struct Op { int stage; std::vector<Step> fwd, bwd; };

static int stage_idx(int stage) { return stage == MST_STAGE_EXTRACT ? 0 : stage == MST_STAGE_INFO ? 1 : 2; }

struct mst_plan {
    mst_dims d; Sizes z; ParamTable pt;
    std::vector<GemmDesc> gemms; std::vector<GatherDesc> gathers; std::vector<SegRedDesc> segreds; std::vector<LstmDesc> lstms;
    std::vector<CombineDesc> combines; std::vector<NotesDesc> notes; std::vector<RowLinDesc> rowlins; std::vector<SlabEntry> slabs[3];
    std::vector<ConvDesc> convs; std::vector<LinDesc> lins;
    std::vector<Op> ops;
    // scheduled launch lists (dependency-levelled, same-level steps merged) and their descriptor arrays
    std::vector<Step> sched[2];        // per-stage merging (stages may run separately)
    std::vector<Step> sched_all[2];    // stage-agnostic merging, used when all stages run together
    const std::vector<Step>& list(int mask, int backward) const { return mask == MST_STAGE_ALL ? sched_all[backward ? 1 : 0] : sched[backward ? 1 : 0]; }
    std::vector<GemmDesc> s_gemms; std::vector<GatherDesc> s_gathers; std::vector<SegRedDesc> s_segreds; std::vector<LstmDesc> s_lstms;
    std::vector<CombineDesc> s_combines; std::vector<NotesDesc> s_notes; std::vector<RowLinDesc> s_rowlins;
    RowLinDesc* d_rowlins = nullptr;
    // member of every workgroup of a clip's block range, per scheduled GEMM launch (Step.c): one uniform load instead of a binary
    // search over the members' first blocks — log2(members) dependent scalar round trips at the head of every workgroup
    std::vector<int> s_gemm_owner; int* d_gemm_owner = nullptr;
    std::map<std::string, T> named;
    int64_t act_top = 0, tmp_top = 0;
    int64_t stage_begin[3] = {0, 0, 0}, stage_end[3] = {0, 0, 0};
    GemmDesc* d_gemms = nullptr; GatherDesc* d_gathers = nullptr; SegRedDesc* d_segreds = nullptr; LstmDesc* d_lstms = nullptr;
    CombineDesc* d_combines = nullptr; NotesDesc* d_notes = nullptr;     // d_notes: the scheduled (per-clip) copies, s_notes
    SlabEntry* d_slabs[3] = {nullptr, nullptr, nullptr};
    std::vector<SlabBlock> slab_blocks[3]; SlabBlock* d_slab_blocks[3] = {nullptr, nullptr, nullptr};
    // all three stages' entries as ONE launch (a whole-model backward): valid when no parameter range is the target of two entries
    std::vector<SlabEntry> slabs_all; SlabEntry* d_slabs_all = nullptr;
    std::vector<SlabBlock> slab_blocks_all; SlabBlock* d_slab_blocks_all = nullptr;
    bool slabs_all_ok = false;
    T t_losses, t_saved, t_gl; int64_t loss_scratch = 0;
    std::vector<ZeroChunk> zero_stage[3], zero_all;      // gradient ranges mst_zero_grads clears (clip 0 coordinates)
    ZeroChunk* d_zero_stage[3] = {nullptr, nullptr, nullptr}; ZeroChunk* d_zero_all = nullptr;
    int err = 0;
    mst_plan_options opt{};       // as given to mst_plan_create_ex (zeros = defaults)
    int mfma = 0;                 // GEMM tiling of this plan: 1 = 64x64 tiles (batched, FLOP-bound plans), 0 = 32x32 split-K tiles (latency)

    // bar tiling (mst_plan_options.tile_rows > 0): this rank owns bars [tile_r0, tile_r0 + tile_rows) of the d.R bars
    bool tiled() const { return opt.tile_rows > 0; }
    int Rl() const { return tiled() ? opt.tile_rows : d.R; }
    int P() const { return d.C * Rl() * d.T; }          // positions / beats this plan computes (all of them when not tiled)
    int Q() const { return Rl() * d.T; }
    struct Xchg { int space; int64_t off; int32_t len; };
    std::vector<CopyDesc> copies; std::vector<FoldDesc> folds; std::vector<Xchg> xchgs;
    CopyDesc* d_copies = nullptr; FoldDesc* d_folds = nullptr;
    bool tags_in_zero = false;             // zero_all also clears the multi-workgroup LSTM's exchange tags
    std::vector<ZeroChunk> zero_fwd; ZeroChunk* d_zero_fwd = nullptr;      // activation ranges cleared before a tiled forward
    int64_t loss_sum_off = 0;                                               // [SP_TMP] 16 floats: the loss partial sums, folded
    int loss_fold[2] = {-1, -1};
    // one train iteration of a tiled plan as phases: each runs launch steps and ends at an exchange (or at the end)
    // what: 0 schedule steps, 1 loss partials, 2 loss tail + seed; xchg[0 .. nx): the exchanges the phase ends at — exchanges of
    // one dependency level are MERGED into one phase end (what runs between them in the level-ordered list cannot depend on them)
    struct Phase { int what; int pass; int begin, end; int nx; int xchg[MST_MAX_XCHG]; };
    std::vector<Phase> phases;
    int K() const { return d.clips > 1 ? d.clips : 1; }
    static int device_cus() {          // compute units of the current device (256 on MI355X in SPX mode)
#ifdef HIPSIM
        return 16;                     // the interpreter co-schedules at most 48 blocks (hipsim::COOP_MAXB): 3 slots x 16
#else
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return cus;
#endif
    }
    // Workgroups of ONE launch of the multi-workgroup LSTM kernels that are certainly resident together on this device: what the
    // occupancy API reports per CU for these kernels minus one slot per CU, times the CUs.  The spare slot is the room for
    // whatever else is running (a second stream's kernel, an RCCL kernel, the tail of the previous launch); a launch that needs
    // more takes the single-workgroup flavour.  On a whole MI355X: (4 - 1) x 256 = 768 workgroups = 64 clips.
    static int coresident_slots() {
        const int per_cu = lstm_multi_blocks_per_cu();
        return per_cu > 1 ? (per_cu - 1) * device_cus() : 0;
    }
    int64_t status_off = 0;            // [SP_WS, clip 0] device status word ("device_status")
    // side streams of run_pass (mst_plan_options.branches): same-level launches fork from / join to the caller's stream
    // The launches of such plans' whole-model lists are spread over 1 + N_SIDE streams along the dependency DAG (assign_streams);
    // a set of side streams / events belongs to ONE caller stream (two accumulation iterations captured side by side must not
    // meet in a shared side stream), created at the caller's first (eager) pass.
    enum { N_SIDE = 3, N_SETS = 4, N_EVENTS = 96 };
    struct SideSet { hipStream_t owner = nullptr; bool used = false; hipStream_t side[N_SIDE] = {}; hipEvent_t ev_fork = nullptr, ev_join[N_SIDE] = {}, ev[N_EVENTS] = {}; };
    mutable SideSet sides[N_SETS];
    bool branches = false;
    SideSet* side_set(hipStream_t main) const {
        for (SideSet& q : sides) if (q.used && q.owner == main) return &q;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;      // no object creation inside a capture
        for (SideSet& q : sides) {
            if (q.used) continue;
            bool ok = hipEventCreateWithFlags(&q.ev_fork, hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < N_SIDE; ++i)
                ok = ok && hipStreamCreateWithFlags(&q.side[i], hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&q.ev_join[i], hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < n_signals; ++i) ok = ok && hipEventCreateWithFlags(&q.ev[i], hipEventDisableTiming) == hipSuccess;
            if (!ok) return nullptr;
            q.used = true; q.owner = main;
            return &q;
        }
        return nullptr;                 // more caller streams than sets: that caller runs its chains back to back
    }
    int64_t ext0_stride() const { return (int64_t)P() * NF * NPN * NPF; }
    int64_t ext1_stride() const { return (int64_t)Q() * NF * NUN * NUF; }
    // offset of clip k's slice of `space` relative to clip 0's
    int64_t shift(int space, int k) const {
        switch (space) {
        case SP_WS: case SP_GRAD: return (int64_t)k * act_top;
        case SP_TMP: return (int64_t)k * tmp_top;
        case SP_EXT0: return (int64_t)k * ext0_stride();
        case SP_EXT1: return (int64_t)k * ext1_stride();
        default: return 0;                                  // parameters are shared
        }
    }
    GemmDesc reloc(GemmDesc g, int k) const {
        for (Operand* o : {&g.A, &g.B}) {
            o->off += shift(o->space, k);
            if (o->kind == OPK_ACTGRAD || o->kind == OPK_CONVGRAD) o->off2 += shift(o->space2, k);
        }
        g.out.off += shift(g.out.space, k);
        if (g.out.bias_space >= 0) g.out.bias_off += shift(g.out.bias_space, k);      // 0 for a parameter bias
        return g;
    }
    GatherDesc reloc(GatherDesc g, int k) const {
        g.out_off += shift(SP_WS, k);
        for (int i = 0; i < g.nseg; ++i) g.seg[i].off += shift(g.seg[i].space, k);
        return g;
    }
    SegRedDesc reloc(SegRedDesc r, int k) const {
        r.src_off += shift(SP_GRAD, k); r.dst_off += shift(SP_GRAD, k); r.part_off += shift(SP_TMP, k); r.y_off += shift(SP_WS, k);
        return r;
    }
    LstmDesc reloc(LstmDesc l, int k) const {           // whht (W_hh transposed) is shared: parameters only
        const int64_t a = shift(SP_WS, k), t = shift(SP_TMP, k);
        l.zx_off += a; l.out_off += a; l.gout_off += a; l.gzx_off += a;
        l.gates_off += t; l.c_off += t; l.hprev_off += t; l.tc_off += t; l.xch_off += t;
        return l;
    }
    CombineDesc reloc(CombineDesc c, int k) const {
        const int64_t a = shift(SP_WS, k), t = shift(SP_TMP, k);
        c.x_off += a; c.out_off += a; c.gx_off += a; c.gout_off += a; c.stats_off += t; c.part_off += t;
        return c;
    }
    RowLinDesc reloc(RowLinDesc r, int k) const {
        r.x_off += shift(SP_WS, k); r.y_off += shift(SP_WS, k); r.slab_off += shift(SP_TMP, k);
        return r;
    }
    NotesDesc reloc(NotesDesc n, int k) const {
        const int64_t a = shift(SP_WS, k), t = shift(SP_TMP, k);
        n.oct_off += a; n.deg_off += a; n.rt_oct_off += a; n.rt_deg_off += a; n.it_oct_off += a; n.it_deg_off += a; n.mel_off += a; n.g_mel_off += a; n.out_off += a; n.g_out_off += a; n.g_oct_off += a; n.g_deg_off += a;
        n.g_ml_off += a; n.x_off += shift(n.x_space, k); n.slab_off += t; n.part_off += t; n.stats_off += t;
        n.itp_oct_off += a; n.itp_deg_off += a; n.loss_saved_off += a; n.loss_gl_off += a;
        return n;
    }

    static int64_t align(int64_t n) { return (n + 63) / 64 * 64; }
    T newT(int rows, int cols, const char* name = nullptr) {
        T t{act_top, rows, cols, cols};
        act_top += align((int64_t)rows * cols);
        if (name) named[name] = t;
        return t;
    }
    int64_t tmp(int64_t n) { int64_t o = tmp_top; tmp_top += align(n); return o; }
    static int tiles(int M, int N) { return ((M + GEMM_BM - 1) / GEMM_BM) * ((N + GEMM_BN - 1) / GEMM_BN); }
    // weight-gradient GEMMs reduce over rows.  One clip per launch (latency-bound): one k-split per 256 rows (2 k-tiles),
    // capped at 64 slabs, for workgroup count.  Batched plans get their parallelism from the clips, so a split covers
    // up to 2048 rows (64 MFMA k-tiles, at 64 clips) and the slab traffic of the deferred reduction shrinks accordingly.
    // batched plans on the 64x64 tiling fold the clips into the reduction of their weight-gradient GEMMs (GemmDesc.fold_rows)
    bool mfma_plan() const { return opt.gemm_tile == 64 || (opt.gemm_tile == 0 && K() >= 6); }      // the throughput GEMM tiling
    bool folds_clips() const { return K() > 1 && mfma_plan(); }
    // k-splits of a folded reduction: enough (tile, split) workgroups to fill every workgroup slot of the chip once (256 CUs x 4), at least
    // 128 reduction rows per split; the slab a split writes is one weight gradient, so many splits of a small weight are cheap
    int fold_splits(int rows, int M, int N, int members) const {
        const int64_t kt = (int64_t)rows * K();
        const int64_t t = (int64_t)((M + 63) / 64) * ((N + 63) / 64) * members;      // `members` like GEMMs share the launch
        int64_t s = 1024 / t;                // rounded DOWN: 1025 workgroups are two rounds of the chip
        const int64_t kmax = kt / 128;
        if (s > kmax) s = kmax;
        while (s > 1 && (int64_t)M * N * s > (int64_t)4 << 20) --s;      // <= 16 MB of slab per weight
        s = s < 1 ? 1 : (s > 512 ? 512 : s);
        const int64_t chunk = ((kt + s - 1) / s + 31) / 32 * 32;          // the kernel rounds a split's share to whole k-tiles:
        return (int)((kt + chunk - 1) / chunk);                           // no split may come out empty (its slab would stay unwritten)
    }
    // turns the weight-gradient GEMM `w` (built for one clip, reduction over `rows`) into the folded form
    void fold(GemmDesc& w, int rows, int members = 1) {
        w.fold_rows = rows; w.K = rows * K(); w.ksplit = fold_splits(rows, w.M, w.N, members);       // the clip strides are set at schedule time
    }
    int splits_for(int rows) const {
        const int per = 32 * K() < 256 ? 256 : (32 * K() > 2048 ? 2048 : 32 * K());
        int s = (rows + per - 1) / per;
        return s < 1 ? 1 : (s > 64 ? 64 : s);
    }

    static SegIn seg(const T& t, int s0, int s1, int s2, int s3, bool grad = true) {
        return SegIn{SP_WS, t.off, t.ld, t.cols, {s0, s1, s2, s3}, grad};
    }
    static SegIn segx(int space, int ld, int width, int s0, int s1, int s2, int s3) {
        return SegIn{space, 0, ld, width, {s0, s1, s2, s3}, false};
    }

    // cat_with_broadcast (style/utils/pytorch.py:54-65) of several sources, materialised once.
    // forward: gather kernel; backward: segment-reduce of the concat's gradient into each source.
    // sum = true: the segments (equal widths) are added instead of concatenated — out = sum of the broadcast sources
    T gather(int stage, const int rs[4], const std::vector<SegIn>& segs, bool sum = false) {
        if (segs.size() > MAX_SEG) err = MST_ERR_UNSUPPORTED;
        const int rows = rs[0] * rs[1] * rs[2] * rs[3];
        int K = 0;
        for (auto& s : segs) { if (sum) { if (K && K != s.width) err = MST_ERR_UNSUPPORTED; K = s.width; } else K += s.width; }
        T out = newT(rows, K);
        GatherDesc g{}; g.rows = rows; g.K = K; g.nseg = (int)segs.size(); g.out_off = out.off; g.sum = sum ? 1 : 0;
        for (int q = 0; q < 4; ++q) g.d[q] = rs[q];
        int start = 0;
        for (size_t i = 0; i < segs.size(); ++i) {
            Seg& s = g.seg[i];
            s.space = segs[i].space; s.off = segs[i].off; s.ld = segs[i].ld; s.start = start; s.width = segs[i].width;
            for (int q = 0; q < 4; ++q) s.s[q] = segs[i].s[q];
            if (!sum) start += segs[i].width;
        }
        Op op; op.stage = stage;
        op.fwd.push_back(Step{K_GATHER, (int)gathers.size(), 1, rows, 0});
        gathers.push_back(g);
        int first = (int)segreds.size(), cnt = 0, maxidx = 1, stage2 = 0;
        start = 0;
        for (auto& s : segs) {
            if (s.grad) {
                SegRedDesc r{}; r.src_off = out.off; r.src_ld = K; r.start = start; r.width = s.width;
                r.dst_off = s.off; r.dst_ld = s.ld; r.nidx = 1;
                for (int q = 0; q < 4; ++q) {
                    r.d[q] = rs[q]; r.s[q] = s.s[q];
                    r.kd[q] = (s.s[q] != 0 && rs[q] > 1) ? rs[q] : 1;
                    r.nidx *= r.kd[q];
                }
                // destination rows must be the natural row-major index of the kept dims
                int nat = 1;
                for (int q = 3; q >= 0; --q) {
                    if (r.kd[q] > 1) { if (s.s[q] != nat) err = MST_ERR_UNSUPPORTED; nat *= r.kd[q]; }
                }
                int nred = 1;
                for (int q = 0; q < 4; ++q) if (r.kd[q] == 1) nred *= rs[q];
                r.nchunk = (nred + 63) / 64;
                if (r.nchunk > 1) {
                    r.part_off = tmp((int64_t)r.nidx * r.nchunk * r.width);
                    const int b2 = (r.nidx * r.width + 255) / 256;
                    if (b2 > stage2) stage2 = b2;
                }
                if (r.nidx * r.nchunk > maxidx) maxidx = r.nidx * r.nchunk;
                segreds.push_back(r); ++cnt;
            }
            if (!sum) start += s.width;
        }
        if (cnt) op.bwd.push_back(Step{K_SEGRED, first, cnt, maxidx, stage2});
        ops.push_back(op);
        return out;
    }

    // nn.Linear (+ activation) on a dense (rows x K, row stride ld) input that lives in `space`.
    // Emits the forward GEMM, the k-split weight|bias-gradient GEMM and, when the input needs a
    // gradient, the input-gradient GEMM accumulating straight into the input's gradient slot.
    T linear(int stage, int space, int64_t xoff, int xld, int rows, int K, bool xgrad, const std::string& wname,
             const std::string& bname, int N, int act, const T* out_opt = nullptr, int pb = 0, int pc = 0,
             const char* name = nullptr) {
        T out = out_opt ? *out_opt : newT(rows, N, name);
        const int64_t woff = pt.off(wname), boff = pt.off(bname);
        Op op; op.stage = stage;
        // Plans on the throughput tiling run their large dense Linears on lin.hip's kernels (2 x 2-blocked MFMA tiles, all clips
        // as rows of one launch): row-major input in the workspace or a borrowed note tensor, >= 512 rows and >= 4 MFLOP per clip
        // and more than one 32-column MFMA block of outputs — or at most 16 outputs of a long input row (the 16x16x4 stream kernels).  The rule does not look at the clip count: a one-clip plan forced
        // onto this tiling takes the same kernels, so "batched == one clip at a time, bit for bit" holds by construction.
        if (mfma_plan() && opt.dense_flavour != 1 && !pb && rows >= 32 && K >= 4 && N >= 4 && (space == SP_WS || ((space == SP_EXT0 || space == SP_EXT1) && !xgrad)) &&
            boff == woff + (int64_t)N * K && (int64_t)this->K() * rows < ((int64_t)1 << 30) &&
            (opt.dense_flavour == 2 || (rows >= 512 && N > 32 && 2.0 * rows * N * K >= 4e6) ||
             (rows >= 512 && N <= 16 && K >= 128 && K + 1 <= 320 && !xgrad))) {       // the N <= 16 stream kernels: 280 -> 16 over every note row
            LinDesc l{}; l.rows = rows; l.K = K; l.N = N; l.act = act; l.xgrad = xgrad ? 1 : 0; l.clips = this->K();
            l.x_space = space; l.x_ld = xld; l.x_off = xoff; l.y_ld = out.ld; l.y_off = out.off; l.w_off = woff; l.b_off = boff;
            const int64_t mtot = (int64_t)this->K() * rows;
            const int tl = lin_dw_tiles(l);
            int64_t sp = std::max<int64_t>(1, 768 / tl);       // three workgroups per CU, all co-resident
            int64_t per = ((mtot + sp - 1) / sp + 31) / 32 * 32;
            if (per < 128) per = 128;
            l.rows_per_split = per; l.splits = (int)((mtot + per - 1) / per);
            l.slab_stride = (int64_t)N * K + N;
            l.slab_off = tmp(l.slab_stride * l.splits);
            const int li = (int)lins.size();
            lins.push_back(l);
            op.fwd.push_back(Step{K_LIN_F, li, 1, 0, 0});
            op.bwd.push_back(Step{K_LIN_W, li, 1, 0, 0});
            if (xgrad) op.bwd.push_back(Step{K_LIN_A, li, 1, 0, 0});
            SlabEntry e1{woff, l.slab_off, l.slab_stride, N * K, l.splits}, e2{boff, l.slab_off + (int64_t)N * K, l.slab_stride, N, l.splits};
            e1.single = e2.single = 1;
            slabs[stage_idx(stage)].push_back(e1);
            slabs[stage_idx(stage)].push_back(e2);
            ops.push_back(op);
            return out;
        }
        {
            GemmDesc g{}; g.M = rows; g.N = N; g.K = K; g.ksplit = 1;
            g.A.kind = OPK_DENSE; g.A.space = space; g.A.off = xoff; g.A.si = xld; g.A.sj = 1; g.A.ones_at = -1; g.A.kfast = 1;
            if (pb) { g.B.kind = OPK_PERMW; g.B.space = SP_PAR; g.B.off = woff; g.B.ld = K; g.B.pb = pb; g.B.pc = pc; g.B.kfast = 1; }
            else { g.B.kind = OPK_DENSE; g.B.space = SP_PAR; g.B.off = woff; g.B.si = 1; g.B.sj = K; g.B.ones_at = -1; g.B.kfast = 1; }
            g.out.kind = OUT_STORE; g.out.space = SP_WS; g.out.ldc = out.ld; g.out.act = act; g.out.off = out.off;
            g.out.bias_space = SP_PAR; g.out.bias_off = boff;
            // one row per clip (the style / song-info heads): batched plans on the 64x64 tiling run the clips as the rows of ONE
            // GEMM (GemmDesc.clip_rows) — like the folded weight gradients, one descriptor for all clips
            // (a one-clip plan on this tiling marks them too: the flag also pins the summation order, gemm.hip)
            g.clip_rows = (mfma_plan() && rows == 1 && !pb && space == SP_WS) ? 1 : 0;
            op.fwd.push_back(Step{(g.clip_rows && folds_clips()) ? K_GEMM_FOLD : K_GEMM, (int)gemms.size(), 1, tiles(rows, N), 1});
            gemms.push_back(g);
        }
        {   // dW | db  =  (dY o act')^T [X | 1]
            GemmDesc w{}; w.M = N; w.N = K + 1; w.K = rows; w.ksplit = splits_for(rows);
            w.A.kind = OPK_ACTGRAD; w.A.space = SP_GRAD; w.A.off = out.off; w.A.space2 = SP_WS; w.A.off2 = out.off;
            w.A.ld = out.ld; w.A.act = act; w.A.transposed = 1; w.A.kfast = 0;
            w.B.kind = OPK_DENSE; w.B.space = space; w.B.off = xoff; w.B.si = xld; w.B.sj = 1; w.B.ones_at = K; w.B.kfast = 0;
            const bool fd = folds_clips();
            if (fd) fold(w, rows);
            const int64_t stride = (int64_t)N * K + N;
            const int64_t slab = tmp(stride * w.ksplit);
            w.out.kind = pb ? OUT_PERMW_SLAB : OUT_SLAB; w.out.space = SP_TMP; w.out.off = slab; w.out.slab_stride = stride;
            w.out.wcols = K; w.out.pb = pb; w.out.pc = pc; w.out.bias_space = -1;
            op.bwd.push_back(Step{fd ? K_GEMM_FOLD : K_GEMM, (int)gemms.size(), 1, tiles(N, K + 1), w.ksplit});
            gemms.push_back(w);
            SlabEntry e1{woff, slab, stride, N * K, w.ksplit}, e2{boff, slab + (int64_t)N * K, stride, N, w.ksplit};
            e1.single = e2.single = fd ? 1 : 0;
            slabs[stage_idx(stage)].push_back(e1);
            slabs[stage_idx(stage)].push_back(e2);
        }
        if (xgrad) {
            if (pb || space != SP_WS) { err = MST_ERR_UNSUPPORTED; }
            GemmDesc a{}; a.M = rows; a.N = K; a.K = N; a.ksplit = 1;
            a.A.kind = OPK_ACTGRAD; a.A.space = SP_GRAD; a.A.off = out.off; a.A.space2 = SP_WS; a.A.off2 = out.off;
            a.A.ld = out.ld; a.A.act = act; a.A.transposed = 0; a.A.kfast = 1;
            a.B.kind = OPK_DENSE; a.B.space = SP_PAR; a.B.off = woff; a.B.si = K; a.B.sj = 1; a.B.ones_at = -1; a.B.kfast = 0;
            a.out.kind = OUT_ACCUM; a.out.space = SP_GRAD; a.out.off = xoff; a.out.ldc = xld; a.out.bias_space = -1; a.out.act = ACT_NONE;
            a.clip_rows = (mfma_plan() && rows == 1 && !pb && space == SP_WS) ? 1 : 0;
            op.bwd.push_back(Step{(a.clip_rows && folds_clips()) ? K_GEMM_FOLD : K_GEMM, (int)gemms.size(), 1, tiles(rows, K), 1});
            gemms.push_back(a);
        }
        ops.push_back(op);
        return out;
    }
    T linear(int stage, const T& x, bool xgrad, const std::string& pre, int N, int act, const char* name = nullptr) {
        return linear(stage, SP_WS, x.off, x.ld, x.rows, x.cols, xgrad, pre + ".weight", pre + ".bias", N, act, nullptr, 0, 0, name);
    }

    // Conv1d(50 -> OC, k=14, s=7, p=4) over the note axis + leaky, as implicit-im2col GEMM (style/model.py:46-53,78-84)
    T conv(int stage) {
        const int P_ = P();
        T x1 = newT(P_, z.OC * NOCT, "pce_conv");
        const int64_t woff = pt.off("pitched_channels_encoder.beats_conv.module.weight");
        const int64_t boff = pt.off("pitched_channels_encoder.beats_conv.module.bias");
        const int K = NF * NPF * CONV_K;
        Op op; op.stage = stage;
        // batched plans on the 64x64 tiling: forward and weight gradient on conv.hip's kernels (window loads instead of per-element
        // im2col index math, 256 x 64 / 64 x 256 tiles); all clips are rows of ONE launch
        if (folds_clips() && z.OC <= 64 && P_ >= 8 && (int64_t)this->K() * P_ * NF * NPN * NPF < ((int64_t)1 << 32)) {
            ConvDesc c{}; c.P = P_; c.clips = this->K(); c.OC = z.OC; c.x1_off = x1.off; c.w_off = woff; c.b_off = boff;
            c.wp_off = tmp((int64_t)NF * 72 * 64);
            const int64_t rows = (int64_t)this->K() * P_ * NOCT;
            int64_t sp = 256;                                 // 3 column tiles x 256 splits = 768 workgroups: three per CU, all co-resident (166 VGPRs)
            int64_t per = ((rows + sp - 1) / sp + 31) / 32 * 32;
            if (per < 256) per = 256;
            c.rows_per_split = per; c.splits = (int)((rows + per - 1) / per);
            c.slab_stride = (int64_t)z.OC * K + z.OC;
            c.slab_off = tmp(c.slab_stride * c.splits);
            const int ci = (int)convs.size();
            convs.push_back(c);
            op.fwd.push_back(Step{K_CONV_P, ci, 1, 0, 0});
            op.fwd.push_back(Step{K_CONV_F, ci, 1, 0, 0});
            op.bwd.push_back(Step{K_CONV_W, ci, 1, 0, 0});
            SlabEntry e1{woff, c.slab_off, c.slab_stride, z.OC * K, c.splits}, e2{boff, c.slab_off + (int64_t)z.OC * K, c.slab_stride, z.OC, c.splits};
            e1.single = e2.single = 1;
            slabs[stage_idx(stage)].push_back(e1);
            slabs[stage_idx(stage)].push_back(e2);
            ops.push_back(op);
            return x1;
        }
        GemmDesc g{}; g.M = P_ * NOCT; g.N = z.OC; g.K = K; g.ksplit = 1;
        g.A.kind = OPK_IM2COL; g.A.space = SP_EXT0; g.A.off = 0; g.A.ones_at = -1; g.A.kfast = 1;
        g.B.kind = OPK_PERMW; g.B.space = SP_PAR; g.B.off = woff; g.B.ld = K; g.B.pb = CONV_K; g.B.pc = NPF; g.B.kfast = 1;
        g.out.kind = OUT_CONV; g.out.space = SP_WS; g.out.off = x1.off; g.out.ldc = z.OC * NOCT;
        g.out.bias_space = SP_PAR; g.out.bias_off = boff;
        op.fwd.push_back(Step{K_GEMM, (int)gemms.size(), 1, tiles(g.M, g.N), 1});
        gemms.push_back(g);
        GemmDesc w{}; w.M = z.OC; w.N = K + 1; w.K = P_ * NOCT; w.ksplit = splits_for(w.K);
        w.A.kind = OPK_CONVGRAD; w.A.space = SP_GRAD; w.A.off = x1.off; w.A.space2 = SP_WS; w.A.off2 = x1.off;
        w.A.oc = z.OC; w.A.kfast = 1;
        w.B.kind = OPK_IM2COL; w.B.space = SP_EXT0; w.B.off = 0; w.B.ones_at = K; w.B.kfast = 0;
        const bool fd = folds_clips() && P_ * NOCT >= 64;      // the conv body's folded loader wants >= one k-tile of rows per clip
        if (fd) fold(w, P_ * NOCT);
        const int64_t stride = (int64_t)z.OC * K + z.OC;
        const int64_t slab = tmp(stride * w.ksplit);
        w.out.kind = OUT_PERMW_SLAB; w.out.space = SP_TMP; w.out.off = slab; w.out.slab_stride = stride; w.out.wcols = K;
        w.out.pb = CONV_K; w.out.pc = NPF; w.out.bias_space = -1;
        op.bwd.push_back(Step{fd ? K_GEMM_FOLD : K_GEMM, (int)gemms.size(), 1, tiles(w.M, w.N), w.ksplit});
        gemms.push_back(w);
        SlabEntry e1{woff, slab, stride, z.OC * K, w.ksplit}, e2{boff, slab + (int64_t)z.OC * K, stride, z.OC, w.ksplit};
        e1.single = e2.single = fd ? 1 : 0;
        slabs[stage_idx(stage)].push_back(e1);
        slabs[stage_idx(stage)].push_back(e2);
        ops.push_back(op);
        return x1;
    }

    // LSTM directions that do not depend on each other: their input projections are ordinary linear
    // ops, the recurrences share ONE launch (blockIdx.y = member), and so do the W_hh-gradient GEMMs.
    struct LstmSpec { T x; int B, S, H, reverse; std::string pre; T out; int coloff; };
    void lstm_group(int stage, const std::vector<LstmSpec>& specs) {
        std::vector<T> zxs;
        for (auto& sp : specs) {
            const std::string sfx = sp.reverse ? "_reverse" : "";
            zxs.push_back(linear(stage, SP_WS, sp.x.off, sp.x.ld, sp.x.rows, sp.x.cols, true, sp.pre + ".weight_ih_l0" + sfx,
                                 sp.pre + ".bias_ih_l0" + sfx, 4 * sp.H, ACT_NONE));
        }
        Op op; op.stage = stage;
        const int first = (int)lstms.size();
        int maxB = 1, maxH = 1, minH = 1 << 30, maxTiles = 1, maxSplit = 1;
        std::vector<GemmDesc> hh;
        for (size_t i = 0; i < specs.size(); ++i) {
            const LstmSpec& sp = specs[i];
            const int H = sp.H;
            if (4 * H > 1024) err = MST_ERR_UNSUPPORTED;
            const std::string sfx = sp.reverse ? "_reverse" : "";
            const int64_t whh = pt.off(sp.pre + ".weight_hh_l0" + sfx), bhh = pt.off(sp.pre + ".bias_hh_l0" + sfx);
            const int64_t n = (int64_t)sp.B * sp.S;
            LstmDesc l{}; l.B = sp.B; l.S = sp.S; l.H = H; l.reverse = sp.reverse; l.zx_off = zxs[i].off; l.whh_off = whh; l.bhh_off = bhh;
            l.out_off = sp.out.off + sp.coloff; l.out_ld = sp.out.ld;
            l.gates_off = tmp(n * 4 * H); l.c_off = tmp(n * H); l.hprev_off = tmp(n * H); l.tc_off = tmp(n * H);
            l.gout_off = sp.out.off + sp.coloff; l.gzx_off = zxs[i].off;
            l.whht_off = H > 64 ? tmp((int64_t)4 * H * H) : 0;
            // the LSTM_NB workgroups of a sequence wait for each other, so every workgroup of the launch must be resident at once:
            // one group per clip, as many as coresident_slots() allows on the device the plan is created on (a partition has fewer
            // CUs); mst_plan_options.lstm_flavour = 1 always takes one workgroup per sequence
            if (opt.lstm_flavour != 1 && H == LSTM_MH && sp.B == 1 && specs.size() == 1 && LSTM_NB * K() <= coresident_slots())
                l.multi = opt.lstm_flavour == 2 ? 2 : 1;
            l.xch_off = l.multi ? tmp(2 * (2 * H + 2 * 4 * H)) : 0;
            l.status_off = status_off;
            lstms.push_back(l);
            if (sp.B > maxB) maxB = sp.B;
            if (H > maxH) maxH = H;
            if (H < minH) minH = H;
            GemmDesc w{}; w.M = 4 * H; w.N = H + 1; w.K = (int)n; w.ksplit = splits_for((int)n);
            w.A.kind = OPK_DENSE; w.A.space = SP_GRAD; w.A.off = zxs[i].off; w.A.si = 1; w.A.sj = 4 * H; w.A.ones_at = -1; w.A.kfast = 0;
            w.B.kind = OPK_DENSE; w.B.space = SP_TMP; w.B.off = l.hprev_off; w.B.si = H; w.B.sj = 1; w.B.ones_at = H; w.B.kfast = 0;
            const bool fd = folds_clips();
            if (fd) fold(w, (int)n, (int)specs.size());
            const int64_t stride = (int64_t)4 * H * H + 4 * H;
            const int64_t slab = tmp(stride * w.ksplit);
            w.out.kind = OUT_SLAB; w.out.space = SP_TMP; w.out.off = slab; w.out.slab_stride = stride; w.out.wcols = H;
            w.out.bias_space = -1;
            hh.push_back(w);
            if (tiles(w.M, w.N) > maxTiles) maxTiles = tiles(w.M, w.N);
            if (w.ksplit > maxSplit) maxSplit = w.ksplit;
            SlabEntry e1{whh, slab, stride, 4 * H * H, w.ksplit}, e2{bhh, slab + (int64_t)4 * H * H, stride, 4 * H, w.ksplit};
            e1.single = e2.single = fd ? 1 : 0;
            slabs[stage_idx(stage)].push_back(e1);
            slabs[stage_idx(stage)].push_back(e2);
        }
        if (maxH > 64 && minH <= 64) err = MST_ERR_UNSUPPORTED;       // one register/L2 flavour per launch
        const int cnt = (int)specs.size();
        if (maxH > 64) op.fwd.push_back(Step{K_LSTM_T, first, cnt, 0, maxH});
        op.fwd.push_back(Step{K_LSTM_F, first, cnt, maxB, maxH});
        op.bwd.push_back(Step{K_LSTM_B, first, cnt, maxB, maxH});
        op.bwd.push_back(Step{folds_clips() ? K_GEMM_FOLD : K_GEMM, (int)gemms.size(), cnt, maxTiles, maxSplit});
        for (auto& w : hh) gemms.push_back(w);
        ops.push_back(op);
    }
    void lstm(int stage, const T& x, int B, int S, int H, int reverse, const std::string& pre, const T& out, int coloff) {
        lstm_group(stage, {LstmSpec{x, B, S, H, reverse, pre, out, coloff}});
    }

    // nn.Linear with K_in <= 8 over a very tall, contiguous input: one lane per row instead of a GEMM of tile padding
    T rowlin(int stage, const T& x, bool xgrad, const std::string& pre, int N, int act, const char* name = nullptr) {
        if (!rowlin_supported(x.cols, N) || x.ld != x.cols) return linear(stage, x, xgrad, pre, N, act, name);
        T out = newT(x.rows, N, name);
        RowLinDesc r{}; r.rows = x.rows; r.kin = x.cols; r.nout = N; r.act = act; r.xgrad = xgrad ? 1 : 0;
        r.x_off = x.off; r.y_off = out.off; r.w_off = pt.off(pre + ".weight"); r.b_off = pt.off(pre + ".bias");
        if (r.b_off != r.w_off + (int64_t)N * x.cols) err = MST_ERR_UNSUPPORTED;
        const int want = (x.rows + 255) / 256;
        const int cap = K() == 1 ? 128 : (1024 / K() < 4 ? 4 : (1024 / K() > 128 ? 128 : 1024 / K()));
        r.nblk = want < cap ? want : cap;
        r.slab_stride = N * x.cols + N; r.slab_off = tmp((int64_t)r.slab_stride * r.nblk);
        Op op; op.stage = stage;
        op.fwd.push_back(Step{K_ROW_F, (int)rowlins.size(), 1, 0, 0});
        op.bwd.push_back(Step{K_ROW_B, (int)rowlins.size(), 1, 0, 0});
        rowlins.push_back(r); ops.push_back(op);
        slabs[stage_idx(stage)].push_back(SlabEntry{r.w_off, r.slab_off, r.slab_stride, r.slab_stride, r.nblk});
        return out;
    }

    // One column block of a Linear whose input is a broadcast-concat (style/utils/pytorch.py:54-65 + nn.Linear):
    //   out[:, col0 : col0 + N] (+)= x (rows x kb) . W[:, k0 : k0 + kb]^T  [+ bias],   no activation.
    // W is the full (N x Kfull) parameter; the bias is the Linear's parameter bias (bname), a one-row activation tensor
    // (bias_act: its gradient is the column sum of dY) or absent.  The concat itself is never formed: the caller adds the
    // blocks' outputs where it consumes them.  Backward: weight-gradient block through a 2-D slab entry, dx += dY . W block.
    // act != ACT_NONE: the block carries the Linear's activation (the weight / input gradients then read dY o act'(Y));
    // row_bias + bias_div: output row m also adds row m / bias_div of `row_bias` (the sum of the Linear's broadcast blocks);
    // its gradient is the sum of dY o act' over each group of bias_div rows.
    void linear_part(int stage, const T& x, bool xgrad, const std::string& wname, int k0, int Kfull, const std::string& bname,
                     const T* bias_act, int N, const T& out, int col0, int act = ACT_NONE, const T* row_bias = nullptr, int bias_div = 0) {
        const int rows = x.rows, kb = x.cols;
        const int64_t woff = pt.off(wname) + k0;
        Op op; op.stage = stage;
        {
            GemmDesc g{}; g.M = rows; g.N = N; g.K = kb; g.ksplit = 1;
            g.A.kind = OPK_DENSE; g.A.space = SP_WS; g.A.off = x.off; g.A.si = x.ld; g.A.sj = 1; g.A.ones_at = -1; g.A.kfast = 1;
            g.B.kind = OPK_DENSE; g.B.space = SP_PAR; g.B.off = woff; g.B.si = 1; g.B.sj = Kfull; g.B.ones_at = -1; g.B.kfast = 1;
            g.out.kind = OUT_STORE; g.out.space = SP_WS; g.out.ldc = out.ld; g.out.act = act; g.out.off = out.off + col0;
            g.out.bias_space = -1;
            if (!bname.empty()) { g.out.bias_space = SP_PAR; g.out.bias_off = pt.off(bname); }
            else if (bias_act) { g.out.bias_space = SP_WS; g.out.bias_off = bias_act->off + col0; }
            else if (row_bias) { g.out.bias_space = SP_WS; g.out.bias_off = row_bias->off; g.out.bias_div = bias_div; g.out.bias_ld = row_bias->ld; }
            if (row_bias && (rows % bias_div || row_bias->rows != rows / bias_div || row_bias->cols != N)) err = MST_ERR_UNSUPPORTED;
            op.fwd.push_back(Step{K_GEMM, (int)gemms.size(), 1, tiles(rows, N), 1});
            gemms.push_back(g);
        }
        const bool pbias = !bname.empty();
        {   // dW block | db  =  dY^T [X | 1]
            GemmDesc w{}; w.M = N; w.N = kb + (pbias ? 1 : 0); w.K = rows; w.ksplit = splits_for(rows);
            if (act != ACT_NONE) {
                w.A.kind = OPK_ACTGRAD; w.A.space = SP_GRAD; w.A.off = out.off + col0; w.A.space2 = SP_WS; w.A.off2 = out.off + col0;
                w.A.ld = out.ld; w.A.act = act; w.A.transposed = 1; w.A.kfast = 0;
            } else { w.A.kind = OPK_DENSE; w.A.space = SP_GRAD; w.A.off = out.off + col0; w.A.si = 1; w.A.sj = out.ld; w.A.ones_at = -1; w.A.kfast = 0; }
            w.B.kind = OPK_DENSE; w.B.space = SP_WS; w.B.off = x.off; w.B.si = x.ld; w.B.sj = 1; w.B.ones_at = pbias ? kb : -1; w.B.kfast = 0;
            const bool fd = folds_clips();
            if (fd) fold(w, rows);
            const int64_t stride = (int64_t)N * kb + N;
            const int64_t slab = tmp(stride * w.ksplit);
            w.out.kind = OUT_SLAB; w.out.space = SP_TMP; w.out.off = slab; w.out.slab_stride = stride; w.out.wcols = kb; w.out.bias_space = -1;
            op.bwd.push_back(Step{fd ? K_GEMM_FOLD : K_GEMM, (int)gemms.size(), 1, tiles(w.M, w.N), w.ksplit});
            gemms.push_back(w);
            SlabEntry e{woff, slab, stride, N * kb, w.ksplit}; e.width = kb; e.dst_ld = Kfull; e.single = fd ? 1 : 0; e.base = pt.off(wname);
            slabs[stage_idx(stage)].push_back(e);
            if (pbias) { SlabEntry eb{pt.off(bname), slab + (int64_t)N * kb, stride, N, w.ksplit}; eb.single = fd ? 1 : 0; slabs[stage_idx(stage)].push_back(eb); }
        }
        if (xgrad) {
            GemmDesc a{}; a.M = rows; a.N = kb; a.K = N; a.ksplit = 1;
            if (act != ACT_NONE) {
                a.A.kind = OPK_ACTGRAD; a.A.space = SP_GRAD; a.A.off = out.off + col0; a.A.space2 = SP_WS; a.A.off2 = out.off + col0;
                a.A.ld = out.ld; a.A.act = act; a.A.transposed = 0; a.A.kfast = 1;
            } else { a.A.kind = OPK_DENSE; a.A.space = SP_GRAD; a.A.off = out.off + col0; a.A.si = out.ld; a.A.sj = 1; a.A.ones_at = -1; a.A.kfast = 1; }
            a.B.kind = OPK_DENSE; a.B.space = SP_PAR; a.B.off = woff; a.B.si = Kfull; a.B.sj = 1; a.B.ones_at = -1; a.B.kfast = 0;
            a.out.kind = OUT_ACCUM; a.out.space = SP_GRAD; a.out.off = x.off; a.out.ldc = x.ld; a.out.bias_space = -1; a.out.act = ACT_NONE;
            op.bwd.push_back(Step{K_GEMM, (int)gemms.size(), 1, tiles(rows, kb), 1});
            gemms.push_back(a);
        }
        if (bias_act) column_sum(op, out.off + col0, out.ld, rows, N, bias_act->off + col0);
        if (row_bias) {          // g_row_bias[j] += sum of (dY o act')[j * bias_div .. + bias_div)
            SegRedDesc r{}; r.src_off = out.off; r.src_ld = out.ld; r.start = col0; r.width = N;
            r.dst_off = row_bias->off; r.dst_ld = row_bias->ld; r.nidx = rows / bias_div;
            const int rs[4] = {rows / bias_div, bias_div, 1, 1};
            for (int q = 0; q < 4; ++q) { r.d[q] = rs[q]; r.s[q] = 0; r.kd[q] = 1; }
            r.s[0] = 1; r.kd[0] = rows / bias_div;
            r.nchunk = (bias_div + 63) / 64;
            int stage2 = 0;
            if (r.nchunk > 1) { r.part_off = tmp((int64_t)r.nidx * r.nchunk * N); stage2 = (r.nidx * N + 255) / 256; }
            r.act = act; r.y_off = out.off;
            op.bwd.push_back(Step{K_SEGRED, (int)segreds.size(), 1, r.nidx * r.nchunk, stage2});
            segreds.push_back(r);
        }
        ops.push_back(op);
    }

    // nn.Linear + activation over cat_with_broadcast(segments) without the concat: the segment that varies along every
    // row-space dim goes through one GEMM whose epilogue adds, per group of rs[3] rows, the SUM of the other segments' blocks
    // (each a small GEMM over its own rows, added up by a broadcast-sum gather); none of the broadcast copies is ever formed.
    //   out[r] = act( x_full[r] W_full^T + sum_i (x_i W_i^T)[index_i(r)] + b )
    // Requires: exactly one full segment (stride pattern of a dense row index), every other segment constant along dim 3.
    T linear_bcast(int stage, const int rs[4], const std::vector<SegIn>& segs, const std::vector<T>& xs, const std::vector<bool>& xg,
                   const std::string& pre, int N, int act, const char* name = nullptr) {
        const int rows = rs[0] * rs[1] * rs[2] * rs[3], mid = rs[0] * rs[1] * rs[2];
        int Kfull = 0, full = -1;
        for (auto& sg : segs) Kfull += sg.width;
        for (size_t i = 0; i < segs.size(); ++i) if (xs[i].rows == rows) full = (int)i;
        if (full < 0) { err = MST_ERR_UNSUPPORTED; return T{}; }
        const std::string wname = pre + ".weight", bname = pre + ".bias";
        std::vector<SegIn> parts;
        int k0 = 0;
        bool bias_done = false;
        for (size_t i = 0; i < segs.size(); ++i) {
            if ((int)i != full) {
                if (segs[i].s[3] != 0) err = MST_ERR_UNSUPPORTED;
                T P = newT(xs[i].rows, N);
                const bool with_bias = !bias_done && xs[i].rows == 1;      // the parameter bias rides on a one-row block
                linear_part(stage, xs[i], xg[i], wname, k0, Kfull, with_bias ? bname : std::string(), nullptr, N, P, 0);
                bias_done |= with_bias;
                SegIn ps = segs[i]; ps.space = SP_WS; ps.off = P.off; ps.ld = P.ld; ps.width = N; ps.grad = true;
                parts.push_back(ps);
            }
            k0 += segs[i].width;
        }
        if (!bias_done) { err = MST_ERR_UNSUPPORTED; return T{}; }
        const int rsm[4] = {rs[0], rs[1], rs[2], 1};
        T S = gather(stage, rsm, parts, true);
        if (S.rows != mid) err = MST_ERR_UNSUPPORTED;
        T out = newT(rows, N, name);
        k0 = 0;
        for (int i = 0; i < full; ++i) k0 += segs[i].width;
        linear_part(stage, xs[full], xg[full], wname, k0, Kfull, std::string(), nullptr, N, out, 0, act, &S, rs[3]);
        return out;
    }

    // backward helper: g[dst_off .. + width) += sum over `rows` rows of g[src_off + r * src_ld .. + width)
    void column_sum(Op& op, int64_t src_off, int src_ld, int rows, int width, int64_t dst_off) {
        SegRedDesc r{}; r.src_off = src_off; r.src_ld = src_ld; r.start = 0; r.width = width; r.dst_off = dst_off; r.dst_ld = width; r.nidx = 1;
        const int rs[4] = {rows, 1, 1, 1};
        for (int q = 0; q < 4; ++q) { r.d[q] = rs[q]; r.s[q] = 0; r.kd[q] = 1; }
        r.nchunk = (rows + 63) / 64;
        int stage2 = 0;
        if (r.nchunk > 1) { r.part_off = tmp((int64_t)r.nchunk * width); stage2 = (width + 255) / 256; }
        op.bwd.push_back(Step{K_SEGRED, (int)segreds.size(), 1, r.nidx * r.nchunk, stage2});
        segreds.push_back(r);
    }

    // backward of "z[c, q] = rt[q] + it[c]" (rows c-major, `width` columns): g_rt[q] += sum_c g_z[c, q], g_it[c] += sum_q g_z[c, q]
    // (with_rt = false: only g_it — the backward note kernel sums over the channels itself and writes g_rt)
    void bcast_add_bwd(int stage, const T& gz, int Cn, int Qn, const T& rt, const T& it, bool with_rt) {
        Op op; op.stage = stage;
        const int first = (int)segreds.size();
        int maxidx = 1, stage2 = 0;
        for (int which = with_rt ? 0 : 1; which < 2; ++which) {
            SegRedDesc r{}; r.src_off = gz.off; r.src_ld = gz.ld; r.start = 0; r.width = gz.cols;
            const T& dst = which ? it : rt;
            r.dst_off = dst.off; r.dst_ld = dst.ld;
            const int rs[4] = {Cn, Qn, 1, 1};
            for (int q = 0; q < 4; ++q) { r.d[q] = rs[q]; r.s[q] = 0; r.kd[q] = 1; }
            if (which) { r.s[0] = 1; r.kd[0] = Cn; r.nidx = Cn; r.nchunk = (Qn + 63) / 64; }      // keep c, reduce q
            else { r.s[1] = 1; r.kd[1] = Qn; r.nidx = Qn; r.nchunk = (Cn + 63) / 64; }            // keep q, reduce c
            if (r.nchunk > 1) {
                r.part_off = tmp((int64_t)r.nidx * r.nchunk * r.width);
                const int b2 = (r.nidx * r.width + 255) / 256;
                if (b2 > stage2) stage2 = b2;
            }
            if (r.nidx * r.nchunk > maxidx) maxidx = r.nidx * r.nchunk;
            segreds.push_back(r);
        }
        op.bwd.push_back(Step{K_SEGRED, first, (int)segreds.size() - first, maxidx, stage2});
        ops.push_back(op);
    }

    // tiled: the exchange of one workspace range over the ranks (all-reduce SUM, done by the host between two phases).
    // forward = the activation range, backward = the same range of the gradient arena (bwd_too), or a scratch range
    void exchange(Op& op, int space, int64_t off, int len, bool bwd_too) {
        op.fwd.push_back(Step{K_XCHG, (int)xchgs.size(), 1, 0, 0});
        xchgs.push_back(Xchg{space, off, len});
        if (space == SP_WS)          // the other ranks' rows must be zero when the SUM runs
            for (int64_t at = off; at < off + len; at += 16384) zero_fwd.push_back(ZeroChunk{at, (int32_t)std::min<int64_t>(16384, off + len - at), 0});
        if (bwd_too) { op.bwd.insert(op.bwd.begin(), Step{K_XCHG, (int)xchgs.size(), 1, 0, 0}); xchgs.push_back(Xchg{SP_GRAD, off, len}); }
    }
    // tiled: this rank's rows of a per-bar tensor into their place in a buffer that holds every rank's rows (the others'
    // rows are zero, so that the exchange's SUM is an all-gather); backward: gather the owned rows' gradient back
    void copy_rows(Op& op, int64_t src_off, int src_sa, int src_sb, int64_t dst_off, int dst_sa, int dst_sb, int na, int nb, int cols) {
        CopyDesc c{}; c.src_off = src_off; c.dst_off = dst_off; c.na = na; c.nb = nb; c.cols = cols;
        c.src_sa = src_sa; c.src_sb = src_sb; c.dst_sa = dst_sa; c.dst_sb = dst_sb;
        op.fwd.insert(op.fwd.begin(), Step{K_COPY_F, (int)copies.size(), 1, 0, 0});
        op.bwd.push_back(Step{K_COPY_B, (int)copies.size(), 1, 0, 0});
        copies.push_back(c);
    }
    // tiled: rank-local partial sums -> `sum` (exchanged) -> back as the only non-zero partial
    void fold_exchange(std::vector<Step>& steps, size_t at, int space, int64_t part_off, int nrows, int row_stride, int ncols,
                       int col_stride) {
        FoldDesc f{}; f.space = space; f.part_off = part_off; f.nrows = nrows; f.row_stride = row_stride; f.ncols = ncols;
        f.col_stride = col_stride; f.sum_off = tmp(ncols);
        const int fi = (int)folds.size();
        folds.push_back(f);
        const int xi = (int)xchgs.size();
        xchgs.push_back(Xchg{SP_TMP, f.sum_off, ncols});
        std::vector<Step> ins = {Step{K_FOLD, fi, 1, 0, 0}, Step{K_XCHG, xi, 1, 0, 0}, Step{K_SPREAD, fi, 1, 0, 0}};
        steps.insert(steps.begin() + at, ins.begin(), ins.end());
    }

    // global = true (tiled plans): the slices hold this rank's rows only, the norms are over every rank's rows
    void combine(int stage, int64_t x_off, int rows, int cols, int ld, int64_t cs, int Cn, const T& out, bool global = false) {
        CombineDesc c{}; c.Cn = Cn; c.rows = rows; c.cols = cols; c.ld = ld; c.x_off = x_off; c.cs = cs; c.out_off = out.off;
        c.stats_off = tmp(64); c.part_off = tmp(COMBINE_MAXBLK * (COMBINE_MAXC + 1));
        c.gx_off = x_off; c.gout_off = out.off;
        if ((int64_t)rows * cols >= ((int64_t)1 << 31)) err = MST_ERR_UNSUPPORTED;      // the kernels index a slice with 32 bits
        int64_t nb = ((int64_t)rows * cols + 1023) / 1024;
        c.nblk = (int)(nb < 1 ? 1 : (nb > COMBINE_MAXBLK ? COMBINE_MAXBLK : nb));
        const int large = (int64_t)rows * cols > COMBINE_SMALL ? 1 : 0;     // Step.b: 1 = needs the two-launch path
        Op op; op.stage = stage;
        if (global && tiled()) {
            const int ci = (int)combines.size();
            op.fwd = {Step{K_COMB_F1, ci, 1, c.nblk, 0}, Step{K_COMB_F2, ci, 1, c.nblk, 0}};
            fold_exchange(op.fwd, 1, SP_TMP, c.part_off, c.nblk, COMBINE_MAXC + 1, Cn, 1);
            op.bwd = {Step{K_COMB_B1, ci, 1, c.nblk, 0}, Step{K_COMB_B2, ci, 1, c.nblk, 0}};
            fold_exchange(op.bwd, 1, SP_TMP, c.part_off, c.nblk, COMBINE_MAXC + 1, Cn + 1, 1);
        } else {
            op.fwd.push_back(Step{K_COMB_F, (int)combines.size(), 1, c.nblk, large});
            op.bwd.push_back(Step{K_COMB_B, (int)combines.size(), 1, c.nblk, large});
        }
        combines.push_back(c);
        ops.push_back(op);
    }

    void build();
    void accesses(const Step& s, std::vector<Acc>& out, bool scheduled = false) const;
    void first_writers(std::vector<Step>& list, size_t begin, bool per_stage, std::vector<Acc>& zero);
    void schedule_pass(const std::vector<Step>& seq, std::vector<Step>& out, bool across_stages);
    bool assign_streams(std::vector<Step>& L);
    int n_signals = 0;                 // event slots assign_streams handed out (<= N_EVENTS)
    void schedule();
    int upload();
};

static const int RS1[4] = {1, 1, 1, 1};

void mst_plan::build() {
    // R = the bars this plan computes per-position work for (its tile when tiled), Rt = the clip's bars: the bar-level chains
    // (bars LSTMs, style encoder, song-info bars LSTM) run replicated over all Rt bars on every rank
    const int C = d.C, R = Rl(), Rt = d.R, r0 = tiled() ? opt.tile_r0 : 0, Tn = d.T, P_ = P(), Q_ = Q();
    const bool TL = tiled();
    const bool U = d.has_unpitched != 0;
    const int E = MST_STAGE_EXTRACT, IN = MST_STAGE_INFO, AP = MST_STAGE_APPLY;
    // ---- inputs / targets (never zeroed, written by the host side)
    T instr = newT(C, z.I, "instr"), mode = newT(1, 2, "mode"), bpm = newT(1, 1, "bpm");
    newT(1, z.NI, "used_instruments"); newT(1, 1, "bpm_target");
    status_off = newT(1, 64, "device_status").off;       // int32 word 0: MST_DEV_* bits (activation arena: never part of a zero list)
    t_losses = newT(1, 64, "losses"); t_saved = newT(1, MST_LOSS_SAVED, "loss_saved"); t_gl = newT(1, 64, "grad_losses");
    loss_scratch = tmp(mst_loss_scratch_floats());
    auto seg0 = [&](const T& t) { return seg(t, 0, 0, 0, 0, true); };      // broadcast over every row
    auto rowsT = [&](const T& t, int rows, int cols, int ld) { return T{t.off, rows, cols, ld}; };
    (void)rowsT;

    // ================================================================= stage 1: extract_style
    stage_begin[0] = act_top;
    std::string m = "pitched_channels_encoder";
    T pce_il = linear(E, instr, false, m + ".instruments_linear", z.PCE_IL, ACT_LEAKY);
    T x1 = conv(E);
    const int rsCQ[4] = {C, R * Tn, 1, 1};
    T pcat = gather(E, rsCQ, {seg(x1, R * Tn, 1, 0, 0), seg(pce_il, 1, 0, 0, 0)});
    T pa = linear(E, pcat, true, m + ".linear", z.H, ACT_LEAKY);
    const std::string mu = "unpitched_channels_encoder";
    T pbeats = newT(P_, z.H, "pitched_beats"), ubeats{}, ua{};
    std::vector<LstmSpec> beat_group = {LstmSpec{pa, C * R, Tn, z.H, 0, m + ".beats_lstm.module", pbeats, 0}};
    if (U) {
        ua = linear(E, SP_EXT1, 0, NF * NUN * NUF, Q_, NF * NUN * NUF, false, mu + ".linear.weight", mu + ".linear.bias",
                    z.H, ACT_LEAKY, nullptr, NUN, NUF);
        ubeats = newT(Q_, z.H, "unpitched_beats");
        beat_group.push_back(LstmSpec{ua, R, Tn, z.H, 0, mu + ".beats_lstm.module", ubeats, 0});
    }
    lstm_group(E, beat_group);
    T plast = newT(Rt, z.H), ulast{};
    T pbl_all{}, ubl_all{};
    if (TL) {
        // last-beat states of every rank's bars, bar-major [(bar, channel), H] | [bar, H]: own rows copied in, one exchange
        pbl_all = newT(Rt * C, z.H);
        if (U) ubl_all = newT(Rt, z.H);
        Op op; op.stage = E;
        exchange(op, SP_WS, pbl_all.off, (int)((U ? ubl_all.off + (int64_t)Rt * z.H : pbl_all.off + (int64_t)Rt * C * z.H) - pbl_all.off), true);
        copy_rows(op, pbeats.off + (int64_t)(Tn - 1) * z.H, Tn * z.H, R * Tn * z.H, pbl_all.off + (int64_t)r0 * C * z.H, C * z.H, z.H, R, C, z.H);
        if (U) copy_rows(op, ubeats.off + (int64_t)(Tn - 1) * z.H, Tn * z.H, 0, ubl_all.off + (int64_t)r0 * z.H, z.H, 0, R, 1, z.H);
        ops.push_back(op);
        combine(E, pbl_all.off, Rt, z.H, C * z.H, z.H, C, plast);
    } else
        combine(E, pbeats.off + (int64_t)(Tn - 1) * z.H, R, z.H, Tn * z.H, (int64_t)R * Tn * z.H, C, plast);
    T pbars = newT(Rt, 2 * z.HB, "pitched_bars"), ubars{};
    std::vector<LstmSpec> bar_group = {LstmSpec{plast, 1, Rt, z.HB, 0, m + ".bars_lstm", pbars, 0},
                                       LstmSpec{plast, 1, Rt, z.HB, 1, m + ".bars_lstm", pbars, z.HB}};
    if (U) {
        ulast = newT(Rt, z.H);
        if (TL) combine(E, ubl_all.off, Rt, z.H, z.H, 0, 1, ulast);
        else combine(E, ubeats.off + (int64_t)(Tn - 1) * z.H, R, z.H, Tn * z.H, 0, 1, ulast);
        ubars = newT(Rt, 2 * z.HB, "unpitched_bars");
        bar_group.push_back(LstmSpec{ulast, 1, Rt, z.HB, 0, mu + ".bars_lstm", ubars, 0});
        bar_group.push_back(LstmSpec{ulast, 1, Rt, z.HB, 1, mu + ".bars_lstm", ubars, z.HB});
    }
    lstm_group(E, bar_group);
    // this rank's bars of the (replicated) bar tensors
    const T pbars_own{pbars.off + (int64_t)r0 * pbars.ld, R, pbars.cols, pbars.ld};
    const T ubars_own = U ? T{ubars.off + (int64_t)r0 * ubars.ld, R, ubars.cols, ubars.ld} : T{};

    m = "pitched_rhythm_encoder";
    T pre_il = linear(E, instr, false, m + ".instruments_linear", z.PRE_IL, ACT_LEAKY);
    T pre_ml = linear(E, mode, false, m + ".mode_linear", z.PRE_ML, ACT_LEAKY);
    T pre_bp = linear(E, bpm, false, m + ".bpm_linear", z.PRE_BPL, ACT_LEAKY);
    T pre_bl = linear(E, pbeats, true, m + ".beats_linear", z.PRE_BL, ACT_LEAKY);
    T pre_br = linear(E, pbars_own, true, m + ".bars_linear", z.PRE_BRL, ACT_LEAKY);
    T pre_cl = linear(E, SP_EXT0, 0, NPN * NPF, P_ * NF, NPN * NPF, false, m + ".channels_linear.weight",
                      m + ".channels_linear.bias", z.PRE_CL, ACT_LEAKY);
    const int rsCRTF[4] = {C, R, Tn, NF};
    // Linear over cat_with_broadcast(beats, bars, channels, instruments, mode, bpm) (style/model.py rhythm encoders) with the
    // concat decomposed away: only the per-fraction `channels` block runs at full row count
    T prh_c = linear_bcast(E, rsCRTF,
                           {seg(pre_bl, R * Tn, Tn, 1, 0), seg(pre_br, 0, 1, 0, 0), seg(pre_cl, R * Tn * NF, Tn * NF, NF, 1),
                            seg(pre_il, 1, 0, 0, 0), seg0(pre_ml), seg0(pre_bp)},
                           {pre_bl, pre_br, pre_cl, pre_il, pre_ml, pre_bp}, {true, true, true, true, true, true},
                           m + ".linear", z.RH, ACT_LEAKY);
    T prh = newT(Q_ * NF, z.RH, "pitched_rhythm");
    combine(E, prh_c.off, Q_ * NF, z.RH, z.RH, (int64_t)Q_ * NF * z.RH, C, prh, true);

    T bars = pbars, rhythm = prh;
    if (U) {
        m = "unpitched_rhythm_encoder";
        T ure_bp = linear(E, bpm, false, m + ".bpm_linear", z.PRE_BPL, ACT_LEAKY);
        T ure_bl = linear(E, ubeats, true, m + ".beats_linear", z.PRE_BL, ACT_LEAKY);
        T ure_br = linear(E, ubars_own, true, m + ".bars_linear", z.PRE_BRL, ACT_LEAKY);
        T ure_cl = linear(E, SP_EXT1, 0, NUN * NUF, Q_ * NF, NUN * NUF, false, m + ".channels_linear.weight",
                          m + ".channels_linear.bias", z.URE_CL, ACT_LEAKY);
        const int rs1RTF[4] = {1, R, Tn, NF};
        T urh_c = linear_bcast(E, rs1RTF, {seg(ure_bl, 0, Tn, 1, 0), seg(ure_br, 0, 1, 0, 0), seg(ure_cl, 0, Tn * NF, NF, 1), seg0(ure_bp)},
                               {ure_bl, ure_br, ure_cl, ure_bp}, {true, true, true, true}, m + ".linear", z.RH, ACT_LEAKY);
        T urh = newT(Q_ * NF, z.RH, "unpitched_rhythm");
        combine(E, urh_c.off, Q_ * NF, z.RH, z.RH, 0, 1, urh, true);
        // combine(pitched, unpitched) stacks the pair on a new leading axis (style/model.py:766-767)
        bars = newT(Rt, z.BAR, "bars");
        combine(E, pbars.off, Rt, z.BAR, z.BAR, ubars.off - pbars.off, 2, bars);
        rhythm = newT(Q_ * NF, z.RH, "rhythm");
        combine(E, prh.off, Q_ * NF, z.RH, z.RH, urh.off - prh.off, 2, rhythm, true);
    } else {
        named["bars"] = bars; named["rhythm"] = rhythm;
    }

    m = "style_encoder";
    T se_il = linear(E, instr, false, m + ".instruments_linear", z.SE_IL, ACT_LEAKY);
    T se_ml = linear(E, mode, false, m + ".mode_linear", z.SE_ML, ACT_LEAKY);
    T se_bp = linear(E, bpm, false, m + ".bpm_linear", z.SE_BL, ACT_LEAKY);
    T sel = newT(Rt, z.SE_L);
    lstm(E, bars, 1, Rt, z.SE_L, 0, m + ".bars_lstm", sel, 0);
    T sel_last{sel.off + (int64_t)(Rt - 1) * z.SE_L, 1, z.SE_L, z.SE_L};
    const int rsC[4] = {C, 1, 1, 1};
    T secat = gather(E, rsC, {seg0(sel_last), seg(se_il, 1, 0, 0, 0), seg0(se_ml), seg0(se_bp)});
    T se_lin = linear(E, secat, true, m + ".linear", z.STYLE, ACT_LEAKY);
    T style = newT(1, z.STYLE, "style");
    combine(E, se_lin.off, 1, z.STYLE, z.STYLE, z.STYLE, C, style);

    m = "melody_encoder";
    T me_il = linear(E, instr, false, m + ".instruments_linear", z.ME_IL, ACT_LEAKY);
    T me_bl = linear(E, pbeats, true, m + ".beats_linear", z.ME_BL, ACT_LEAKY);
    T me_br = linear(E, pbars_own, true, m + ".bars_linear", z.ME_BRL, ACT_LEAKY);
    const int rsCRT[4] = {C, R, Tn, 1};
    T ycat = gather(E, rsCRT, {seg(me_bl, R * Tn, Tn, 1, 0), seg(me_br, 0, 1, 0, 0), seg(me_il, 1, 0, 0, 0)});
    T me_oct = linear(E, ycat, true, m + ".octave_linear", z.MEL * NOCT, ACT_LEAKY);
    T me_deg = linear(E, ycat, true, m + ".scale_degree_linear", z.MEL * NDEG, ACT_LEAKY);
    // note tail + channel combine fused (notes.hip): melody is written directly, the per-channel tensor never exists
    T melody = newT(Q_ * NF * NPN, z.MEL, "melody");
    {
        NotesDesc n{}; n.C = C; n.Q = Q_; n.W = z.MEL; n.CW = z.ME_CW; n.ML = z.PSA_ML;
        n.oct_off = me_oct.off; n.deg_off = me_deg.off; n.x_off = 0; n.x_space = SP_EXT0;
        n.wc_off = pt.off(m + ".channels_linear.weight"); n.bc_off = pt.off(m + ".channels_linear.bias");
        n.wl_off = pt.off(m + ".linear.weight"); n.bl_off = pt.off(m + ".linear.bias");
        n.out_off = melody.off; n.g_out_off = melody.off; n.g_oct_off = me_oct.off; n.g_deg_off = me_deg.off;
        const int nw = z.ME_CW * NPF + z.ME_CW + z.MEL * (z.MEL + z.ME_CW) + z.MEL;
        // reduction passes: waves per channel (one partial each), a wave per position up to 64
        // (the same count for every clip count: the grouping of the partial sums decides the last bits of n_c, and a clip's
        // activations are bit-identical in one-clip and batched plans)
        n.nwc = Q_ < 64 ? Q_ : 64;
        // forward: a wave per (q, group of fractions).  One clip per launch is latency-bound and wants every wave it can get
        // (a wave per fraction: 10 x Q waves); batched plans have the clips for that and keep two waves per q
        n.fhn = K() == 1 ? NF : 2;
        n.part_off = tmp((int64_t)(C + 1) * n.nwc); n.stats_off = tmp(C + 1);
        // backward: two waves per position (one per half of the fractions), two positions per workgroup up to the cap; every
        // wave leaves one slab row for the deferred reduction
        const int me_blk = K() == 1 ? 256 : (2048 / K() < 4 ? 4 : (2048 / K() > 256 ? 256 : 2048 / K()));
        const int want = (P_ + 1) / 2;
        n.nblk = want < me_blk ? want : me_blk; n.slab_stride = nw; n.slab_off = tmp((int64_t)nw * n.nblk);
        Op op; op.stage = E;
        op.fwd.push_back(Step{K_ME_SQ, (int)notes.size(), 1, 0, 0});
        op.fwd.push_back(Step{K_ME_F, (int)notes.size(), 1, 0, 0});
        op.bwd.push_back(Step{K_ME_RED, (int)notes.size(), 1, 0, 0});
        op.bwd.push_back(Step{K_ME_B, (int)notes.size(), 1, 0, 0});
        if (TL) {      // the channel norms are over every rank's positions
            fold_exchange(op.fwd, 1, SP_TMP, n.part_off, n.nwc, 1, C, n.nwc);
            fold_exchange(op.bwd, 1, SP_TMP, n.part_off, n.nwc, 1, C + 1, n.nwc);
        }
        notes.push_back(n); ops.push_back(op);
        // channels_linear.{weight,bias}, linear.{weight,bias} are contiguous in the flat buffer
        slabs[0].push_back(SlabEntry{n.wc_off, n.slab_off, nw, nw, n.nblk});
    }
    stage_end[0] = act_top;

    // ================================================================= stage 2: predict_song_info
    stage_begin[1] = act_top;
    m = "song_info_model";
    T rhy_rows{rhythm.off, Q_, NF * z.RH, NF * z.RH};                    // squash_dims(rhythm, -2)
    T sbl = newT(Q_, z.SIM_BL);
    lstm(IN, rhy_rows, R, Tn, z.SIM_BL, 0, m + ".beats_lstm.module", sbl, 0);
    T slast{sbl.off + (int64_t)(Tn - 1) * z.SIM_BL, R, z.SIM_BL, Tn * z.SIM_BL};
    if (TL) {          // every rank's last-beat rhythm features, then the bars LSTM replicated
        T sl_all = newT(Rt, z.SIM_BL);
        Op op; op.stage = IN;
        exchange(op, SP_WS, sl_all.off, Rt * z.SIM_BL, true);
        copy_rows(op, slast.off, Tn * z.SIM_BL, 0, sl_all.off + (int64_t)r0 * z.SIM_BL, z.SIM_BL, 0, R, 1, z.SIM_BL);
        ops.push_back(op);
        slast = sl_all;
    }
    T sbr = newT(Rt, z.NRF);
    lstm(IN, slast, 1, Rt, z.NRF, 0, m + ".bars_lstm", sbr, 0);
    T feats{sbr.off + (int64_t)(Rt - 1) * z.NRF, 1, z.NRF, z.NRF};
    struct Head { const char* nm; int sw, rw, n, act; const char* out; };
    const Head heads[3] = {{"instruments", z.SIM_SI, z.SIM_RI, z.NI, ACT_NONE, "instruments_pred"},
                           {"mode", z.SIM_SM, z.SIM_RM, 2, ACT_NONE, "mode_pred"},
                           {"bpm", z.SIM_SB, z.SIM_RB, 1, ACT_BPM, "bpm_pred"}};
    for (const Head& h : heads) {
        T hs = linear(IN, style, true, m + ".style_" + h.nm + "_linear", h.sw, ACT_LEAKY);
        T hr = linear(IN, feats, true, m + ".rhythm_" + h.nm + "_linear", h.rw, ACT_LEAKY);
        T hcat = gather(IN, RS1, {seg0(hs), seg0(hr)});
        linear(IN, hcat, true, m + "." + h.nm + "_linear", h.n, h.act, h.out);
    }
    stage_end[1] = act_top;

    // ================================================================= stage 3: apply_style
    stage_begin[2] = act_top;
    m = "pitched_style_applier";
    T psa_sl = linear(AP, style, true, m + ".style_linear", z.PSA_SL, ACT_LEAKY);
    T psa_rl = linear(AP, rhythm, true, m + ".rhythm_linear", z.PSA_RL, ACT_LEAKY);
    T psa_il = linear(AP, instr, false, m + ".instruments_linear", z.PSA_IL, ACT_LEAKY);
    // octave_linear / scale_degree_linear (88 -> 240 / 210) act on cat_with_broadcast([style (1), rhythm (q, f), instrument (c)]):
    // per row that is 39 600 multiply-adds of which all but the rhythm block repeat for every row.  The pre-activations
    // decompose exactly into z[c, qf] = rt[qf] + it[c] (it = instrument block + style block + bias), three small column-block
    // GEMMs each (linear_part); the note kernels add and activate on the fly, and the (positions x 450) octave / degree tensors
    // and the (positions x 88) concat are never written.  The backward note kernel leaves dL/dz per row; its column sums over c
    // and over qf are the gradients of rt and it.
    const int KA = z.PSA_SL + z.PSA_RL + z.PSA_IL, QF_ = Q_ * NF;
    // backward note kernel: one workgroup per qf (a wave per channel pair) up to a cap sized to keep ~15 waves per CU busy over
    // all clips; every workgroup leaves one row of weight-gradient partials and one row of it-gradient partials
    const int psa_np = psa_bwd_waves(C);
    int psa_nblk;
    {
        const int cap = K() == 1 ? 320 : std::max(4, (256 * 15) / (K() * psa_np));
        psa_nblk = std::min(QF_, cap);
    }
    T rt_x[2], it_x[2], itp_x[2];
    for (int which = 0; which < 2; ++which) {
        const std::string lin = m + (which ? ".scale_degree_linear" : ".octave_linear");
        const int Nw = NPF * 6 * (which ? NDEG : NOCT);
        T sb = newT(1, Nw);
        it_x[which] = newT(C, Nw); rt_x[which] = newT(QF_, Nw);
        linear_part(AP, psa_sl, true, lin + ".weight", 0, KA, lin + ".bias", nullptr, Nw, sb, 0);
        linear_part(AP, psa_il, true, lin + ".weight", z.PSA_SL + z.PSA_RL, KA, "", &sb, Nw, it_x[which], 0);
        linear_part(AP, psa_rl, true, lin + ".weight", z.PSA_SL, KA, "", nullptr, Nw, rt_x[which], 0);
        // gradient of it[c] = sum over qf of dL/dz[c, qf]: the note kernel sums its own qf's, this sums the workgroups' rows
        // (only the gradient slot of itp is used)
        itp_x[which] = newT(psa_nblk, C * Nw);
        Op op; op.stage = AP;
        column_sum(op, itp_x[which].off, C * Nw, psa_nblk, C * Nw, it_x[which].off);
        ops.push_back(op);
    }
    // melody_linear (W -> ML, leaky; style/model.py:606-610,660-662) has no op of its own: the note kernels apply it per note
    T xp = newT(P_ * NF * NPN, NPF, "pitched_pred");
    {
        NotesDesc n{}; n.C = C; n.Q = Q_; n.W = z.MEL; n.CW = z.ME_CW; n.ML = z.PSA_ML;
        n.rt_oct_off = rt_x[0].off; n.rt_deg_off = rt_x[1].off; n.it_oct_off = it_x[0].off; n.it_deg_off = it_x[1].off;
        n.mel_off = melody.off; n.g_mel_off = melody.off; n.wm_off = pt.off(m + ".melody_linear.weight");
        n.wl_off = pt.off(m + ".linear.weight"); n.bl_off = pt.off(m + ".linear.bias");
        if (pt.off(m + ".melody_linear.bias") != n.wm_off + (int64_t)z.PSA_ML * z.MEL || n.wl_off != n.wm_off + (int64_t)z.PSA_ML * z.MEL + z.PSA_ML ||
            n.bl_off != n.wl_off + (int64_t)NPF * (NPF * 6 + z.PSA_ML))
            err = MST_ERR_UNSUPPORTED;                        // the slab row below is these four parameters back to back
        n.out_off = xp.off; n.g_out_off = xp.off;
        n.itp_oct_off = itp_x[0].off; n.itp_deg_off = itp_x[1].off;
        n.x_space = SP_EXT0; n.x_off = 0;                     // the target of the fused loss backward: the pitched input itself
        n.loss_saved_off = t_saved.off; n.loss_gl_off = t_gl.off;
        const int nw = z.PSA_ML * z.MEL + z.PSA_ML + NPF * (NPF * 6 + z.PSA_ML) + NPF;
        n.nblk = psa_nblk; n.slab_stride = nw; n.slab_off = tmp((int64_t)nw * n.nblk);
        Op op; op.stage = AP;
        op.fwd.push_back(Step{K_PSA_F, (int)notes.size(), 1, 0, 0});
        op.bwd.push_back(Step{K_PSA_B, (int)notes.size(), 1, 0, 0});
        notes.push_back(n); ops.push_back(op);
        slabs[2].push_back(SlabEntry{n.wm_off, n.slab_off, nw, nw, n.nblk});
    }
    if (U) {
        m = "unpitched_style_applier";
        T usa_sl = linear(AP, style, true, m + ".style_linear", NF * z.USA_SL, ACT_LEAKY);
        T usa_rl = linear(AP, rhythm, true, m + ".rhythm_linear", z.USA_RL, ACT_LEAKY);
        T sl_view{usa_sl.off, NF, z.USA_SL, z.USA_SL};                   // x.view(1, 1, 1, n_beat_fractions, -1)
        const int rsQ_F[4] = {Q_, NF, 1, 1};
        T ucat = gather(AP, rsQ_F, {seg(sl_view, 0, 1, 0, 0), seg(usa_rl, NF, 1, 0, 0)});
        T v = linear(AP, ucat, true, m + ".notes_linear", NUN * NUF * 4, ACT_LEAKY);
        T v_rows{v.off, Q_ * NF * NUN, NUF * 4, NUF * 4};                  // x.view(..., n_unpitched_notes, -1)
        rowlin(AP, v_rows, true, m + ".linear", NUF, ACT_SIGOUT, "unpitched_pred");
    }
    stage_end[2] = act_top;
    if (!notes_widths_supported(z.MEL, z.ME_CW, z.PSA_ML)) err = MST_ERR_UNSUPPORTED;
    if (z.H > 256 || z.SE_L > 256 || z.HB > 256) err = MST_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------ scheduler
// Every launch step's read / write ranges are derived from its descriptors; a step's level is one
// more than the deepest earlier step it conflicts with (RAW / WAR / WAW on overlapping ranges of the
// activation, gradient or scratch arena; parameters and the borrowed note tensors are read-only).
// All steps of one level, stage and kernel are then merged into one launch (blockIdx.y = member).
static void acc_add(std::vector<Acc>& v, int space, int64_t lo, int64_t len, bool w, bool accum = false, bool dense = true) {
    if (space != SP_WS && space != SP_GRAD && space != SP_TMP) return;
    if (len <= 0) return;
    Acc a{space, lo, lo + len, w}; a.accum = accum; a.dense = dense;
    v.push_back(a);
}

static void operand_acc(std::vector<Acc>& v, const Operand& o, int di, int dj, int ones) {
    switch (o.kind) {
    case OPK_DENSE: {
        const int64_t si = o.si < 0 ? -o.si : o.si, sj = o.sj < 0 ? -o.sj : o.sj;
        const int dj_eff = ones >= 0 ? dj - 1 : dj;
        acc_add(v, o.space, o.off, (int64_t)(di - 1) * si + (int64_t)(dj_eff > 0 ? dj_eff - 1 : 0) * sj + 1, false);
        break;
    }
    case OPK_ACTGRAD: {
        const int rows = o.transposed ? dj : di;
        acc_add(v, o.space, o.off, (int64_t)rows * o.ld, false);
        acc_add(v, o.space2, o.off2, (int64_t)rows * o.ld, false);
        break;
    }
    case OPK_CONVGRAD: {
        const int64_t n = (int64_t)(dj / NOCT) * o.oc * NOCT;
        acc_add(v, o.space, o.off, n, false);
        acc_add(v, o.space2, o.off2, n, false);
        break;
    }
    default: break;      // IM2COL / PERMW read the borrowed inputs / parameters
    }
}

void mst_plan::accesses(const Step& s, std::vector<Acc>& v, bool scheduled) const {
    // scheduled: s.first indexes the scheduled (per-clip relocated) descriptor arrays; clip 0's copies equal the originals
    const std::vector<GemmDesc>& gemms = scheduled ? s_gemms : this->gemms;
    const std::vector<GatherDesc>& gathers = scheduled ? s_gathers : this->gathers;
    const std::vector<SegRedDesc>& segreds = scheduled ? s_segreds : this->segreds;
    const std::vector<LstmDesc>& lstms = scheduled ? s_lstms : this->lstms;
    const std::vector<CombineDesc>& combines = scheduled ? s_combines : this->combines;
    const std::vector<NotesDesc>& notes = scheduled ? s_notes : this->notes;
    const std::vector<RowLinDesc>& rowlins = scheduled ? s_rowlins : this->rowlins;
    for (int i = 0; i < s.count; ++i) {
        switch (s.kind) {
        case K_GEMM: case K_GEMM_FOLD: {
            const GemmDesc& g = gemms[s.first + i];
            const int gk = g.fold_rows ? g.fold_rows : g.K;      // a folded reduction touches, per clip, what one clip's would
            const int gm = g.clip_rows ? 1 : g.M;                // ... and so do clips-as-rows (its scheduled copy has M = clips)
            Operand ga = g.A;
            if (g.clip_rows && ga.kind == OPK_ACTGRAD) ga.ld = gk;       // one row of gk columns (its scheduled pitch is the clip stride)
            operand_acc(v, ga, gm, gk, -1);
            operand_acc(v, g.B, gk, g.N, g.B.ones_at);
            const OutSpec& o = g.out;
            if (o.bias_space >= 0)       // a bias row that is an activation (linear_part); bias_div: one row per bias_div output rows
                acc_add(v, o.bias_space, o.bias_off, o.bias_div > 0 ? (int64_t)((g.M - 1) / o.bias_div) * o.bias_ld + g.N : g.N, false);
            if (o.kind == OUT_STORE) acc_add(v, o.space, o.off, (int64_t)(gm - 1) * o.ldc + g.N, true);
            else if (o.kind == OUT_ACCUM) acc_add(v, o.space, o.off, (int64_t)(gm - 1) * o.ldc + g.N, true, true, g.N == o.ldc || gm == 1);
            else if (o.kind == OUT_CONV) acc_add(v, o.space, o.off, (int64_t)(g.M / NOCT) * o.ldc, true);
            else acc_add(v, o.space, o.off, o.slab_stride * g.ksplit, true);
            break;
        }
        case K_LIN_F: case K_LIN_A: case K_LIN_W: {
            if (i > 0) break;
            const LinDesc& l = lins[s.first];
            const int64_t nx = (int64_t)(l.rows - 1) * l.x_ld + l.K, ny = (int64_t)(l.rows - 1) * l.y_ld + l.N;
            if (s.kind == K_LIN_F) { acc_add(v, l.x_space, l.x_off, nx, false); acc_add(v, SP_WS, l.y_off, ny, true); }
            else if (s.kind == K_LIN_W) {
                acc_add(v, l.x_space, l.x_off, nx, false); acc_add(v, SP_WS, l.y_off, ny, false); acc_add(v, SP_GRAD, l.y_off, ny, false);
                acc_add(v, SP_TMP, l.slab_off, l.slab_stride * l.splits, true);
            } else {
                acc_add(v, SP_WS, l.y_off, ny, false); acc_add(v, SP_GRAD, l.y_off, ny, false);
                acc_add(v, SP_GRAD, l.x_off, nx, true, true, l.x_ld == l.K || l.rows == 1);
            }
            break;
        }
        case K_CONV_P: case K_CONV_F: case K_CONV_W: {
            if (i > 0) break;
            const ConvDesc& c = convs[s.first];
            const int64_t n = (int64_t)c.P * c.OC * NOCT;
            if (s.kind == K_CONV_P) acc_add(v, SP_TMP, c.wp_off, (int64_t)NF * 72 * 64, true);
            else if (s.kind == K_CONV_F) { acc_add(v, SP_TMP, c.wp_off, (int64_t)NF * 72 * 64, false); acc_add(v, SP_WS, c.x1_off, n, true); }
            else { acc_add(v, SP_WS, c.x1_off, n, false); acc_add(v, SP_GRAD, c.x1_off, n, false); acc_add(v, SP_TMP, c.slab_off, c.slab_stride * c.splits, true); }
            break;
        }
        case K_GATHER: {
            const GatherDesc& g = gathers[s.first + i];
            for (int q = 0; q < g.nseg; ++q) {
                const Seg& sg = g.seg[q];
                int64_t maxrow = 0;
                for (int k = 0; k < 4; ++k) maxrow += (int64_t)(g.d[k] - 1) * sg.s[k];
                acc_add(v, sg.space, sg.off, maxrow * sg.ld + sg.width, false);
            }
            acc_add(v, SP_WS, g.out_off, (int64_t)g.rows * g.K, true);
            break;
        }
        case K_SEGRED: {
            const SegRedDesc& r = segreds[s.first + i];
            const int64_t rows = (int64_t)r.d[0] * r.d[1] * r.d[2] * r.d[3];
            acc_add(v, SP_GRAD, r.src_off, rows * r.src_ld, false);
            if (r.act != ACT_NONE) acc_add(v, SP_WS, r.y_off, rows * r.src_ld, false);
            acc_add(v, SP_GRAD, r.dst_off, (int64_t)(r.nidx - 1) * r.dst_ld + r.width, true, true, r.width == r.dst_ld || r.nidx == 1);
            if (r.nchunk > 1) acc_add(v, SP_TMP, r.part_off, (int64_t)r.nidx * r.nchunk * r.width, true);
            break;
        }
        case K_LSTM_T: {
            const LstmDesc& l = lstms[s.first + i];
            if (l.multi) { acc_add(v, SP_TMP, l.xch_off, 2 * (2 * l.H + 8 * l.H), true); break; }
            if (l.H > 64) acc_add(v, SP_TMP, l.whht_off, (int64_t)4 * l.H * l.H, true);
            break;
        }
        case K_LSTM_F: {
            const LstmDesc& l = lstms[s.first + i];
            const int64_t n = (int64_t)l.B * l.S;
            acc_add(v, SP_WS, l.zx_off, n * 4 * l.H, false);
            acc_add(v, SP_WS, l.out_off, (n - 1) * l.out_ld + l.H, true);
            acc_add(v, SP_TMP, l.gates_off, n * 4 * l.H, true);
            acc_add(v, SP_TMP, l.c_off, n * l.H, true);
            acc_add(v, SP_TMP, l.hprev_off, n * l.H, true);
            acc_add(v, SP_TMP, l.tc_off, n * l.H, true);
            if (l.multi) acc_add(v, SP_TMP, l.xch_off, 2 * (2 * l.H + 8 * l.H), true);
            else if (l.H > 64) acc_add(v, SP_TMP, l.whht_off, (int64_t)4 * l.H * l.H, false);
            break;
        }
        case K_LSTM_B: {
            const LstmDesc& l = lstms[s.first + i];
            const int64_t n = (int64_t)l.B * l.S;
            acc_add(v, SP_TMP, l.gates_off, n * 4 * l.H, false);
            acc_add(v, SP_TMP, l.c_off, n * l.H, false);
            acc_add(v, SP_TMP, l.tc_off, n * l.H, false);
            acc_add(v, SP_GRAD, l.gout_off, (n - 1) * l.out_ld + l.H, false);
            acc_add(v, SP_GRAD, l.gzx_off, n * 4 * l.H, true);
            if (l.multi) acc_add(v, SP_TMP, l.xch_off, 2 * (2 * l.H + 8 * l.H), true);
            break;
        }
        case K_COMB_F: case K_COMB_B: {
            const CombineDesc& c = combines[s.first + i];
            const int64_t span = (int64_t)(c.Cn - 1) * (c.cs < 0 ? -c.cs : c.cs) + (int64_t)(c.rows - 1) * c.ld + c.cols;
            const int64_t lo = c.cs < 0 ? c.x_off + (int64_t)(c.Cn - 1) * c.cs : c.x_off;
            const int64_t n = (int64_t)c.rows * c.cols;
            acc_add(v, SP_WS, lo, span, false);
            acc_add(v, SP_TMP, c.part_off, COMBINE_MAXBLK * (COMBINE_MAXC + 1), true);
            if (s.kind == K_COMB_F) {
                acc_add(v, SP_WS, c.out_off, n, true);
                acc_add(v, SP_TMP, c.stats_off, 64, true);
            } else {
                acc_add(v, SP_WS, c.out_off, n, false);
                acc_add(v, SP_TMP, c.stats_off, 64, false);
                acc_add(v, SP_GRAD, c.gout_off, n, false);
                acc_add(v, SP_GRAD, lo - c.x_off + c.gx_off, span, true, true, span == (int64_t)c.Cn * n);
            }
            break;
        }
        case K_COMB_F1: case K_COMB_F2: case K_COMB_B1: case K_COMB_B2: {
            const CombineDesc& c = combines[s.first + i];
            const int64_t span = (int64_t)(c.Cn - 1) * (c.cs < 0 ? -c.cs : c.cs) + (int64_t)(c.rows - 1) * c.ld + c.cols;
            const int64_t lo = c.cs < 0 ? c.x_off + (int64_t)(c.Cn - 1) * c.cs : c.x_off;
            const int64_t n = (int64_t)c.rows * c.cols;
            acc_add(v, SP_WS, lo, span, false);
            const bool part_w = s.kind == K_COMB_F1 || s.kind == K_COMB_B1;
            acc_add(v, SP_TMP, c.part_off, COMBINE_MAXBLK * (COMBINE_MAXC + 1), part_w);
            if (s.kind == K_COMB_F2) { acc_add(v, SP_WS, c.out_off, n, true); acc_add(v, SP_TMP, c.stats_off, 64, true); }
            if (s.kind == K_COMB_B1 || s.kind == K_COMB_B2) {
                acc_add(v, SP_WS, c.out_off, n, false);
                acc_add(v, SP_GRAD, c.gout_off, n, false);
            }
            if (s.kind == K_COMB_B2) {
                acc_add(v, SP_TMP, c.stats_off, 64, false);
                acc_add(v, SP_GRAD, lo - c.x_off + c.gx_off, span, true, true, span == (int64_t)c.Cn * n);
            }
            break;
        }
        case K_COPY_F: case K_COPY_B: {
            const CopyDesc& c = copies[s.first + i];
            const int64_t ss = (int64_t)(c.na - 1) * c.src_sa + (int64_t)(c.nb - 1) * c.src_sb + c.cols;
            const int64_t ds = (int64_t)(c.na - 1) * c.dst_sa + (int64_t)(c.nb - 1) * c.dst_sb + c.cols;
            if (s.kind == K_COPY_F) { acc_add(v, SP_WS, c.src_off, ss, false); acc_add(v, SP_WS, c.dst_off, ds, true); }
            else { acc_add(v, SP_GRAD, c.dst_off, ds, false); acc_add(v, SP_GRAD, c.src_off, ss, true, true, false); }
            break;
        }
        case K_FOLD: case K_SPREAD: {
            const FoldDesc& f = folds[s.first + i];
            const int64_t span = (int64_t)(f.nrows - 1) * f.row_stride + (int64_t)(f.ncols - 1) * f.col_stride + 1;
            acc_add(v, f.space, f.part_off, span, s.kind == K_SPREAD);
            acc_add(v, SP_TMP, f.sum_off, f.ncols, s.kind == K_FOLD);
            break;
        }
        case K_XCHG: {
            const Xchg& x = xchgs[s.first + i];
            acc_add(v, x.space, x.off, x.len, true);
            break;
        }
        case K_ROW_F: case K_ROW_B: {
            const RowLinDesc& r = rowlins[s.first + i];
            const int64_t nx = (int64_t)r.rows * r.kin, ny = (int64_t)r.rows * r.nout;
            acc_add(v, SP_WS, r.x_off, nx, false);
            acc_add(v, SP_WS, r.y_off, ny, s.kind == K_ROW_F);
            if (s.kind == K_ROW_B) {
                acc_add(v, SP_GRAD, r.y_off, ny, false);
                if (r.xgrad) acc_add(v, SP_GRAD, r.x_off, nx, true, true, true);
                acc_add(v, SP_TMP, r.slab_off, (int64_t)r.slab_stride * r.nblk, true);
            }
            break;
        }
        case K_ME_SQ: case K_ME_F: case K_ME_RED: case K_ME_B: {
            const NotesDesc& n = notes[s.first + i];
            const int64_t rows = (int64_t)n.C * n.Q, mel = (int64_t)n.Q * NF * NPN * n.W;
            const int64_t nparts = (int64_t)(n.C + 1) * n.nwc;
            acc_add(v, SP_WS, n.oct_off, rows * NOCT * n.W, false);
            acc_add(v, SP_WS, n.deg_off, rows * NDEG * n.W, false);
            if (s.kind == K_ME_SQ) acc_add(v, SP_TMP, n.part_off, nparts, true);
            else if (s.kind == K_ME_F) {
                acc_add(v, SP_TMP, n.part_off, nparts, false);
                acc_add(v, SP_TMP, n.stats_off, n.C + 1, true);
                acc_add(v, SP_WS, n.out_off, mel, true);
            } else if (s.kind == K_ME_RED) {
                acc_add(v, SP_WS, n.out_off, mel, false);
                acc_add(v, SP_GRAD, n.g_out_off, mel, false);
                acc_add(v, SP_TMP, n.part_off, nparts, true);
            } else {
                acc_add(v, SP_TMP, n.part_off, nparts, false);
                acc_add(v, SP_TMP, n.stats_off, n.C + 1, false);
                acc_add(v, SP_GRAD, n.g_out_off, mel, false);
                acc_add(v, SP_GRAD, n.g_oct_off, rows * NOCT * n.W, true);
                acc_add(v, SP_GRAD, n.g_deg_off, rows * NDEG * n.W, true);
                acc_add(v, SP_TMP, n.slab_off, (int64_t)n.slab_stride * n.nblk, true);
            }
            break;
        }
        case K_PSA_F: case K_PSA_B: {
            const NotesDesc& n = notes[s.first + i];
            const bool me = false, bwd = s.kind == K_PSA_B;
            const int64_t pos = (int64_t)n.C * n.Q * NF * NPN;
            const int64_t rows = me ? (int64_t)n.C * n.Q : (int64_t)n.C * n.Q * NF;
            const int ow = me ? NOCT * n.W : NOCT * 30, dw = me ? NDEG * n.W : NDEG * 30, outw = me ? n.W : NPF;
            const int64_t mln = (int64_t)n.Q * NF * NPN * n.W;            // melody rows (melody_linear is applied inside the kernels)
            acc_add(v, SP_WS, n.rt_oct_off, (int64_t)n.Q * NF * ow, false);
            acc_add(v, SP_WS, n.rt_deg_off, (int64_t)n.Q * NF * dw, false);
            acc_add(v, SP_WS, n.it_oct_off, (int64_t)n.C * ow, false);
            acc_add(v, SP_WS, n.it_deg_off, (int64_t)n.C * dw, false);
            if (!me) acc_add(v, SP_WS, n.mel_off, mln, false);
            acc_add(v, SP_WS, n.out_off, pos * outw, !bwd);
            if (bwd) {
                // (under MST_BF_LOSS_FUSED the upstream gradient is not read; declaring the read keeps the generic path safe)
                acc_add(v, SP_GRAD, n.g_out_off, pos * outw, false);
                acc_add(v, SP_WS, n.loss_saved_off, MST_LOSS_SAVED, false);
                acc_add(v, SP_WS, n.loss_gl_off, 64, false);
                acc_add(v, SP_GRAD, n.itp_oct_off, (int64_t)n.nblk * n.C * ow, true);      // per-workgroup partial gradients of it
                acc_add(v, SP_GRAD, n.itp_deg_off, (int64_t)n.nblk * n.C * dw, true);
                acc_add(v, SP_GRAD, n.rt_oct_off, (int64_t)n.Q * NF * ow, true);       // channel sums of dL/dz = gradient of rt
                acc_add(v, SP_GRAD, n.rt_deg_off, (int64_t)n.Q * NF * dw, true);
                if (!me) acc_add(v, SP_GRAD, n.g_mel_off, mln, true);
                acc_add(v, SP_TMP, n.slab_off, (int64_t)n.slab_stride * n.nblk, true);
            }
            break;
        }
        }
    }
}

static bool conflicts(const std::vector<Acc>& a, const std::vector<Acc>& b) {
    for (const Acc& x : a)
        for (const Acc& y : b)
            if (x.space == y.space && (x.w || y.w) && x.lo < y.hi && y.lo < x.hi) return true;
    return false;
}

void mst_plan::schedule_pass(const std::vector<Step>& seq, std::vector<Step>& out, bool across_stages) {
    const int n = (int)seq.size();
    std::vector<std::vector<Acc>> acc(n);
    std::vector<int> level(n, 0);
    int maxlevel = 0;
    for (int i = 0; i < n; ++i) {
        accesses(seq[i], acc[i]);
        for (int j = 0; j < i; ++j)
            if (level[j] >= level[i] && conflicts(acc[i], acc[j])) level[i] = level[j] + 1;
        if (level[i] > maxlevel) maxlevel = level[i];
    }
    std::vector<char> done(n, 0);
    for (int lv = 0; lv <= maxlevel; ++lv) {
        for (int i = 0; i < n; ++i) {
            if (done[i] || level[i] != lv) continue;
            const Step& s0 = seq[i];
            const bool mergeable = !opt.no_merge && (s0.kind == K_GEMM || s0.kind == K_GEMM_FOLD || s0.kind == K_GATHER || s0.kind == K_SEGRED || s0.kind == K_LSTM_T ||
                                   s0.kind == K_LSTM_F || s0.kind == K_LSTM_B || s0.kind == K_COMB_F || s0.kind == K_COMB_B);
            Step m = s0; m.count = 0; m.lvl = lv;
            const bool is_lstm = s0.kind == K_LSTM_T || s0.kind == K_LSTM_F || s0.kind == K_LSTM_B;
            const bool is_comb = s0.kind == K_COMB_F || s0.kind == K_COMB_B || (s0.kind >= K_COMB_F1 && s0.kind <= K_COMB_B2);
            const bool is_notes = s0.kind == K_ME_F || s0.kind == K_ME_B || s0.kind == K_PSA_F || s0.kind == K_PSA_B || s0.kind == K_ME_SQ || s0.kind == K_ME_RED;
            if (s0.kind == K_GEMM || s0.kind == K_GEMM_FOLD) m.first = (int)s_gemms.size();
            else if (s0.kind == K_GATHER) m.first = (int)s_gathers.size();
            else if (s0.kind == K_SEGRED) m.first = (int)s_segreds.size();
            else if (is_lstm) m.first = (int)s_lstms.size();
            else if (is_comb) m.first = (int)s_combines.size();
            else if (is_notes) m.first = (int)s_notes.size();
            else if (s0.kind == K_ROW_F || s0.kind == K_ROW_B) m.first = (int)s_rowlins.size();
            std::vector<int> members;                       // indices into the per-kind descriptor vectors (one clip)
            for (int j = i; j < n; ++j) {
                if (done[j] || level[j] != lv) continue;
                const Step& s = seq[j];
                if (s.kind != s0.kind || (!across_stages && s.stage != s0.stage)) continue;
                // LSTM launches come in a register-resident (H <= 64) and an L2 flavour
                if (is_lstm && ((s.b > 64) != (s0.b > 64))) continue;
                // ... and the multi-workgroup flavour is a kernel of its own (grid and hidden size fixed): never merged with another
                if (is_lstm && j != i && (lstms[s.first].multi || lstms[s0.first].multi)) continue;
                if (j != i && !mergeable) continue;
                done[j] = 1;
                for (int q = 0; q < s.count; ++q) members.push_back(s.first + q);
                m.stage |= s.stage;
                if (s.a > m.a) m.a = s.a;
                if (s.b > m.b) m.b = s.b;
            }
            // clip-major replication: clip k's copy of every member, relocated to clip k's workspace slices.
            // The W_hh transpose reads parameters only, so it runs once for all clips.
            // (the multi-workgroup flavour's K_LSTM_T step clears per-clip exchange tags instead: one copy per clip)
            const bool shared_T = s0.kind == K_LSTM_T && !lstms[s0.first].multi;
            const int copies = (shared_T || s0.kind == K_GEMM_FOLD || s0.kind >= K_CONV_P) ? 1 : K();
            for (int k = 0; k < copies; ++k) {
                for (int idx : members) {
                    if (s0.kind == K_GEMM || s0.kind == K_GEMM_FOLD) {
                        GemmDesc g = reloc(gemms[idx], k); g.variant = gemm_variant(g);
                        if (g.fold_rows) {          // arena sizes are final now: clip strides of the folded reduction
                            g.acs = shift(g.A.space, 1); g.acs2 = (g.A.kind == OPK_ACTGRAD || g.A.kind == OPK_CONVGRAD) ? shift(g.A.space2, 1) : 0; g.bcs = shift(g.B.space, 1);
                            // the loader adds clip * stride as a 32-bit element offset
                            if ((uint64_t)std::max(g.acs, std::max(g.acs2, g.bcs)) * (uint64_t)K() > 0xFFFFFFFFull) err = MST_ERR_UNSUPPORTED;
                        }
                        if (g.clip_rows) {          // the clips become the rows: row stride = the clip stride of the operand's arena
                            g.M = K();
                            if (g.A.kind == OPK_DENSE) g.A.si = shift(g.A.space, 1);
                            else g.A.ld = (int32_t)shift(g.A.space, 1);              // OPK_ACTGRAD: dY and Y share the stride
                            g.out.ldc = (int32_t)shift(g.out.space, 1);
                            if ((uint64_t)act_top * (uint64_t)K() > 0x7FFFFFFFull) err = MST_ERR_UNSUPPORTED;     // 32-bit row offsets
                        }
                        const int kr = (g.K + g.ksplit - 1) / g.ksplit;
                        g.kdsel = kr <= 32 ? 0 : (kr <= 64 ? 1 : 2);
                        // tiles per workgroup (64x64 tiling): measured on MI355X at 64 clips per launch, runs of 2 / 4 / 8 tiles
                        // lost 7 / 22 / 33 % against one tile per workgroup (fewer, longer workgroups: worse balance over the
                        // 256 CUs than the saved per-workgroup latency buys), so the default stays 1; the option remains for experiments
                        g.run = (mfma && opt.gemm_run > 1) ? opt.gemm_run : 1;
                        if (g.variant < 0) err = MST_ERR_UNSUPPORTED;
                        s_gemms.push_back(g);
                    }
                    else if (s0.kind == K_GATHER) s_gathers.push_back(reloc(gathers[idx], k));
                    else if (s0.kind == K_SEGRED) s_segreds.push_back(reloc(segreds[idx], k));
                    else if (is_lstm) s_lstms.push_back(reloc(lstms[idx], k));
                    else if (is_comb) s_combines.push_back(reloc(combines[idx], k));
                    else if (is_notes) s_notes.push_back(reloc(notes[idx], k));
                    else if (s0.kind == K_ROW_F || s0.kind == K_ROW_B) s_rowlins.push_back(reloc(rowlins[idx], k));
                }
            }
            m.count = (int)members.size() * copies;
            if (m.kind == K_GEMM || m.kind == K_GEMM_FOLD) {      // flat grid: a clip's block range concatenates its members' (tile, k-split) workgroups
                const int nm = (int)members.size();
                int total = 0;
                for (int q = 0; q < nm; ++q) {
                    const GemmDesc& g0 = s_gemms[m.first + q];
                    for (int k = 0; k < copies; ++k) s_gemms[m.first + k * nm + q].blk_begin = total;
                    total += gemm_blocks(g0, mfma);
                }
                m.a = total; m.b = nm;        // blocks per clip, members per clip
                m.c = (int)s_gemm_owner.size();
                for (int q = 0; q < nm; ++q) {
                    const int nb = gemm_blocks(s_gemms[m.first + q], mfma);
                    for (int i = 0; i < nb; ++i) s_gemm_owner.push_back(q);
                }
            }
            if (m.kind == K_SEGRED) {    // same flat layout; m.b keeps the stage-2 block count, members = count / clips
                const int nm = (int)members.size();
                int total = 0;
                for (int q = 0; q < nm; ++q) {
                    const SegRedDesc& r0 = s_segreds[m.first + q];
                    for (int k = 0; k < copies; ++k) s_segreds[m.first + k * nm + q].blk_begin = total;
                    total += (r0.nidx * r0.nchunk * r0.width + 255) / 256;
                }
                m.a = total;
            }
            out.push_back(m);
        }
    }
}

// Gradient slots without a memset.  Walk a scheduled backward list in launch order and decide, for every writer that can
// accumulate (input-gradient GEMMs, segment reduces, combine backward, the row-wise Linear's dx), whether it is the FIRST
// writer of a dense, so far untouched range of the gradient arena — then it stores (=) and nothing has to be cleared — or
// lands on data written earlier in the pass — then it accumulates (+=).  Whatever is read, or accumulated into, without
// having been written before (rows of an LSTM output nobody consumes, strided / partially covered targets) goes on the
// zero list: the only ranges mst_zero_grads still clears.  (It was one memset of the whole arena: ~1 GB of writes per
// 64-clip pass, more than the algorithmic bytes of the pass.)
static bool covered(const std::vector<Acc>& have, int64_t lo, int64_t hi) {
    // is [lo, hi) inside the union of the ranges in `have`?  (few dozen ranges: quadratic sweep is fine)
    int64_t at = lo;
    bool moved = true;
    while (at < hi && moved) {
        moved = false;
        for (const Acc& a : have)
            if (a.lo <= at && a.hi > at) { at = a.hi; moved = true; }
    }
    return at >= hi;
}
static bool touches(const std::vector<Acc>& have, int64_t lo, int64_t hi) {
    for (const Acc& a : have) if (a.lo < hi && lo < a.hi) return true;
    return false;
}

void mst_plan::first_writers(std::vector<Step>& list, size_t begin, bool per_stage, std::vector<Acc>& zero) {
    std::vector<Acc> have;                               // ranges holding defined data at this point of the pass
    auto external = [&](const char* name) {              // written by the caller / the loss before the pass starts
        auto it = named.find(name);
        if (it != named.end()) have.push_back(Acc{SP_GRAD, it->second.off, it->second.off + MST_ALIGN_UP((int64_t)it->second.rows * it->second.cols), true});
    };
    for (const char* nm : {"pitched_pred", "unpitched_pred", "instruments_pred", "mode_pred", "bpm_pred"}) external(nm);
    if (per_stage) for (const char* nm : {"style", "melody", "rhythm"}) external(nm);      // seeded or cleared by the caller
    const int copies = K();
    for (size_t si = begin; si < list.size(); ++si) {
        Step& m = list[si];
        const bool plain = m.kind >= K_COPY_F && m.kind != K_GEMM_FOLD;   // indexes an unscheduled (one-clip) descriptor vector
        const int nm = (m.kind == K_GEMM || m.kind == K_GEMM_FOLD) ? m.b : (plain ? m.count : m.count / copies);
        for (int q = 0; q < nm; ++q) {
            std::vector<Acc> acc;
            Step one = m; one.first = m.first + q; one.count = 1;
            accesses(one, acc, !plain);
            for (const Acc& a : acc) {
                if (a.space != SP_GRAD || a.w) continue;
                if (!covered(have, a.lo, a.hi)) { Acc z = a; z.w = true; z.space = m.stage; zero.push_back(z); have.push_back(a); }
            }
            int first = 0;
            for (const Acc& a : acc) {
                if (a.space != SP_GRAD || !a.w) continue;
                if (!a.accum) { if (a.dense) have.push_back(a); continue; }      // plain store of a dense range
                if (a.dense && !touches(have, a.lo, a.hi)) { first = 1; have.push_back(a); }
                else if (!covered(have, a.lo, a.hi)) { Acc z = a; z.space = m.stage; zero.push_back(z); have.push_back(a); }
            }
            if (m.kind == K_COPY_B) this->copies[m.first + q].first = first;
            if (m.kind == K_LIN_A) this->lins[m.first + q].first[per_stage ? 0 : 1] = first;
            for (int k = 0; k < copies && !plain; ++k) {
                const int idx = m.first + k * nm + q;
                if (m.kind == K_GEMM) s_gemms[idx].out.first = first;
                else if (m.kind == K_GEMM_FOLD) { if (k == 0) s_gemms[m.first + q].out.first = first; }      // one descriptor for all clips (clips-as-rows dX)
                else if (m.kind == K_SEGRED) s_segreds[idx].first = first;
                else if (m.kind == K_COMB_B || m.kind == K_COMB_B2) s_combines[idx].first = first;
                else if (m.kind == K_ROW_B) s_rowlins[idx].first = first;
            }
        }
    }
}

// Spread the launches of a scheduled whole-model list over 1 + N_SIDE streams along its dependency DAG (list scheduling): a
// launch depends on every earlier launch it shares memory with (any overlap with a write on either side: conflicts()).  In list
// order, each launch goes to the stream where it can start first under a simple cost model (a launch on the stream of the
// predecessor it waits for longest starts right behind it; a dependency that crosses streams costs an event wait, measured
// ~5 us per edge inside a replayed hipGraph against ~2 us for a same-stream boundary — so the model is reluctant to cross).
// Cross-stream dependencies become (signal, wait) pairs: the producer records an event slot behind it, the consumer's stream
// waits for it; waits already implied by stream order or by an earlier wait are dropped (vector clocks).  The result is checked:
// every conflicting pair must be ordered by stream order and waits, else the plan keeps one stream.
bool mst_plan::assign_streams(std::vector<Step>& L) {
    const int n = (int)L.size(), NS = 1 + N_SIDE;
    std::vector<std::vector<Acc>> acc(n);
    for (int i = 0; i < n; ++i) accesses(L[i], acc[i], true);
    auto dur = [&](const Step& st) {           // rough one-clip durations in us (profiles/: pass_timeline_one_clip)
        switch (st.kind) {
        case K_LSTM_F: case K_LSTM_B: return s_lstms[st.first].multi ? 30.0 : 15.0;
        case K_ME_F: case K_ME_B: case K_PSA_F: case K_PSA_B: return 18.0;
        case K_GEMM: case K_GEMM_FOLD: return st.count > 8 ? 10.0 : 5.0;
        case K_SEGRED: return 9.0;
        default: return 5.0;
        }
    };
    const double EDGE_IN = 2.0, EDGE_X = 6.0;
    std::vector<double> fin(n, 0.0);
    std::vector<int> str(n, 0);
    double ready[1 + N_SIDE] = {};
    // clock[i][r]: the latest launch index on stream r known to be complete when launch i starts (through stream order and waits)
    std::vector<std::array<int, 1 + N_SIDE>> clock(n);
    int tail[1 + N_SIDE]; for (int q = 0; q < NS; ++q) tail[q] = -1;
    for (int i = 0; i < n; ++i) {
        std::vector<int> preds;
        for (int j = 0; j < i; ++j) if (conflicts(acc[i], acc[j])) preds.push_back(j);
        int best = 0; double best_start = 1e30;
        for (int q = 0; q < NS; ++q) {
            double start = ready[q];
            for (int j : preds) start = std::max(start, fin[j] + (str[j] == q ? 0.0 : EDGE_X));
            if (start < best_start - 1e-9) { best_start = start; best = q; }
        }
        const int q = best;
        str[i] = q; L[i].chain = q;
        fin[i] = best_start + dur(L[i]) + EDGE_IN; ready[q] = fin[i];
        // what is known complete when i starts: its stream's tail (and what that knew), then the waits
        std::array<int, 1 + N_SIDE> ck; ck.fill(-1);
        if (tail[q] >= 0) { ck = clock[tail[q]]; ck[q] = tail[q]; }
        int need[1 + N_SIDE]; for (int r = 0; r < NS; ++r) need[r] = -1;
        for (int j : preds) if (str[j] != q && j > ck[str[j]]) need[str[j]] = std::max(need[str[j]], j);
        int nw = 0;
        for (int r = 0; r < NS; ++r) {
            if (need[r] < 0) continue;
            const int j = need[r];
            if (L[j].signal < 0) { if (n_signals >= N_EVENTS) return false; L[j].signal = n_signals++; }
            L[i].wait[nw++] = L[j].signal;
            for (int t = 0; t < NS; ++t) ck[t] = std::max(ck[t], clock[j][t]);
            ck[r] = std::max(ck[r], j);
        }
        clock[i] = ck; tail[q] = i;
    }
    // check: every conflicting pair (j < i) is ordered
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (conflicts(acc[i], acc[j]) && !(str[j] == str[i] || clock[i][str[j]] >= j)) return false;
    return true;
}

void mst_plan::schedule() {
    // Two GEMM tilings, both on v_mfma_f32_32x32x2_f32: few clips per launch are latency-bound and want many small
    // workgroups with a short k chain (32x32 tiles, 16 waves split the k-tile); from about six clips per launch on
    // the launches fill the chip and the 64x64-tile kernel (16-byte tile loads) wins (measured crossover between 4 and 8
    // clips: 2475 vs 2306 clip-it/s at 4, 3060 vs 3340 at 8).  mst_plan_options.gemm_tile = 32 | 64 forces one (experiments, tests).
    mfma = opt.gemm_tile == 64 ? 1 : (opt.gemm_tile == 32 ? 0 : (K() >= 6));
    std::vector<Step> fwd, bwd;
    for (auto& op : ops) for (auto s : op.fwd) { s.stage = op.stage; fwd.push_back(s); }
    for (size_t i = ops.size(); i-- > 0;) for (auto s : ops[i].bwd) { s.stage = ops[i].stage; bwd.push_back(s); }
    schedule_pass(fwd, sched[0], false);
    schedule_pass(bwd, sched[1], false);
    // mst_plan_options.branches: the whole-model lists' launches are spread over streams along the dependency DAG
    branches = opt.branches == 1 && !tiled();
    schedule_pass(fwd, sched_all[0], true);
    schedule_pass(bwd, sched_all[1], true);
    std::vector<Acc> zs, za;
    first_writers(sched[1], 0, true, zs);
    first_writers(sched_all[1], 0, false, za);
    if (branches && !(assign_streams(sched_all[0]) && assign_streams(sched_all[1]))) {
        branches = false;
        for (int ps = 0; ps < 2; ++ps) for (Step& st : sched_all[ps]) { st.chain = -1; st.signal = -1; st.wait[0] = st.wait[1] = st.wait[2] = -1; }
    }
    // zero lists -> chunks of <= 16 K floats (one workgroup each); Acc.space carries the stage of the step that needs it
    auto chunks = [&](const std::vector<Acc>& z, int stage_mask, std::vector<ZeroChunk>& out) {
        for (const Acc& a : z) {
            if (!(a.space & stage_mask)) continue;
            for (int64_t at = a.lo; at < a.hi; at += 16384) out.push_back(ZeroChunk{at, (int32_t)std::min<int64_t>(16384, a.hi - at), 0});
        }
    };
    for (int st = 0; st < 3; ++st) chunks(zs, 1 << st, zero_stage[st]);
    chunks(za, MST_STAGE_ALL, zero_all);
    // one-clip plans: the exchange tags of the multi-workgroup LSTM ride on the same launch (the scratch arena follows the
    // gradient arena, so a chunk can address it); mst_train_iteration then skips the separate clear
    if (K() == 1)
        for (const LstmDesc& l : lstms)
            if (l.multi) { zero_all.push_back(ZeroChunk{act_top + l.xch_off, (int32_t)(2 * (2 * l.H + 8 * l.H)), 0}); tags_in_zero = true; }
    if (!tiled()) return;
    // ---- one train iteration of a tiled plan as phases that end at an exchange
    {   // loss partial sums (7 per note tensor, one row per workgroup) -> 16 floats that the ranks sum
        const bool U = d.has_unpitched != 0;
        loss_sum_off = tmp(16);
        const int64_t np = (int64_t)P() * NF * NPN, nu = U ? (int64_t)Q() * NF * NUN : 0;
        FoldDesc f{}; f.space = SP_TMP; f.part_off = loss_scratch; f.nrows = loss_blocks(np); f.row_stride = 8; f.ncols = 7; f.col_stride = 1;
        f.sum_off = loss_sum_off;
        loss_fold[0] = (int)folds.size(); folds.push_back(f);
        f.part_off = loss_scratch + (mst_loss_scratch_floats() - 64) / 2; f.nrows = U ? loss_blocks(nu) : 1; f.sum_off = loss_sum_off + 8;
        loss_fold[1] = (int)folds.size(); folds.push_back(f);
    }
    auto mk = [](int what, int pass, int begin, int end) { Phase ph{}; ph.what = what; ph.pass = pass; ph.begin = begin; ph.end = end; ph.nx = 0; return ph; };
    // An exchange is DEFERRED past every following step that does not touch its range (the list is a valid order, and such a step
    // cannot depend on what the exchange delivers): it joins the group of pending exchanges, and the phase is cut only in front of
    // the first step that reads or writes a pending range — all pending exchanges then travel as one collective.  13 ranges per
    // iteration become the number of true dependency points between exchanged data and its consumers.
    auto split = [&](int pass) {
        const std::vector<Step>& L = sched_all[pass];
        int begin = 0;
        Phase cur = mk(0, pass, 0, 0);
        auto touches_pending = [&](const Step& st) {
            if (!cur.nx) return false;
            std::vector<Acc> acc;
            const bool plain = st.kind >= K_COPY_F && st.kind != K_GEMM_FOLD;
            accesses(st, acc, !plain);
            for (int q = 0; q < cur.nx; ++q) {
                const Xchg& x = xchgs[cur.xchg[q]];
                for (const Acc& a : acc)
                    if (a.space == x.space && a.lo < x.off + x.len && x.off < a.hi) return true;
            }
            return false;
        };
        for (int i = 0; i < (int)L.size(); ++i) {
            const bool is_x = L[i].kind == K_XCHG;
            if ((is_x && cur.nx == MST_MAX_XCHG) || (!is_x && touches_pending(L[i]))) {
                cur.begin = begin; cur.end = i;          // steps [begin, i) run (exchange steps among them are skipped), then the exchanges
                phases.push_back(cur);
                begin = i; cur = mk(0, pass, 0, 0);
            }
            if (is_x) cur.xchg[cur.nx++] = L[i].first;
        }
        cur.begin = begin; cur.end = (int)L.size();
        phases.push_back(cur);                            // (a trailing group of exchanges, if any, is delivered after the last step)
    };
    split(0);
    const int lx = (int)xchgs.size();
    xchgs.push_back(Xchg{SP_TMP, loss_sum_off, 16});
    { Phase ph = mk(1, 0, 0, 0); ph.nx = 1; ph.xchg[0] = lx; phases.push_back(ph); }
    phases.push_back(mk(2, 0, 0, 0));
    split(1);
}

// two column blocks of one row-major matrix (row pitch ld), given by the offsets of their first elements from the matrix's
// element (0, 0): do they share no element?  (both start in row 0 or any row: only the column intervals matter)
static bool slab_columns_disjoint(int64_t off_x, int width_x, int64_t off_y, int width_y, int ld) {
    const int64_t cx = off_x % ld, cy = off_y % ld;
    if (cx + width_x > ld || cy + width_y > ld) return false;       // a block that wraps a row is not a column block
    return cx + width_x <= cy || cy + width_y <= cx;
}
extern "C" int32_t mst_debug_slab_columns_disjoint(int64_t off_x, int32_t width_x, int64_t off_y, int32_t width_y, int32_t ld) {
    return slab_columns_disjoint(off_x, width_x, off_y, width_y, ld) ? 1 : 0;
}

template <class D>
static int up(const std::vector<D>& v, D** dev) {
    *dev = nullptr;
    if (v.empty()) return 0;
    if (hipMalloc((void**)dev, v.size() * sizeof(D)) != hipSuccess) return MST_ERR_ALLOC;
    if (hipMemcpy(*dev, v.data(), v.size() * sizeof(D), hipMemcpyHostToDevice) != hipSuccess) return MST_ERR_ALLOC;
    return 0;
}

int mst_plan::upload() {
    int e = 0;
    e |= up(s_gemms, &d_gemms); e |= up(s_gathers, &d_gathers); e |= up(s_segreds, &d_segreds); e |= up(s_lstms, &d_lstms);
    e |= up(s_combines, &d_combines); e |= up(s_notes, &d_notes); e |= up(s_rowlins, &d_rowlins);
    e |= up(s_gemm_owner, &d_gemm_owner);
    e |= up(copies, &d_copies); e |= up(folds, &d_folds); e |= up(zero_fwd, &d_zero_fwd);
    for (int st = 0; st < 3; ++st) e |= up(zero_stage[st], &d_zero_stage[st]);
    e |= up(zero_all, &d_zero_all);
    for (int s = 0; s < 3; ++s) {
        for (auto& ent : slabs[s]) { ent.reps = ent.single ? 1 : K(); ent.rep_stride = tmp_top; }
        e |= up(slabs[s], &d_slabs[s]);
        for (size_t i = 0; i < slabs[s].size(); ++i)
            for (int st = 0; st < slabs[s][i].count; st += 64) slab_blocks[s].push_back(SlabBlock{(int)i, st});
        e |= up(slab_blocks[s], &d_slab_blocks[s]);
    }
    {   // one list for the whole-model backward; two workgroups must never add into the same parameter element concurrently
        std::vector<std::pair<int64_t, int64_t>> spans;
        for (int s = 0; s < 3; ++s)
            for (auto& ent : slabs[s]) {
                const int64_t hi = ent.width > 0 ? ent.dst + (int64_t)(ent.count / ent.width - 1) * ent.dst_ld + ent.width : ent.dst + ent.count;
                spans.push_back({ent.dst, hi});
                slabs_all.push_back(ent);
            }
        slabs_all_ok = true;
        std::vector<std::pair<int64_t, int64_t>> sorted_spans = spans;
        std::sort(sorted_spans.begin(), sorted_spans.end());
        for (size_t i = 1; i < sorted_spans.size(); ++i)
            if (sorted_spans[i].first < sorted_spans[i - 1].second) {
                // column blocks of one weight matrix interleave inside a bounding span: accept only exact 2-D disjointness
                slabs_all_ok = false;
            }
        if (!slabs_all_ok) {          // precise check: element sets of 2-D entries (rows of `width` at stride dst_ld)
            slabs_all_ok = true;
            for (size_t i = 0; i < slabs_all.size() && slabs_all_ok; ++i)
                for (size_t j = i + 1; j < slabs_all.size() && slabs_all_ok; ++j) {
                    const SlabEntry& x = slabs_all[i]; const SlabEntry& y = slabs_all[j];
                    if (spans[i].second <= spans[j].first || spans[j].second <= spans[i].first) continue;
                    // overlapping bounding spans: disjoint only if both are column blocks of the SAME parameter matrix (same base,
                    // same row pitch) with disjoint column ranges, columns counted from the matrix's own first element
                    if (!(x.width > 0 && y.width > 0 && x.dst_ld == y.dst_ld && x.base == y.base)) { slabs_all_ok = false; break; }
                    slabs_all_ok = slab_columns_disjoint(x.dst - x.base, x.width, y.dst - y.base, y.width, x.dst_ld);
                }
        }
        for (size_t i = 0; i < slabs_all.size(); ++i)
            for (int st = 0; st < slabs_all[i].count; st += 64) slab_blocks_all.push_back(SlabBlock{(int)i, st});
        e |= up(slabs_all, &d_slabs_all);
        e |= up(slab_blocks_all, &d_slab_blocks_all);
    }
    return e ? MST_ERR_ALLOC : MST_OK;
}

extern "C" int32_t mst_widths_supported(const mst_dims* d) {
    if (!dims_ok(d)) return MST_ERR_ARG;
    const Sizes z = mst_sizes(*d);
    if (!notes_widths_supported(z.MEL, z.ME_CW, z.PSA_ML)) return MST_ERR_UNSUPPORTED;
    if (z.H > 256 || z.SE_L > 256 || z.HB > 256) return MST_ERR_UNSUPPORTED;
    return MST_OK;
}

extern "C" mst_plan* mst_plan_create(const mst_dims* d, int32_t* status) { return mst_plan_create_ex(d, nullptr, status); }

extern "C" mst_plan* mst_plan_create_ex(const mst_dims* d, const mst_plan_options* opt, int32_t* status) {
    int32_t dummy; if (!status) status = &dummy;
    if (!dims_ok(d)) { *status = MST_ERR_ARG; return nullptr; }
    if (opt && ((opt->gemm_tile != 0 && opt->gemm_tile != 32 && opt->gemm_tile != 64) || opt->gemm_run < 0 || opt->gemm_run > 64)) { *status = MST_ERR_ARG; return nullptr; }
    if (opt && (opt->lstm_flavour < 0 || opt->lstm_flavour > 2 || opt->dense_flavour < 0 || opt->dense_flavour > 2 || opt->branches < 0 || opt->branches > 1)) { *status = MST_ERR_ARG; return nullptr; }
    if (opt && (opt->tile_rows < 0 || opt->tile_r0 < 0 || (opt->tile_rows > 0 && (opt->tile_r0 + opt->tile_rows > d->R || d->clips > 1)))) {
        *status = MST_ERR_ARG; return nullptr;
    }
    mst_plan* p = new mst_plan();
    if (opt) p->opt = *opt;
    p->d = *d; p->z = mst_sizes(*d);
    if (p->d.clips < 1) p->d.clips = 1;
    build_params(*d, p->z, p->pt);
    p->build();
    if (!p->err) p->schedule();
    if (p->err) { *status = p->err; delete p; return nullptr; }
    int e = p->upload();
    if (e) { *status = e; mst_plan_destroy(p); return nullptr; }
    *status = MST_OK;
    return p;
}

extern "C" void mst_plan_destroy(mst_plan* p) {
    if (!p) return;
    hipFree(p->d_gemm_owner); hipFree(p->d_rowlins); hipFree(p->d_zero_all); hipFree(p->d_copies); hipFree(p->d_folds); hipFree(p->d_zero_fwd);
    for (int st = 0; st < 3; ++st) hipFree(p->d_zero_stage[st]);
    hipFree(p->d_gemms); hipFree(p->d_gathers); hipFree(p->d_segreds); hipFree(p->d_lstms); hipFree(p->d_combines); hipFree(p->d_notes);
    for (int s = 0; s < 3; ++s) { hipFree(p->d_slabs[s]); hipFree(p->d_slab_blocks[s]); }
    hipFree(p->d_slabs_all); hipFree(p->d_slab_blocks_all);
    for (mst_plan::SideSet& q : p->sides) {
        if (q.ev_fork) hipEventDestroy(q.ev_fork);
        for (int i = 0; i < mst_plan::N_SIDE; ++i) { if (q.ev_join[i]) hipEventDestroy(q.ev_join[i]); if (q.side[i]) hipStreamDestroy(q.side[i]); }
        for (hipEvent_t e : q.ev) if (e) hipEventDestroy(e);
    }
    delete p;
}

extern "C" int64_t mst_plan_workspace_floats(const mst_plan* p) { return p ? (int64_t)p->K() * (2 * p->act_top + p->tmp_top) : MST_ERR_ARG; }

extern "C" int32_t mst_plan_gemm_tile(const mst_plan* p) { return p ? (p->mfma ? 64 : 32) : MST_ERR_ARG; }

extern "C" int32_t mst_plan_layout(const mst_plan* p, int64_t out[4]) {
    if (!p || !out) return MST_ERR_ARG;
    out[0] = p->K(); out[1] = p->act_top; out[2] = p->tmp_top; out[3] = (int64_t)p->K() * p->act_top;
    return MST_OK;
}

extern "C" int32_t mst_plan_tensor(const mst_plan* p, const char* name, int64_t* off, int64_t* goff, int64_t* numel) {
    if (!p || !name) return MST_ERR_ARG;
    auto it = p->named.find(name);
    if (it == p->named.end()) return MST_ERR_ARG;
    if (off) *off = it->second.off;
    if (goff) *goff = (int64_t)p->K() * p->act_top + it->second.off;
    if (numel) *numel = (int64_t)it->second.rows * it->second.cols;
    return MST_OK;
}

extern "C" int32_t mst_plan_launch_count(const mst_plan* p, int32_t mask, int32_t backward) {
    if (!p) return MST_ERR_ARG;
    int n = 0;
    for (auto& s : p->list(mask, backward)) {
        if (!(s.stage & mask)) continue;
        n += (((s.kind == K_COMB_F || s.kind == K_COMB_B) && s.b != 0) || (s.kind == K_SEGRED && s.b > 0)) ? 2 : 1;
    }
    if (backward) for (int s = 0; s < 3; ++s) if ((mask >> s) & 1) n += 1;
    return n;
}

static Bases make_bases(const mst_plan* p, const float* params, float* gparams, float* ws, const float* pitched,
                        const float* unpitched) {
    Bases b;
    b.p[SP_WS] = ws; b.p[SP_PAR] = const_cast<float*>(params); b.p[SP_GPAR] = gparams;
    b.p[SP_EXT0] = const_cast<float*>(pitched); b.p[SP_EXT1] = const_cast<float*>(unpitched);
    b.p[SP_GRAD] = ws + (int64_t)p->K() * p->act_top; b.p[SP_TMP] = ws + 2 * (int64_t)p->K() * p->act_top;
    b.flags = 0;
    return b;
}

static int run_step(const mst_plan* p, const Step& s, const Bases& b, hipStream_t st) {
    switch (s.kind) {
    case K_GEMM: case K_GEMM_FOLD: return launch_gemm(p->d_gemms + s.first, p->d_gemm_owner + s.c, s.b, s.a, s.count / s.b, p->mfma, b, st);
    case K_GATHER: return launch_gather(p->d_gathers + s.first, s.count, s.a, b, st);
    case K_SEGRED: return launch_segred(p->d_segreds + s.first, s.count / p->K(), s.a, p->K(), s.b, b, st);
    case K_LSTM_T: return launch_lstm_transpose(p->d_lstms + s.first, s.count, s.b, p->s_lstms[s.first].multi, b, st);
    case K_LSTM_F: return launch_lstm_fwd(p->d_lstms + s.first, s.count, s.a, s.b, p->s_lstms[s.first].multi, b, st);
    case K_LSTM_B: return launch_lstm_bwd(p->d_lstms + s.first, s.count, s.a, s.b, p->s_lstms[s.first].multi, b, st);
    case K_COMB_F: return launch_combine_fwd(p->d_combines + s.first, s.count, s.a, s.b == 0 ? (p->d.C <= 4 ? 2 : 1) : 0, b, st);
    case K_COMB_B: return launch_combine_bwd(p->d_combines + s.first, s.count, s.a, s.b == 0 ? (p->d.C <= 4 ? 2 : 1) : 0, b, st);
    case K_COMB_F1: return launch_combine_phase(p->d_combines + s.first, s.a, 0, b, st);
    case K_COMB_F2: return launch_combine_phase(p->d_combines + s.first, s.a, 1, b, st);
    case K_COMB_B1: return launch_combine_phase(p->d_combines + s.first, s.a, 2, b, st);
    case K_COMB_B2: return launch_combine_phase(p->d_combines + s.first, s.a, 3, b, st);
    case K_COPY_F: return launch_copy_rows(p->d_copies + s.first, p->copies[s.first], 0, b, st);
    case K_COPY_B: return launch_copy_rows(p->d_copies + s.first, p->copies[s.first], 1, b, st);
    case K_FOLD: return launch_fold(p->d_folds + s.first, 0, b, st);
    case K_SPREAD: return launch_fold(p->d_folds + s.first, 1, b, st);
    case K_XCHG: return MST_ERR_UNSUPPORTED;        // a tiled plan runs through mst_tiled_phase, which stops at exchanges
    case K_LIN_F: case K_LIN_A: case K_LIN_W: {
        LinDesc l = p->lins[s.first];
        l.x_cs = p->shift(l.x_space, 1); l.y_cs = p->act_top; l.gx_cs = p->act_top;
        if (s.kind == K_LIN_F) return launch_lin_fwd(l, b, st);
        if (s.kind == K_LIN_W) return launch_lin_dw(l, b, st);
        return launch_lin_dx(l, b, l.first[(b.flags & MST_BF_ALL_STAGES) ? 1 : 0], st);
    }
    case K_CONV_P: { ConvDesc c = p->convs[s.first]; c.clip_stride = p->act_top; return launch_conv_prep(c, b, st); }
    case K_CONV_F: { ConvDesc c = p->convs[s.first]; c.clip_stride = p->act_top; return launch_conv_fwd(c, b, st); }
    case K_CONV_W: { ConvDesc c = p->convs[s.first]; c.clip_stride = p->act_top; return launch_conv_dw(c, b, st); }
    case K_ROW_F: return launch_rowlin_fwd(p->d_rowlins + s.first, p->s_rowlins[s.first], s.count, b, st);
    case K_ROW_B: return launch_rowlin_bwd(p->d_rowlins + s.first, p->s_rowlins[s.first], s.count, b, st);
    case K_ME_SQ: return launch_me_sumsq(p->d_notes + s.first, p->s_notes[s.first], s.count, b, st);
    case K_ME_RED: return launch_me_bwd_reduce(p->d_notes + s.first, p->s_notes[s.first], s.count, b, st);
    case K_ME_F: return launch_me_notes_fwd(p->d_notes + s.first, p->s_notes[s.first], s.count, b, st);
    case K_ME_B: return launch_me_notes_bwd(p->d_notes + s.first, p->s_notes[s.first], s.count, b, st);
    case K_PSA_F: return launch_psa_notes_fwd(p->d_notes + s.first, p->s_notes[s.first], s.count, b, st);
    case K_PSA_B: return launch_psa_notes_bwd(p->d_notes + s.first, p->s_notes[s.first], s.count, b, st);
    }
    return MST_ERR_ARG;
}

// Run one pass (a scheduled list filtered by stage) on the caller's stream.
static int run_pass(const mst_plan* p, const std::vector<Step>& list, int mask, const Bases& b0, hipStream_t main, bool tags_cleared = false) {
    Bases b = b0;
    if (mask == MST_STAGE_ALL) b.flags |= MST_BF_ALL_STAGES;            // p->list(mask, .) is the whole-model list then
    auto runs = [&](const Step& s) {
        if (!(s.stage & mask)) return false;
        if (tags_cleared && s.kind == K_LSTM_T && p->s_lstms[s.first].multi) return false;     // its only job was the clear
        return true;
    };
    // Streams (plans with `branches`, whole-model lists): launch s goes to stream s.chain (0 = the caller's, c = side stream c - 1)
    // behind the waits assign_streams gave it; a side stream's first launch of the pass also waits for the caller's stream (fork),
    // and the caller's stream waits for every side stream at the end (join) — parallel branches once captured into a hipGraph.
    mst_plan::SideSet* ss = (p->branches && mask == MST_STAGE_ALL) ? p->side_set(main) : nullptr;
    bool forked = false, used[mst_plan::N_SIDE] = {};
    for (auto& s : list) {
        if (!runs(s)) continue;
        hipStream_t st = main;
        if (ss) {
            if (s.chain > 0) {
                const int c = s.chain - 1;
                if (!forked) { if (hipEventRecord(ss->ev_fork, main) != hipSuccess) return MST_ERR_LAUNCH; forked = true; }
                if (!used[c]) { if (hipStreamWaitEvent(ss->side[c], ss->ev_fork, 0) != hipSuccess) return MST_ERR_LAUNCH; used[c] = true; }
                st = ss->side[c];
            }
            for (int w = 0; w < 3; ++w)
                if (s.wait[w] >= 0 && hipStreamWaitEvent(st, ss->ev[s.wait[w]], 0) != hipSuccess) return MST_ERR_LAUNCH;
        }
        int e = run_step(p, s, b, st);
        if (e) return e < 0 ? e : MST_ERR_LAUNCH;
        if (ss && s.signal >= 0 && hipEventRecord(ss->ev[s.signal], st) != hipSuccess) return MST_ERR_LAUNCH;
    }
    if (ss)
        for (int c = 0; c < mst_plan::N_SIDE; ++c)
            if (used[c] && (hipEventRecord(ss->ev_join[c], ss->side[c]) != hipSuccess || hipStreamWaitEvent(main, ss->ev_join[c], 0) != hipSuccess)) return MST_ERR_LAUNCH;
    return MST_OK;
}

extern "C" int32_t mst_forward(const mst_plan* p, int32_t mask, const float* params, float* ws, const float* pitched,
                               const float* unpitched, mst_stream stream) {
    if (!p || !params || !ws) return MST_ERR_ARG;
    if ((mask & MST_STAGE_EXTRACT) && (!pitched || (p->d.has_unpitched && !unpitched))) return MST_ERR_ARG;
    const Bases b = make_bases(p, params, nullptr, ws, pitched, unpitched);
    return run_pass(p, p->list(mask, 0), mask, b, (hipStream_t)stream);
}

extern "C" int32_t mst_plan_status(const mst_plan* p, float* ws, int32_t clear, int32_t* status, mst_stream stream) {
    if (!p || !ws || !status) return MST_ERR_ARG;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return MST_ERR_LAUNCH;
    int32_t word = 0;
    if (hipMemcpy(&word, ws + p->status_off, sizeof(word), hipMemcpyDeviceToHost) != hipSuccess) return MST_ERR_LAUNCH;
    *status = word;
    if (clear && word) {
        const int32_t zero = 0;
        if (hipMemcpy(ws + p->status_off, &zero, sizeof(zero), hipMemcpyHostToDevice) != hipSuccess) return MST_ERR_LAUNCH;
    }
    return MST_OK;
}

extern "C" int32_t mst_zero_grads(const mst_plan* p, int32_t mask, float* ws, mst_stream stream) {
    if (!p || !ws) return MST_ERR_ARG;
    // only the ranges the first-writer analysis could not prove written before they are read / accumulated into
    float* g = ws + (int64_t)p->K() * p->act_top;
    if ((mask & MST_STAGE_ALL) == MST_STAGE_ALL)
        return launch_zero(p->d_zero_all, (int)p->zero_all.size(), p->K(), g, p->act_top, (hipStream_t)stream) ? MST_ERR_LAUNCH : MST_OK;
    for (int s = 0; s < 3; ++s) {
        if (!((mask >> s) & 1)) continue;
        if (launch_zero(p->d_zero_stage[s], (int)p->zero_stage[s].size(), p->K(), g, p->act_top, (hipStream_t)stream)) return MST_ERR_LAUNCH;
    }
    return MST_OK;
}

static int32_t backward_impl(const mst_plan* p, int32_t mask, const float* params, float* gparams, float* ws,
                             const float* pitched, const float* unpitched, mst_stream stream, int32_t flags) {
    if (!p || !params || !gparams || !ws) return MST_ERR_ARG;
    if ((mask & MST_STAGE_EXTRACT) && (!pitched || (p->d.has_unpitched && !unpitched))) return MST_ERR_ARG;
    Bases b = make_bases(p, params, gparams, ws, pitched, unpitched);
    b.flags = flags;
    {
        int e = run_pass(p, p->list(mask, 1), mask, b, (hipStream_t)stream);
        if (e) return e;
    }
    if ((mask & MST_STAGE_ALL) == MST_STAGE_ALL && p->slabs_all_ok)      // whole model: the three stages' reductions in one launch
        return launch_slab_reduce(p->d_slabs_all, p->d_slab_blocks_all, (int)p->slab_blocks_all.size(), b, (hipStream_t)stream) ? MST_ERR_LAUNCH : MST_OK;
    for (int s = 2; s >= 0; --s) {
        if (!((mask >> s) & 1)) continue;
        if (launch_slab_reduce(p->d_slabs[s], p->d_slab_blocks[s], (int)p->slab_blocks[s].size(), b, (hipStream_t)stream))
            return MST_ERR_LAUNCH;
    }
    return MST_OK;
}

extern "C" int32_t mst_backward(const mst_plan* p, int32_t mask, const float* params, float* gparams, float* ws,
                                const float* pitched, const float* unpitched, mst_stream stream) {
    return backward_impl(p, mask, params, gparams, ws, pitched, unpitched, stream, 0);
}

extern "C" int32_t mst_train_iteration(const mst_plan* p, const float* params, float* gparams, float* ws,
                                       const float* pitched, const float* unpitched, float* losses, mst_stream stream) {
    if (!p || !params || !gparams || !ws || !pitched) return MST_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int e = mst_zero_grads(p, MST_STAGE_ALL, ws, stream);
    if (e) return e;
    if (p->d.has_unpitched && !unpitched) return MST_ERR_ARG;
    e = run_pass(p, p->list(MST_STAGE_ALL, 0), MST_STAGE_ALL, make_bases(p, params, nullptr, ws, pitched, unpitched), st, p->tags_in_zero);
    if (e) return e;
    const bool U = p->d.has_unpitched != 0;
    const int K = p->K();
    float* g = ws + (int64_t)K * p->act_top;
    auto at = [&](const char* n) { return p->named.at(n).off; };
    const int64_t np = (int64_t)p->P() * NF * NPN, nu = U ? (int64_t)p->Q() * NF * NUN : 0;
    float* lscratch = ws + 2 * (int64_t)K * p->act_top + p->loss_scratch;
    const LossBatch lb = {K, p->act_top, p->act_top, p->tmp_top, p->ext0_stride(), p->ext1_stride()};
    e = loss_fwd_batched(ws + at("pitched_pred"), pitched, np, U ? ws + at("unpitched_pred") : nullptr, U ? unpitched : nullptr,
                         nu, ws + at("instruments_pred"), ws + at("used_instruments"), p->z.NI, ws + at("mode_pred"),
                         ws + at("mode"), ws + at("bpm_pred"), ws + at("bpm_target"), 1, ws + p->t_losses.off,
                         ws + p->t_saved.off, lscratch, lb, st, ws + p->t_gl.off, losses);
    if (e) return e;
    e = loss_bwd_batched(ws + at("pitched_pred"), pitched, np, U ? ws + at("unpitched_pred") : nullptr, U ? unpitched : nullptr,
                         nu, ws + at("instruments_pred"), ws + at("used_instruments"), p->z.NI, ws + at("mode_pred"),
                         ws + at("mode"), ws + at("bpm_pred"), ws + at("bpm_target"), ws + p->t_saved.off,
                         ws + p->t_gl.off, nullptr /* the applier's backward kernel derives dL/d pitched_pred itself */,
                         U ? g + at("unpitched_pred") : nullptr,
                         g + at("instruments_pred"), g + at("mode_pred"), g + at("bpm_pred"), lb, st);
    if (e) return e;
    e = backward_impl(p, MST_STAGE_ALL, params, gparams, ws, pitched, unpitched, stream, MST_BF_LOSS_FUSED);
    if (e) return e;
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}


// ------------------------------------------------------------------------------------------ tiled plans
// One train iteration (train-model.py:113-126) of ONE long clip whose bars are tiled over several ranks (BASELINE.json
// configs[4]).  Every rank builds a plan for its bar tile (mst_plan_options.tile_r0 / tile_rows) and walks the
// same phase list; after a phase that reports xlen > 0 the host all-reduces (SUM) ws[xoff, xoff + xlen) over the ranks
// before the next phase.  What crosses ranks: the last-beat states feeding the replicated bar-level LSTMs (an all-gather as a
// SUM over zero-padded buffers), the partial sums of every global reduction (combine's per-channel sums of squares and its
// backward's sum(g x), sum(g out); the loss sums) and, backward, the gradients of the gathered rows.  Everything else is local
// or replicated; gradients of replicated operations computed from a rank's PARTIAL upstream gradient add up to the true
// gradient in the final all-reduce of gparams because backward is linear in the upstream gradient — the song-info loss
// gradients, which are complete on every rank, are therefore seeded on the root rank only.
extern "C" int32_t mst_tiled_phase_count(const mst_plan* p) { return p ? (int32_t)p->phases.size() : MST_ERR_ARG; }

extern "C" int32_t mst_tiled_phase(const mst_plan* p, int32_t phase, const float* params, float* gparams, float* ws,
                                   const float* pitched, const float* unpitched, float* losses, int32_t is_root,
                                   mst_stream stream, int64_t* xoff, int64_t* xlen, int32_t* nx) {
    if (!p || !p->tiled() || phase < 0 || phase >= (int)p->phases.size() || !params || !gparams || !ws || !pitched || !xoff || !xlen || !nx)
        return MST_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const mst_plan::Phase& ph = p->phases[phase];
    Bases b = make_bases(p, params, gparams, ws, pitched, unpitched);
    b.flags |= MST_BF_ALL_STAGES;
    const bool U = p->d.has_unpitched != 0;
    float* g = ws + p->act_top;
    auto at = [&](const char* n) { return p->named.at(n).off; };
    const int64_t np = (int64_t)p->P() * NF * NPN, nu = U ? (int64_t)p->Q() * NF * NUN : 0;
    float* lscratch = ws + 2 * p->act_top + p->loss_scratch;
    if (phase == 0) {
        if (launch_zero(p->d_zero_all, (int)p->zero_all.size(), 1, g, p->act_top, st)) return MST_ERR_LAUNCH;
        if (launch_zero(p->d_zero_fwd, (int)p->zero_fwd.size(), 1, ws, p->act_top, st)) return MST_ERR_LAUNCH;
    }
    if (ph.what == 0) {
        const std::vector<Step>& L = p->sched_all[ph.pass];
        for (int i = ph.begin; i < ph.end; ++i) {
            if (L[i].kind == K_XCHG) continue;          // an earlier exchange of the level this phase ends at: delivered with the others
            int e = run_step(p, L[i], b, st);
            if (e) return e < 0 ? e : MST_ERR_LAUNCH;
        }
        if (ph.pass == 1 && phase == (int)p->phases.size() - 1)
            for (int s = 2; s >= 0; --s)
                if (launch_slab_reduce(p->d_slabs[s], p->d_slab_blocks[s], (int)p->slab_blocks[s].size(), b, st)) return MST_ERR_LAUNCH;
    } else if (ph.what == 1) {
        int e = loss_fwd_partials(ws + at("pitched_pred"), pitched, np, U ? ws + at("unpitched_pred") : nullptr, U ? unpitched : nullptr, nu,
                                  lscratch, st);
        if (e) return e;
        if (launch_fold(p->d_folds + p->loss_fold[0], 0, b, st) || (U && launch_fold(p->d_folds + p->loss_fold[1], 0, b, st))) return MST_ERR_LAUNCH;
    } else {
        if (launch_fold(p->d_folds + p->loss_fold[0], 1, b, st) || (U && launch_fold(p->d_folds + p->loss_fold[1], 1, b, st))) return MST_ERR_LAUNCH;
        int e = loss_fwd_tail(np, nu, U ? 1 : 0, ws + at("instruments_pred"), ws + at("used_instruments"), p->z.NI, ws + at("mode_pred"),
                              ws + at("mode"), ws + at("bpm_pred"), ws + at("bpm_target"), 1, ws + p->t_losses.off, ws + p->t_saved.off,
                              lscratch, st, ws + p->t_gl.off, losses);
        if (e) return e;
        const LossBatch lb = {1, 0, 0, 0, 0, 0};
        e = loss_bwd_batched(ws + at("pitched_pred"), pitched, np, U ? ws + at("unpitched_pred") : nullptr, U ? unpitched : nullptr, nu,
                             ws + at("instruments_pred"), ws + at("used_instruments"), p->z.NI, ws + at("mode_pred"), ws + at("mode"),
                             ws + at("bpm_pred"), ws + at("bpm_target"), ws + p->t_saved.off, ws + p->t_gl.off, g + at("pitched_pred"),
                             U ? g + at("unpitched_pred") : nullptr, g + at("instruments_pred"), g + at("mode_pred"), g + at("bpm_pred"),
                             lb, st, is_root ? 1.f : 0.f);
        if (e) return e;
    }
    *nx = ph.nx;
    for (int q = 0; q < ph.nx; ++q) {
        const mst_plan::Xchg& x = p->xchgs[ph.xchg[q]];
        const int64_t base = x.space == SP_WS ? 0 : (x.space == SP_GRAD ? p->act_top : 2 * p->act_top);
        xoff[q] = base + x.off; xlen[q] = x.len;
    }
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------ instrumentation
// Per-launch-step timing with HIP events recorded on the caller's stream (bench.py's roofline leg).
// Not graph-capturable (it synchronises on events) and never called from the training path.
static void step_cost(const mst_plan* p, const Step& s, double* flops, double* bytes) {
    double f = 0, b = 0;
    switch (s.kind) {
    case K_GEMM: case K_GEMM_FOLD:
        for (int i = 0; i < s.count; ++i) {
            const GemmDesc& g = p->s_gemms[s.first + i];
            f += 2.0 * g.M * g.N * g.K;
            b += 4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N);
        }
        break;
    case K_GATHER:
        for (int i = 0; i < s.count; ++i) { const GatherDesc& g = p->s_gathers[s.first + i]; b += 8.0 * g.rows * g.K; }
        break;
    case K_LIN_F: case K_LIN_A: case K_LIN_W: {
        const LinDesc& l = p->lins[s.first];
        const double rows = (double)l.clips * l.rows;
        f = 2.0 * rows * l.N * l.K;
        b = 4.0 * (rows * l.K + rows * l.N * (s.kind == K_LIN_F ? 1.0 : 2.0) + (double)l.N * l.K);
        break;
    }
    case K_CONV_F: case K_CONV_W: {
        const ConvDesc& c = p->convs[s.first];
        const double rows = (double)c.clips * c.P * NOCT, kk = NF * NPF * CONV_K;
        f = 2.0 * rows * c.OC * kk;
        b = 4.0 * (rows / NOCT * NF * NPN * NPF + rows * c.OC * (s.kind == K_CONV_F ? 1.0 : 2.0) + c.OC * kk);
        break;
    }
    case K_SEGRED:
        for (int i = 0; i < s.count; ++i) {
            const SegRedDesc& r = p->s_segreds[s.first + i];
            const double rows = (double)r.d[0] * r.d[1] * r.d[2] * r.d[3];
            f += rows * r.width; b += 4.0 * (rows * r.width + 2.0 * r.nidx * r.width);
        }
        break;
    case K_LSTM_F: case K_LSTM_B:
        for (int i = 0; i < s.count; ++i) {
            const LstmDesc& l = p->s_lstms[s.first + i];
            f += (double)l.B * l.S * (8.0 * l.H * l.H + 30.0 * l.H);
            b += 4.0 * ((double)l.B * l.S * 11.0 * l.H + 4.0 * l.H * l.H);
        }
        break;
    case K_COMB_F: case K_COMB_B:
        for (int i = 0; i < s.count; ++i) {
            const CombineDesc& c = p->s_combines[s.first + i];
            const double n = (double)c.rows * c.cols;
            f += n * c.Cn * (s.kind == K_COMB_F ? 4.0 : 8.0);
            b += 4.0 * n * (s.kind == K_COMB_F ? 2.0 * c.Cn + 1 : 4.0 * c.Cn + 2);
        }
        break;
    case K_ROW_F: case K_ROW_B:
        for (int i = 0; i < s.count; ++i) {
            const RowLinDesc& r = p->s_rowlins[s.first + i];
            const double mm = 2.0 * r.rows * r.kin * r.nout;
            f += s.kind == K_ROW_F ? mm : 2.0 * mm;
            b += 4.0 * r.rows * (s.kind == K_ROW_F ? r.kin + r.nout : 2.0 * r.kin + 2.0 * r.nout + (r.xgrad ? r.kin : 0));
        }
        break;
    case K_ME_SQ: case K_ME_F: case K_ME_RED: case K_ME_B: {
        // algorithmic = the reference's (unfused) shapes; the recomputation the fused kernels do on top is not counted
        const NotesDesc& n = p->s_notes[s.first];
        const double pos = (double)n.C * n.Q * NF * NPN * s.count, mel = pos / n.C * n.W;
        const double per = 2.0 * n.W + 2.0 * n.CW * NPF + 2.0 * n.W * (n.W + n.CW);
        if (s.kind == K_ME_SQ) { f = pos * (per + 2.0 * n.W); b = 4.0 * pos * NPF; }
        else if (s.kind == K_ME_F) { f = pos * 2.0 * n.W; b = 4.0 * (pos * NPF + mel); }
        else if (s.kind == K_ME_RED) { f = pos * 2.0 * n.W; b = 4.0 * (pos * NPF + 2.0 * mel); }
        else { f = pos * per * 2.0; b = 4.0 * (pos * NPF + mel); }
        break;
    }
    case K_PSA_F: case K_PSA_B: {
        const NotesDesc& n = p->s_notes[s.first];
        const double pos = (double)n.C * n.Q * NF * NPN * s.count;
        const double per = 2.0 * 30 + 2.0 * NPF * (30 + n.ML);
        f = pos * per * (s.kind == K_PSA_F ? 1.0 : 3.0);
        b = 4.0 * (pos * NPF * (s.kind == K_PSA_F ? 1.0 : 2.0) + (double)s.count * n.C * n.Q * NF * 450 * (s.kind == K_PSA_F ? 1.0 : 2.0) +
                   (double)s.count * n.Q * NF * NPN * n.ML * (s.kind == K_PSA_F ? 1.0 : 2.0));
        break;
    }
    }
    *flops = f; *bytes = b;
}

extern "C" int64_t mst_plan_zero_floats(const mst_plan* p, int32_t mask) {
    if (!p) return MST_ERR_ARG;
    int64_t n = 0;
    if ((mask & MST_STAGE_ALL) == MST_STAGE_ALL) { for (auto& c : p->zero_all) n += c.len; return n; }
    for (int s = 0; s < 3; ++s) if ((mask >> s) & 1) for (auto& c : p->zero_stage[s]) n += c.len;
    return n;
}

extern "C" int32_t mst_plan_step_count(const mst_plan* p, int32_t mask, int32_t backward) {
    if (!p) return MST_ERR_ARG;
    int n = 0;
    for (auto& s : p->list(mask, backward)) if (s.stage & mask) ++n;
    return n;
}

// shape of step i of a pass: GEMM {M,N,K,ksplit} of its first descriptor (+count), LSTM {B,S,H,count},
// segment-reduce {nidx max, width, rows, count}; then the member count, the step kind, its dependency level and its chain (-1: none)
extern "C" int32_t mst_plan_step_info(const mst_plan* p, int32_t mask, int32_t backward, int32_t* info /* 8 per step */) {
    if (!p || !info) return MST_ERR_ARG;
    std::vector<const Step*> steps;
    for (auto& s : p->list(mask, backward)) if (s.stage & mask) steps.push_back(&s);
    int idx = 0;
    for (const Step* s : steps) {
        int32_t* o = info + 8 * idx++;
        o[0] = o[1] = o[2] = o[3] = 0; o[4] = s->count; o[5] = s->kind; o[6] = s->lvl; o[7] = s->chain;
        if (s->kind == K_GEMM || s->kind == K_GEMM_FOLD) { const GemmDesc& g = p->s_gemms[s->first]; o[0] = g.M; o[1] = g.N; o[2] = g.K; o[3] = g.ksplit; }
        else if (s->kind >= K_LIN_F && s->kind <= K_LIN_W) { const LinDesc& l = p->lins[s->first]; o[0] = l.rows; o[1] = l.N; o[2] = l.K; o[3] = l.splits; }
        else if (s->kind == K_LSTM_F || s->kind == K_LSTM_B) { const LstmDesc& l = p->s_lstms[s->first]; o[0] = l.B; o[1] = l.S; o[2] = l.H; o[3] = l.multi; }
        else if (s->kind == K_GATHER) { const GatherDesc& g = p->s_gathers[s->first]; o[0] = g.rows; o[1] = g.K; o[2] = g.nseg; }
        else if (s->kind == K_SEGRED) { const SegRedDesc& r = p->s_segreds[s->first]; o[0] = s->a; o[1] = r.width; o[2] = r.d[0] * r.d[1] * r.d[2] * r.d[3]; }
        else if (s->kind == K_COMB_F || s->kind == K_COMB_B) { const CombineDesc& c = p->s_combines[s->first]; o[0] = c.Cn; o[1] = c.rows; o[2] = c.cols; o[3] = c.nblk; }
    }
    return idx;
}

// members of GEMM step i of a pass, one clip's worth: {M, N, K, ksplit, fold_rows, workgroups} each (tools/step_profile.py)
extern "C" int32_t mst_plan_step_gemms(const mst_plan* p, int32_t mask, int32_t backward, int32_t step, int32_t* out, int32_t cap) {
    if (!p || !out) return MST_ERR_ARG;
    int idx = 0;
    for (auto& s : p->list(mask, backward)) {
        if (!(s.stage & mask)) continue;
        if (idx++ != step) continue;
        if (s.kind != K_GEMM && s.kind != K_GEMM_FOLD) return 0;
        int n = 0;
        for (int q = 0; q < s.b && n < cap; ++q, ++n) {
            const GemmDesc& g = p->s_gemms[s.first + q];
            int32_t* o = out + 6 * n;
            o[0] = g.M; o[1] = g.N; o[2] = g.K; o[3] = g.ksplit; o[4] = g.fold_rows; o[5] = gemm_blocks(g, p->mfma);
        }
        return n;
    }
    return MST_ERR_ARG;
}

extern "C" int32_t mst_plan_time_steps(const mst_plan* p, int32_t mask, int32_t backward, const float* params, float* gparams,
                                       float* ws, const float* pitched, const float* unpitched, mst_stream stream,
                                       int32_t reps, float* ms, int32_t* kind, double* flops, double* bytes) {
    if (!p || !params || !ws || !ms || !kind || !flops || !bytes || reps < 1) return MST_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    Bases b = make_bases(p, params, gparams, ws, pitched, unpitched);
    if (backward && !p->tiled()) b.flags = MST_BF_LOSS_FUSED;          // what mst_train_iteration launches
    if ((mask & MST_STAGE_ALL) == MST_STAGE_ALL) b.flags |= MST_BF_ALL_STAGES;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MST_ERR_ALLOC;
    std::vector<const Step*> steps;
    for (auto& s : p->list(mask, backward)) if (s.stage & mask) steps.push_back(&s);
    int idx = 0;
    for (const Step* s : steps) {
        run_step(p, *s, b, st);                       // warm
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) run_step(p, *s, b, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float t = 0.f;
        hipEventElapsedTime(&t, e0, e1);
        ms[idx] = t / reps; kind[idx] = s->kind == K_GEMM_FOLD ? K_GEMM : s->kind;      // same kernel
        step_cost(p, *s, &flops[idx], &bytes[idx]);
        ++idx;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return idx;
}

extern "C" const char* mst_version(void) { return "mst_amd 0.1 (gfx950)"; }
