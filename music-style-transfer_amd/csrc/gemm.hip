// Generic descriptor-driven fp32 GEMM for gfx950: C[m,n] = epi(sum_k A(m,k) B(k,n)).
//
// Every dense contraction of the model goes through this kernel: the broadcast-concat
// Linears (cat_with_broadcast + nn.Linear, style/utils/pytorch.py:54-65), the note-axis
// Conv1d as an implicit-im2col GEMM (style/model.py:46-53,82), the LSTM input projections,
// and all their weight/input gradients.  Operands are *accessors* (dense, activation-gradient,
// im2col, permuted weight, conv-gradient): the conv's im2col matrix and the transposed /
// permuted weight views are never materialised, and bias/activation (or the activation derivative
// on the backward side) are fused into the tile load / epilogue.  The broadcast-concat input of a
// Linear IS materialised once per consumer group by gather_kernel below (its backward is
// segred_kernel), so those GEMMs read one dense operand.
//
// Latency flavour (gemm_kernel): 32x32 outputs x KD-deep k-tile (KD = 32 / 64 / 128 by the descriptor's k range) per
// 1024-thread workgroup; the sixteen waves split the k-tile (in-block split-K), each running its k rows through
// v_mfma_f32_32x32x2_f32 (one 32x32 accumulator tile per wave), LDS tiles stored k-major (+1 pad).  One clip's GEMMs are
// small (tens of MFLOP) and latency-bound, so the tile is chosen for workgroup count and a short dependent k chain.
// Throughput flavour (gemm_mfma_kernel, plans with >= 6 clips per launch): further down.
// Independent GEMMs share one launch: the 1-D grid is the concatenation of every member's
// (tile, k-split) workgroups; k-splits of weight gradients write deterministic slabs that are
// reduced later in order.
#include "mst_common.h"

__device__ __forceinline__ float act_fwd(int act, float z, int col) {
    if (act == ACT_LEAKY) return z > 0.f ? z : z * LEAKY;
    if (act == ACT_SIGOUT) { float s = 1.f / (1.f + expf(-z)); return col == 0 ? 6.f * s : s; }
    if (act == ACT_BPM) { float s = 1.f / (1.f + expf(-z)); return s * 150.f + 50.f; }
    return z;
}

// derivative of the activation expressed through its OUTPUT y (nothing else is saved)
__device__ __forceinline__ float act_bwd(int act, float y, int col) {
    if (act == ACT_LEAKY) return y > 0.f ? 1.f : LEAKY;
    if (act == ACT_SIGOUT) return col == 0 ? y * (1.f - y * (1.f / 6.f)) : y * (1.f - y);
    if (act == ACT_BPM) { float s = (y - 50.f) * (1.f / 150.f); return 150.f * s * (1.f - s); }
    return 1.f;
}

// ---- operand access ------------------------------------------------------------------------
// Operand and epilogue kinds are template parameters, every operand is an affine function of its
// two indices, element offsets are 32-bit, and all loads go through explicit global-address-space
// pointers (a generic pointer would become flat_load, which also counts on lgkmcnt and so stalls
// every LDS read of the FMA loop).  Loading is two-phase: compute every address of the next
// k-tile, issue all loads unconditionally (padding / bias-ones columns read a constant from
// memory instead of branching) so they fly under the current tile's FMAs, consume one tile later.
typedef const MST_GLOBAL_AS float* gcptr;
// LDS tiles are k-major, row stride = tile edge + GEMM_PAD floats.  An ODD stride makes the k-fast tile stores (lanes walk
// k: bank = k * stride mod 32) conflict-free; the MFMA fragment reads (lanes walk the row) are conflict-free at any stride.
// (PMC with the old +4 pad: 2 bank-conflict cycles per LDS instruction in the 64x64 kernel.)
#define GEMM_PAD 1

// Element offset of operand(i, j) — branch-free; `ok` is cleared for im2col padding taps.
// (i, j) = (m, k) for the A operand and (k, n) for the B operand.
template <int KIND>
__device__ __forceinline__ int off_of(const Operand& o, int i, int j, bool& ok) {
    if constexpr (KIND == OPK_DENSE) {
        return i * (int)o.si + j * (int)o.sj;
    } else if constexpr (KIND == OPK_ACTGRAD) {      // value(row, col): transposed => i is the column
        return o.transposed ? j * o.ld + i : i * o.ld + j;
    } else if constexpr (KIND == OPK_IM2COL) {       // i = (p, octave), j = (fraction, tap, feature)
        const int p = i >> 3, oc = i & 7;
        const int f = j / (CONV_K * NPF), rem = j - f * (CONV_K * NPF);
        const int e = (NDEG * oc - CONV_PAD) * NPF + rem;
        ok = ok & (e >= 0) & (e < NPN * NPF);
        return p * (NF * NPN * NPF) + f * (NPN * NPF) + e;
    } else if constexpr (KIND == OPK_PERMW) {        // i = (a, b, c) in activation memory order, j = out feature
        // conv: (fraction, tap, feature) reads W[oc, fraction*5+feature, tap]  (pb=14, pc=5)
        // unpitched linear: (fraction, note, feature) reads W[j, fraction*94 + feature*47 + note]
        // the two permutations the model has, with compile-time divisors (a runtime divisor is a ~25-instruction sequence,
        // twice per element and k-tile: most of what the conv GEMM's waves issued)
        if (o.pb == CONV_K && o.pc == NPF) {
            constexpr int BC = CONV_K * NPF;
            const int a = i / BC, rem = i - a * BC;
            const int bb = rem / NPF, c = rem - bb * NPF;
            return j * o.ld + a * BC + c * CONV_K + bb;
        }
        if (o.pb == NUN && o.pc == NUF) {
            constexpr int BC = NUN * NUF;
            const int a = i / BC, rem = i - a * BC;
            const int bb = rem / NUF, c = rem - bb * NUF;
            return j * o.ld + a * BC + c * NUN + bb;
        }
        const int bc = o.pb * o.pc;
        const int a = i / bc, rem = i - a * bc;
        const int bb = rem / o.pc, c = rem - bb * o.pc;
        return j * o.ld + a * bc + c * o.pb + bb;
    } else {                                          // OPK_CONVGRAD: i = out channel, j = (p, octave)
        return (j >> 3) * (o.oc * NOCT) + i * NOCT + (j & 7);
    }
}

template <int OK>
__device__ __forceinline__ void store_out(const GemmDesc& d, float* cbase, const float* bias, int m, int n, int split, float acc) {
    const OutSpec& o = d.out;
    if constexpr (OK == OUT_STORE) {
        float v = acc;
        if (bias) v += o.bias_div > 0 ? bias[(unsigned)((m / o.bias_div) * o.bias_ld + n)] : bias[n];
        cbase[(unsigned)(m * o.ldc + n)] = act_fwd(o.act, v, n);
    } else if constexpr (OK == OUT_ACCUM) {
        float* c = cbase + (unsigned)(m * o.ldc + n);
        *c = o.first ? acc : *c + acc;              // first writer of this range in the pass: no read, nothing to clear
    } else if constexpr (OK == OUT_CONV) {          // m = (p, octave), n = out channel -> x1[p, n*8 + octave]
        cbase[(unsigned)((m >> 3) * o.ldc + n * NOCT + (m & 7))] = act_fwd(ACT_LEAKY, acc + bias[n], 0);
    } else if constexpr (OK == OUT_SLAB) {
        const unsigned idx = n < o.wcols ? (unsigned)(m * o.wcols + n) : (unsigned)(d.M * o.wcols + m);
        cbase[(int64_t)split * o.slab_stride + idx] = acc;
    } else {                                         // OUT_PERMW_SLAB: m = out feature, n = (a, b, c) | bias column
        unsigned idx;
        if (n < o.wcols) {
            int a, bb, c, bc;
            if (o.pb == CONV_K && o.pc == NPF) { bc = CONV_K * NPF; a = n / (CONV_K * NPF); const int rem = n - a * (CONV_K * NPF); bb = rem / NPF; c = rem - bb * NPF; }
            else if (o.pb == NUN && o.pc == NUF) { bc = NUN * NUF; a = n / (NUN * NUF); const int rem = n - a * (NUN * NUF); bb = rem / NUF; c = rem - bb * NUF; }
            else { bc = o.pb * o.pc; a = n / bc; const int rem = n - a * bc; bb = rem / o.pc; c = rem - bb * o.pc; }
            idx = (unsigned)(m * o.wcols + a * bc + c * o.pb + bb);
        } else {
            idx = (unsigned)(d.M * o.wcols + m);
        }
        cbase[(int64_t)split * o.slab_stride + idx] = acc;
    }
}

template <int AK, int BKIND, int OK, int AKF, int BKF, int KD>
__device__ __forceinline__ void gemm_body(const GemmDesc& d, const Bases& b, const int tile, const int split,
                                          float (*As)[GEMM_BM + GEMM_PAD], float (*Bs)[GEMM_BN + GEMM_PAD]) {
    // 32x32 output tile per workgroup, 128-deep k-tile: wave w owns k rows [32w, 32w+32) of the
    // tile (in-block split-K), so the dependent k chain is K/128 steps and small problems still
    // spread over many workgroups; the four partial tiles are summed through LDS at the end.
    // AKF/BKF: whether consecutive lanes walk k (1) or the m / n index (0) when loading a tile —
    // the contiguous direction of that operand in memory.
    const int tid = threadIdx.x;
    const int M = d.M, N = d.N;
    const int tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
    const gcptr baseA = (gcptr)(b.p[d.A.space] + d.A.off);
    const gcptr baseA2 = (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) ? (gcptr)(b.p[d.A.space2] + d.A.off2) : baseA;
    const gcptr baseB = (gcptr)(b.p[d.B.space] + d.B.off);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    // KD = depth of the k-tile staged per step (32 / 64 / 128, chosen per descriptor from its k range):
    // shallow reductions (K = 8..64 is common here) do not pay for a 128-deep tile
    int kchunk = (d.K + d.ksplit - 1) / d.ksplit;
    kchunk = (kchunk + KD - 1) / KD * KD;
    const int k0 = split * kchunk;
    const int k1 = min(d.K, k0 + kchunk);
    const int wv = tid >> 6, lane = tid & 63;
    const int a_act = (AK == OPK_ACTGRAD) ? d.A.act : ACT_LEAKY;
    const int a_tr = d.A.transposed;
    const int b_ones = (BKIND == OPK_DENSE || BKIND == OPK_IM2COL) ? d.B.ones_at : -1;   // bias-gradient column of B
    // this wave's 32x32 partial tile over its k rows, on the matrix cores: v_mfma_f32_32x32x2_f32 is exact f32 (a k-ordered
    // fmaf chain).  Lane l feeds A[row l&31][k l>>5], B[k l>>5][col l&31]; register r holds C[(r&3) + 8(r>>2) + 4(l>>5)][l&31].
    typedef float acc_f32x16 __attribute__((ext_vector_type(16)));
    acc_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    constexpr int NT = GEMM_THREADS, NW = NT / 64;
    constexpr int NL = GEMM_BM * KD / NT;            // tile elements per lane and operand (2 / 4 / 8)
    constexpr int KW = KD / NW;                      // k rows of the tile each wave reduces
    float va[NL], ya[NL], vb[NL];

    // element e = tid + NT*i of the 32 x KD tile: (row, k) = (e / KD, e % KD) when lanes walk k,
    // (e % 32, e / 32) when lanes walk the row index
#define A_ROW(i) (AKF ? (tid / KD) + (NT / KD) * (i) : (tid & 31))
#define A_KL(i) (AKF ? (tid % KD) : (tid >> 5) + (NT / 32) * (i))
#define B_ROW(i) (BKF ? (tid / KD) + (NT / KD) * (i) : (tid & 31))
#define B_KL(i) (BKF ? (tid % KD) : (tid >> 5) + (NT / 32) * (i))
#define GEMM_ISSUE(KT)                                                                                     \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NL; ++i) {                                                   \
            const int m = tm * GEMM_BM + A_ROW(i), ka = (KT) + A_KL(i);                                    \
            bool oka = (m < M) & (ka < k1);                                                                \
            int ia = off_of<AK>(d.A, m, ka, oka);                                                          \
            ia = oka ? ia : 0;                                                                             \
            const float x = baseA[(unsigned)ia];                                                           \
            if constexpr (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) {                                       \
                const float y = baseA2[(unsigned)ia];                                                      \
                const int col = (AK == OPK_ACTGRAD) ? (a_tr ? m : ka) : 0;                                 \
                ya[i] = act_bwd(a_act, y, col);                                                            \
            } else ya[i] = 1.f;                                                                            \
            va[i] = oka ? x : 0.f;                                                                         \
            const int n = tn * GEMM_BN + B_ROW(i), kb = (KT) + B_KL(i);                                    \
            bool okb = (n < N) & (kb < k1);                                                                \
            const bool one = okb & (n == b_ones);                                                          \
            int ib = off_of<BKIND>(d.B, kb, n, okb);                                                       \
            ib = (okb & !one) ? ib : 0;                                                                    \
            const float w = baseB[(unsigned)ib];                                                           \
            vb[i] = one ? 1.f : (okb ? w : 0.f);                                                           \
        }                                                                                                  \
    }

    if (k0 < k1) GEMM_ISSUE(k0)
    for (int kt = k0; kt < k1; kt += KD) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            As[A_KL(i)][A_ROW(i)] = (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) ? va[i] * ya[i] : va[i];
            Bs[B_KL(i)][B_ROW(i)] = vb[i];
        }
        __syncthreads();
        if (kt + KD < k1) GEMM_ISSUE(kt + KD)                // next tile's loads fly under this tile's FMAs
#pragma unroll
        for (int kk = 0; kk < KW; kk += 2) {
            const int k = wv * KW + kk + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k][lane & 31], Bs[k][lane & 31], acc, 0, 0, 0);
        }
        __syncthreads();
    }
#undef GEMM_ISSUE
#undef A_ROW
#undef A_KL
#undef B_ROW
#undef B_KL
    // sum the waves' partial tiles (fixed order) and run the epilogue
    float* red = &As[0][0];                          // NW x 32 x 32 floats fit in the A+B tile storage (contiguous)
#pragma unroll
    for (int r = 0; r < 16; ++r)
        red[wv * (GEMM_BM * GEMM_BN) + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * GEMM_BN + (lane & 31)] = acc[r];
    __syncthreads();
    float* cbase = b.p[d.out.space] + d.out.off;
    const float* bias = d.out.bias_space >= 0 ? b.p[d.out.bias_space] + d.out.bias_off : nullptr;
#pragma unroll
    for (int q = 0; q < GEMM_BM * GEMM_BN / NT; ++q) {
        const int o = tid + q * NT;
        const int m = tm * GEMM_BM + o / GEMM_BN, n = tn * GEMM_BN + o % GEMM_BN;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w * (GEMM_BM * GEMM_BN) + o];
        if (m < M && n < N) store_out<OK>(d, cbase, bias, m, n, split, v);
    }
}

// ---- throughput flavour: f32 MFMA -----------------------------------------------------------
// Batched plans (many clips per launch) are FLOP-bound, not latency-bound, so their GEMMs run on the
// matrix cores: v_mfma_f32_32x32x2_f32 is exact f32 (bit-for-bit a k-ordered fmaf chain) at the
// chip's f32 peak.  64x64 output tile per 256-lane workgroup, one 32x32 MFMA tile per wave, 32-deep
// k-tile staged k-major in LDS by the same branch-free accessor loaders as above (so every operand
// kind / epilogue of the model is covered), next k-tile prefetched into registers under the MFMAs.
// Fragment maps (cdna_hip_programming.md): lane l feeds A[row = l&31][k = l>>5], B[k = l>>5][col = l&31];
// accumulator register r of lane l is C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
#define MF_BM 64
#define MF_BN 64
#define MF_KD 32
#define MF_THREADS 256
typedef float mf_f32x16 __attribute__((ext_vector_type(16)));
typedef float mf_f4u __attribute__((ext_vector_type(4), aligned(4)));    // 16-byte load that only needs 4-byte alignment

// FOLD: the reduction index runs over (clip, row) (GemmDesc.fold_rows >= MF_KD rows per clip): a k-tile then crosses at most one
// clip boundary, found with one uniform division per k-tile
template <int AK, int BKIND, int OK, int AKF, int BKF, bool FOLD = false>
__device__ __forceinline__ void gemm_mfma_body(const GemmDesc& d, const Bases& b, const int tile, const int split,
                                               float (*As)[MF_BM + GEMM_PAD], float (*Bs)[MF_BN + GEMM_PAD]) {
    const int tid = threadIdx.x;
    const int M = d.M, N = d.N;
    const int tiles_n = (N + MF_BN - 1) / MF_BN;
    const gcptr baseA = (gcptr)(b.p[d.A.space] + d.A.off);
    const gcptr baseA2 = (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) ? (gcptr)(b.p[d.A.space2] + d.A.off2) : baseA;
    const gcptr baseB = (gcptr)(b.p[d.B.space] + d.B.off);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    int kchunk = (d.K + d.ksplit - 1) / d.ksplit;
    kchunk = (kchunk + MF_KD - 1) / MF_KD * MF_KD;
    const int k0 = split * kchunk;
    const int k1 = min(d.K, k0 + kchunk);
    const int wv = tid >> 6, lane = tid & 63;
    const int wm = wv >> 1, wn = wv & 1;             // this wave's 32x32 quadrant of the 64x64 tile
    const int a_act = (AK == OPK_ACTGRAD) ? d.A.act : ACT_LEAKY;
    const int a_tr = d.A.transposed;
    const int b_ones = (BKIND == OPK_DENSE || BKIND == OPK_IM2COL) ? d.B.ones_at : -1;
    const bool live = (tm * MF_BM + wm * 32 < M) & (tn * MF_BN + wn * 32 < N);    // wave-uniform
    const int fr = FOLD ? d.fold_rows : 1;
    const unsigned acs = (unsigned)d.acs, acs2 = (unsigned)d.acs2, bcs = (unsigned)d.bcs;
    mf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NT = MF_THREADS;
    constexpr int NL = MF_BM * MF_KD / NT;           // 8 tile elements per lane and operand
    float va[NL], ya[NL], vb[NL];
    // Slot i = 4*g + j of a lane: element j of its group g.  A group is 4 consecutive elements along the operand's
    // memory-contiguous direction (k when the *KF flag is set, the row index otherwise), so a dense / activation-gradient
    // operand is fetched with ONE 16-byte load per group instead of four 4-byte loads (the kernel is load-latency bound:
    // PMC shows its waves parked on memory 60 % of the time).  A group that crosses M / N / k1, holds the bias-ones
    // column or is not unit-stride falls back to four predicated scalar loads; im2col / permuted-weight operands always do.
    // (operands that are never vectorised keep the lane-contiguous element order: consecutive lanes, consecutive addresses)
#define VROW(KF, g, j, TBX) ((KF) ? (tid + NT * (g)) / (MF_KD / 4) : ((tid + NT * (g)) % ((TBX) / 4)) * 4 + (j))
#define VKL(KF, g, j, TBX) ((KF) ? ((tid + NT * (g)) % (MF_KD / 4)) * 4 + (j) : (tid + NT * (g)) / ((TBX) / 4))
#define SROW(KF, i, TBX) ((KF) ? (tid / MF_KD) + (NT / MF_KD) * (i) : (tid & ((TBX) - 1)))
#define SKL(KF, i, TBX) ((KF) ? (tid % MF_KD) : (tid / (TBX)) + (NT / (TBX)) * (i))
#define A_ROW(i) (VEC_A ? VROW(AKF, (i) >> 2, (i) & 3, MF_BM) : SROW(AKF, i, MF_BM))
#define A_KL(i) (VEC_A ? VKL(AKF, (i) >> 2, (i) & 3, MF_BM) : SKL(AKF, i, MF_BM))
#define B_ROW(i) (VEC_B ? VROW(BKF, (i) >> 2, (i) & 3, MF_BN) : SROW(BKF, i, MF_BN))
#define B_KL(i) (VEC_B ? VKL(BKF, (i) >> 2, (i) & 3, MF_BN) : SKL(BKF, i, MF_BN))
    constexpr bool VEC_A = AK == OPK_DENSE || AK == OPK_ACTGRAD;
    constexpr bool VEC_B = BKIND == OPK_DENSE;
#define MF_ISSUE(KT)                                                                                       \
    {                                                                                                      \
        const int fc0 = FOLD ? (KT) / fr : 0, fkb = FOLD ? (fc0 + 1) * fr : 0;                              \
        _Pragma("unroll") for (int g = 0; g < NL / 4; ++g) {                                               \
            int ia[4], ib[4];                                                                              \
            unsigned ca[4], ca2[4], cb[4];                                                                 \
            bool oka[4], okb[4], one[4];                                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
                const int i = 4 * g + j;                                                                   \
                const int m = tm * MF_BM + A_ROW(i), ka = (KT) + A_KL(i);                                  \
                oka[j] = (m < M) & (ka < k1);                                                              \
                const int cla = FOLD ? (ka >= fkb ? fc0 + 1 : fc0) : 0;                                    \
                ia[j] = off_of<AK>(d.A, m, FOLD ? ka - cla * fr : ka, oka[j]);                             \
                ia[j] = oka[j] ? ia[j] : 0;                                                                \
                ca[j] = (FOLD && oka[j]) ? (unsigned)cla * acs : 0u;                                        \
                ca2[j] = (FOLD && oka[j]) ? (unsigned)cla * acs2 : 0u;                                      \
                if constexpr (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) {                                   \
                    /* derivative column index; the value itself is applied below */                      \
                }                                                                                          \
                const int n = tn * MF_BN + B_ROW(i), kb = (KT) + B_KL(i);                                  \
                okb[j] = (n < N) & (kb < k1);                                                              \
                one[j] = okb[j] & (n == b_ones);                                                           \
                const int clb = FOLD ? (kb >= fkb ? fc0 + 1 : fc0) : 0;                                    \
                ib[j] = off_of<BKIND>(d.B, FOLD ? kb - clb * fr : kb, n, okb[j]);                          \
                ib[j] = (okb[j] & !one[j]) ? ib[j] : 0;                                                    \
                cb[j] = (FOLD && okb[j] && !one[j]) ? (unsigned)clb * bcs : 0u;                             \
            }                                                                                              \
            float xa[4], y4[4], xb[4];                                                                     \
            const bool veca = !FOLD && VEC_A && oka[0] && oka[3] && (ia[3] - ia[0] == 3);                  \
            if (veca) {                                                                                    \
                const mf_f4u t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseA + (unsigned)ia[0]);  \
                xa[0] = t[0]; xa[1] = t[1]; xa[2] = t[2]; xa[3] = t[3];                            \
                if constexpr (AK == OPK_ACTGRAD) {                                                         \
                    const mf_f4u u = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseA2 + (unsigned)ia[0]); \
                    y4[0] = u[0]; y4[1] = u[1]; y4[2] = u[2]; y4[3] = u[3];                        \
                }                                                                                          \
            } else {                                                                                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
                    xa[j] = baseA[(unsigned)ia[j] + ca[j]];                                                \
                    if constexpr (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) y4[j] = baseA2[(unsigned)ia[j] + ca2[j]]; \
                }                                                                                          \
            }                                                                                              \
            const bool vecb = !FOLD && VEC_B && okb[0] && okb[3] && (ib[3] - ib[0] == 3) && !(one[0] | one[1] | one[2] | one[3]); \
            if (vecb) {                                                                                    \
                const mf_f4u t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseB + (unsigned)ib[0]);  \
                xb[0] = t[0]; xb[1] = t[1]; xb[2] = t[2]; xb[3] = t[3];                            \
            } else {                                                                                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) xb[j] = baseB[(unsigned)ib[j] + cb[j]];      \
            }                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
                const int i = 4 * g + j;                                                                   \
                if constexpr (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) {                                   \
                    const int m = tm * MF_BM + A_ROW(i), ka = (KT) + A_KL(i);                              \
                    const int col = (AK == OPK_ACTGRAD) ? (a_tr ? m : ka) : 0;                             \
                    ya[i] = act_bwd(a_act, y4[j], col);                                                    \
                } else ya[i] = 1.f;                                                                        \
                va[i] = oka[j] ? xa[j] : 0.f;                                                              \
                vb[i] = one[j] ? 1.f : (okb[j] ? xb[j] : 0.f);                                             \
            }                                                                                              \
        }                                                                                                  \
    }
    if (k0 < k1) MF_ISSUE(k0)
    for (int kt = k0; kt < k1; kt += MF_KD) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            As[A_KL(i)][A_ROW(i)] = (AK == OPK_ACTGRAD || AK == OPK_CONVGRAD) ? va[i] * ya[i] : va[i];
            Bs[B_KL(i)][B_ROW(i)] = vb[i];
        }
        __syncthreads();
        if (kt + MF_KD < k1) MF_ISSUE(kt + MF_KD)          // next tile's loads fly under this tile's MFMAs
        if (live) {
#pragma unroll
            for (int kk = 0; kk < MF_KD / 2; ++kk) {
                const int k = kk * 2 + (lane >> 5);
                const float a = As[k][wm * 32 + (lane & 31)];
                const float bb = Bs[k][wn * 32 + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef MF_ISSUE
#undef A_ROW
#undef A_KL
#undef B_ROW
#undef B_KL
#undef VROW
#undef VKL
#undef SROW
#undef SKL
    if (!live) return;
    float* cbase = b.p[d.out.space] + d.out.off;
    const float* bias = d.out.bias_space >= 0 ? b.p[d.out.bias_space] + d.out.bias_off : nullptr;
    const int n = tn * MF_BN + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = tm * MF_BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n < N) store_out<OK>(d, cbase, bias, m, n, split, acc[r]);
    }
}

// ---- lean loader for the dense operands of the throughput flavour --------------------------------
// PMC on the generic body above (64 clips per launch): 17.7 VALU instructions per MFMA — the per-element index / predicate
// arithmetic of the accessor loaders, re-derived every k-tile, kept the vector ALU busier than the matrix cores (37 % MFMA
// busy).  Every operand of the Linear GEMMs (forward, input gradient, weight gradient, recurrent-weight gradient) is a
// strided 2-D view with ONE unit-stride direction, so here a lane owns two groups of 4 consecutive elements along that
// direction whose offsets are computed once; the k loop adds a scalar per tile and issues one 16-byte load per group
// (two for an activation-gradient operand: dY and Y).  Only groups on an edge (M / N / k1 not a multiple of 4, the
// bias-gradient ones column) take the predicated per-element path.
// A workgroup owns a RUN of consecutive output tiles of one member (GemmDesc.run, same k-split) and walks them as one
// flat sequence of (tile, k-tile) steps: the loads of the next step — also across a tile boundary — are issued before the
// current step's MFMAs, so the descriptor fetch / first-load latency is paid once per run instead of once per tile (the
// model's reductions are short: K = 64..135 is 2-5 k-tiles, and a one-tile workgroup spent most of its life waiting).
// One barrier per k-tile: the LDS tiles are double-buffered (a wave that passed barrier t + 1 has finished reading buffer
// t & 1), and the barrier is LDS-only (MST_LDS_BARRIER), so the next step's global loads stay in flight across it —
// __syncthreads() would drain them (vmcnt(0)) before every barrier.
typedef float (*mf_tile_t)[MF_BM + GEMM_PAD];
template <int AK, int OK, int AKF, int BKF>
__device__ __forceinline__ void gemm_fast_body(const GemmDesc& d, const Bases& b, const int t_begin, const int t_end, const int split,
                                               float* smem) {
    static_assert(AK == OPK_DENSE || AK == OPK_ACTGRAD, "dense operands only");
    const int tid = threadIdx.x;
    const int M = d.M, N = d.N;
    const int tiles_n = (N + MF_BN - 1) / MF_BN;
    int kchunk = (d.K + d.ksplit - 1) / d.ksplit;
    kchunk = (kchunk + MF_KD - 1) / MF_KD * MF_KD;
    const int k0 = split * kchunk;
    const int k1 = min(d.K, k0 + kchunk);
    if (k0 >= k1 || t_begin >= t_end) return;
    const int wv = tid >> 6, lane = tid & 63;
    const gcptr baseA = (gcptr)(b.p[d.A.space] + d.A.off);
    const gcptr baseY = AK == OPK_ACTGRAD ? (gcptr)(b.p[d.A.space2] + d.A.off2) : baseA;
    const gcptr baseB = (gcptr)(b.p[d.B.space] + d.B.off);
    // element strides: A(m, k) = baseA[m * sAm + k * sAk], B(k, n) = baseB[k * sBk + n * sBn]; the *KF direction is unit stride
    const int sAm = AK == OPK_DENSE ? (int)d.A.si : (d.A.transposed ? 1 : d.A.ld);
    const int sAk = AK == OPK_DENSE ? (int)d.A.sj : (d.A.transposed ? d.A.ld : 1);
    const int sBk = (int)d.B.si, sBn = (int)d.B.sj;
    const int a_act = AK == OPK_ACTGRAD ? d.A.act : ACT_NONE;
    const int a_tr = d.A.transposed;
    const int b_ones = d.B.ones_at;
    // clips folded into the reduction (weight gradients of a batched plan): k = (clip, row)
    const int fr = d.fold_rows;
    // group geometry.  KF = 1 (k is unit stride): row = (tid >> 3) + 32 g, k = 4 (tid & 7) + j.
    //                  KF = 0 (row is unit stride): k = (tid >> 4) + 16 g, row = 4 (tid & 15) + j.
    const int arow0 = AKF ? (tid >> 3) : 4 * (tid & 15), akl0 = AKF ? 4 * (tid & 7) : (tid >> 4);
    const int brow0 = BKF ? (tid >> 3) : 4 * (tid & 15), bkl0 = BKF ? 4 * (tid & 7) : (tid >> 4);
    // per-tile constants of this lane's groups: element offset of the first element at k = 0, and whether the group's rows
    // are all inside the matrix (KF = 0) / its row is (KF = 1)
    struct Geo { int tm, tn, aoff[2], boff[2]; bool avec[2], bvec[2]; bool clean; };
    auto geo_of = [&](const int t, Geo& q) {
        q.tm = t / tiles_n; q.tn = t - q.tm * tiles_n;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int mr = q.tm * MF_BM + arow0 + (AKF ? 32 * g : 0);
            q.avec[g] = AKF ? mr < M : mr + 3 < M;
            q.aoff[g] = min(mr, M - 1) * sAm + (AKF ? akl0 : 0);       // KF = 0: the k part is added per k-tile (folded clips)
            const int nr = q.tn * MF_BN + brow0 + (BKF ? 32 * g : 0);
            q.bvec[g] = BKF ? nr < N : (nr + 3 < N) & !((b_ones >= nr) & (b_ones <= nr + 3));
            q.boff[g] = min(nr, N - 1) * sBn + (BKF ? bkl0 : 0);
        }
        // clean tile (workgroup-uniform): every group of every lane is either wholly inside the matrix or wholly outside it
        // and none holds the bias-ones column — a full k-tile then needs no per-element predicates at all
        const bool a_clean = AKF || (M % 4 == 0) || ((q.tm + 1) * MF_BM <= M);
        const bool ones_in = (b_ones >= q.tn * MF_BN) & (b_ones < (q.tn + 1) * MF_BN);
        const bool b_clean = (BKF || (N % 4 == 0) || ((q.tn + 1) * MF_BN <= N)) & !ones_in;
        q.clean = a_clean & b_clean;
    };
    mf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float va[8], vy[8], vb[8];
    // folded reductions: (clip, row) of this lane's first k (group 0; both operands' row-contiguous groups sit at the same k)
    // is carried from k-tile to k-tile — a 32-bit division per group and k-tile, plus 64-bit pointer arithmetic per clip
    // stride, was most of what the waves of the skinny weight-gradient GEMMs issued.  Clip strides are 32-bit element
    // offsets off the uniform base (the plan only folds when clips * stride fits)
    const unsigned acs = (unsigned)d.acs, acs2 = (unsigned)d.acs2, bcs = (unsigned)d.bcs;
    int f_kt = -(1 << 30), f_clip = 0, f_kr = 0;
    auto fold_at = [&](const int kt, int (&clip)[2], int (&kr)[2]) {
        const int kk = kt + (tid >> 4);
        if (fr >= MF_KD && kt == f_kt + MF_KD) {
            f_kr += MF_KD;
            if (f_kr >= fr) { f_kr -= fr; ++f_clip; }
        } else { f_clip = kk / fr; f_kr = kk - f_clip * fr; }
        f_kt = kt;
        clip[0] = f_clip; kr[0] = f_kr;
        if (fr >= 16) { const bool w = f_kr + 16 >= fr; clip[1] = f_clip + (w ? 1 : 0); kr[1] = f_kr + 16 - (w ? fr : 0); }
        else { clip[1] = (kk + 16) / fr; kr[1] = kk + 16 - clip[1] * fr; }
    };
    auto issue = [&](const Geo& q, const int kt) {
        int fclip[2] = {0, 0}, fkr[2] = {0, 0};
        if ((!AKF || !BKF) && fr) fold_at(kt, fclip, fkr);
        if (q.clean & (kt + MF_KD <= k1)) {
            // fast path: a full k-tile of a clean tile.  One 16-byte load per group (two with the activation), the address is a
            // lane constant plus a per-k-tile term; a group outside the matrix (lane-invariant) holds zeros.  This is the
            // path of nearly every k-tile of the large GEMMs; the general code below cost ~300 VALU instructions per wave.
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                {
                    const int kr = AKF ? 0 : (fr ? fkr[g] : kt + akl0 + 16 * g);
                    const unsigned o = (unsigned)(q.aoff[g] + (AKF ? kt : kr * sAk)) + ((!AKF && fr) ? (unsigned)fclip[g] * acs : 0u);
                    const unsigned oy = (unsigned)(q.aoff[g] + (AKF ? kt : kr * sAk)) + ((!AKF && fr) ? (unsigned)fclip[g] * acs2 : 0u);
                    mf_f4u t = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
                    if (q.avec[g]) {
                        t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseA + o);
                        if constexpr (AK == OPK_ACTGRAD) u = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseY + oy);
                    }
                    va[4 * g] = t[0]; va[4 * g + 1] = t[1]; va[4 * g + 2] = t[2]; va[4 * g + 3] = t[3];
                    if constexpr (AK == OPK_ACTGRAD) { vy[4 * g] = u[0]; vy[4 * g + 1] = u[1]; vy[4 * g + 2] = u[2]; vy[4 * g + 3] = u[3]; }
                }
                {
                    const int kr = BKF ? 0 : (fr ? fkr[g] : kt + bkl0 + 16 * g);
                    const unsigned o = (unsigned)(q.boff[g] + (BKF ? kt : kr * sBk)) + ((!BKF && fr) ? (unsigned)fclip[g] * bcs : 0u);
                    mf_f4u t = {0.f, 0.f, 0.f, 0.f};
                    if (q.bvec[g]) t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseB + o);
                    vb[4 * g] = t[0]; vb[4 * g + 1] = t[1]; vb[4 * g + 2] = t[2]; vb[4 * g + 3] = t[3];
                }
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // ---- A
            if (AKF) {
                const int kk = kt + akl0;                              // this group's first k
                const int o = q.aoff[g] + kt;
                if (q.avec[g] & (kk + 3 < k1)) {
                    const mf_f4u t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseA + (unsigned)o);
                    va[4 * g] = t[0]; va[4 * g + 1] = t[1]; va[4 * g + 2] = t[2]; va[4 * g + 3] = t[3];
                    if constexpr (AK == OPK_ACTGRAD) {
                        const mf_f4u u = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseY + (unsigned)o);
                        vy[4 * g] = u[0]; vy[4 * g + 1] = u[1]; vy[4 * g + 2] = u[2]; vy[4 * g + 3] = u[3];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = q.avec[g] & (kk + j < k1);
                        va[4 * g + j] = ok ? baseA[(unsigned)(ok ? o + j : 0)] : 0.f;
                        if constexpr (AK == OPK_ACTGRAD) vy[4 * g + j] = ok ? baseY[(unsigned)(ok ? o + j : 0)] : 0.f;
                    }
                }
            } else {
                const int kk = kt + akl0 + 16 * g;
                const bool kok = kk < k1;
                const int kr = fr ? fkr[g] : kk;
                const unsigned ca = kok ? (unsigned)fclip[g] * acs : 0u, cy = kok ? (unsigned)fclip[g] * acs2 : 0u;
                const unsigned o = (unsigned)(q.aoff[g] + kr * sAk);
                if (q.avec[g] & kok) {
                    const mf_f4u t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseA + (o + ca));
                    va[4 * g] = t[0]; va[4 * g + 1] = t[1]; va[4 * g + 2] = t[2]; va[4 * g + 3] = t[3];
                    if constexpr (AK == OPK_ACTGRAD) {
                        const mf_f4u u = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseY + (o + cy));
                        vy[4 * g] = u[0]; vy[4 * g + 1] = u[1]; vy[4 * g + 2] = u[2]; vy[4 * g + 3] = u[3];
                    }
                } else {
                    const int mr = q.tm * MF_BM + arow0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = kok & (mr + j < M);
                        const unsigned oj = ok ? (unsigned)((mr + j) * sAm + kr * sAk) : 0u;
                        va[4 * g + j] = ok ? baseA[oj + ca] : 0.f;
                        if constexpr (AK == OPK_ACTGRAD) vy[4 * g + j] = ok ? baseY[oj + cy] : 0.f;
                    }
                }
            }
            // ---- B
            if (BKF) {
                const int kk = kt + bkl0;
                const int o = q.boff[g] + kt;
                if (q.bvec[g] & (kk + 3 < k1)) {
                    const mf_f4u t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseB + (unsigned)o);
                    vb[4 * g] = t[0]; vb[4 * g + 1] = t[1]; vb[4 * g + 2] = t[2]; vb[4 * g + 3] = t[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = q.bvec[g] & (kk + j < k1);
                        vb[4 * g + j] = ok ? baseB[(unsigned)(ok ? o + j : 0)] : 0.f;
                    }
                }
            } else {
                const int kk = kt + bkl0 + 16 * g;
                const bool kok = kk < k1;
                const int kr = fr ? fkr[g] : kk;
                const unsigned cb = kok ? (unsigned)fclip[g] * bcs : 0u;
                const unsigned o = (unsigned)(q.boff[g] + kr * sBk);
                if (q.bvec[g] & kok) {
                    const mf_f4u t = *reinterpret_cast<const MST_GLOBAL_AS mf_f4u*>(baseB + (o + cb));
                    vb[4 * g] = t[0]; vb[4 * g + 1] = t[1]; vb[4 * g + 2] = t[2]; vb[4 * g + 3] = t[3];
                } else {
                    const int nr = q.tn * MF_BN + brow0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = kok & (nr + j < N);
                        const bool one = ok & (nr + j == b_ones);
                        const float w = baseB[(ok & !one) ? (unsigned)(kr * sBk + (nr + j) * sBn) + cb : 0u];
                        vb[4 * g + j] = one ? 1.f : (ok ? w : 0.f);
                    }
                }
            }
        }
    };
    constexpr int TILE_F = MF_KD * (MF_BM + GEMM_PAD);
    float* cbase = b.p[d.out.space] + d.out.off;
    const float* bias = d.out.bias_space >= 0 ? b.p[d.out.bias_space] + d.out.bias_off : nullptr;
    Geo gl;                                  // geometry of the step whose operands sit in va / vy / vb
    geo_of(t_begin, gl);
    issue(gl, k0);
    int t = t_begin, kt = k0, buf = 0;
    while (true) {                           // workgroup-uniform
        const mf_tile_t As = reinterpret_cast<mf_tile_t>(smem + buf * 2 * TILE_F);
        const mf_tile_t Bs = reinterpret_cast<mf_tile_t>(smem + buf * 2 * TILE_F + TILE_F);
        const int tm = gl.tm, tn = gl.tn;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ar = AKF ? arow0 + 32 * g : arow0 + j, ak = AKF ? akl0 + j : akl0 + 16 * g;
                float v = va[4 * g + j];
                if constexpr (AK == OPK_ACTGRAD) {
                    const int col = a_tr ? tm * MF_BM + ar : kt + ak;
                    v *= act_bwd(a_act, vy[4 * g + j], col);
                }
                As[ak][ar] = v;
                const int br = BKF ? brow0 + 32 * g : brow0 + j, bk = BKF ? bkl0 + j : bkl0 + 16 * g;
                Bs[bk][br] = vb[4 * g + j];
            }
        }
        MST_LDS_BARRIER();
        // next step: the next k-tile of this tile, or the first k-tile of the next tile of the run
        const bool last_k = kt + MF_KD >= k1;
        const int nt = last_k ? t + 1 : t, nkt = last_k ? k0 : kt + MF_KD;
        const bool more = nt < t_end;
        if (more) {
            if (last_k) geo_of(nt, gl);
            issue(gl, nkt);                  // flies under this step's MFMAs, the epilogue and the barrier
        }
        // Skinny tiles (<= 32 live rows and / or columns: weight gradients of 16-wide layers, N = 16 projections) used to
        // run full 32x32 MFMAs on all four waves, two or three of them on padding.  The waves a tile does not need for
        // blocks take a slice of the k-tile instead: nblk = live 32x32 blocks, wave w -> block w % nblk, k-slice w / nblk;
        // the slices' accumulators meet in LDS at the tile's last k-tile, summed in slice order.
        // (clips-as-rows GEMMs never slice: their row count is the clip count, and a clip's results must not depend on how many
        // clips share the launch — every output stays one k-ascending chain whatever M is)
        const int mb = (d.clip_rows || M - tm * MF_BM > 32) ? 2 : 1, nb = (d.clip_rows || N - tn * MF_BN > 32) ? 2 : 1, nblk = mb * nb;
        const int blk = wv % nblk, ks = wv / nblk, nsl = 4 / nblk;
        const int bm = blk / nb, bn = blk - bm * nb;
        {
            const int per = (MF_KD / 2) / nsl, kk0 = ks * per;
            const int live = (min(k1 - kt, MF_KD) + 1) >> 1;         // k pairs of this k-tile that hold data (K = 10 -> 5 of 16)
            if (nsl == 1 && live == MF_KD / 2) {
#pragma unroll
                for (int kk = 0; kk < MF_KD / 2; ++kk) {
                    const int k = kk * 2 + (lane >> 5);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k][bm * 32 + (lane & 31)], Bs[k][bn * 32 + (lane & 31)], acc, 0, 0, 0);
                }
            } else {
                for (int kk = kk0; kk < min(kk0 + per, live); ++kk) {
                    const int k = kk * 2 + (lane >> 5);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k][bm * 32 + (lane & 31)], Bs[k][bn * 32 + (lane & 31)], acc, 0, 0, 0);
                }
            }
        }
        if (last_k) {
            if (nsl > 1) {                                   // workgroup-uniform (tile geometry)
                float* part = smem + buf * 2 * TILE_F;       // this step's tiles: free once every wave is past its MFMAs
                MST_LDS_BARRIER();
                if (ks > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[((ks - 1) * nblk + blk) * 1024 + r * 64 + lane] = acc[r];
                }
                MST_LDS_BARRIER();
                if (ks == 0) {
                    for (int q = 0; q < nsl - 1; ++q) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += part[(q * nblk + blk) * 1024 + r * 64 + lane];
                    }
                }
            }
            const int n = tn * MF_BN + bn * 32 + (lane & 31);
            if ((ks == 0) & (tm * MF_BM + bm * 32 < M) & (tn * MF_BN + bn * 32 < N)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = tm * MF_BM + bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < M && n < N) store_out<OK>(d, cbase, bias, m, n, split, acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        if (!more) break;
        t = nt; kt = nkt; buf ^= 1;
    }
}

__global__ __launch_bounds__(MF_THREADS, 4) void gemm_mfma_kernel(const GemmDesc* __restrict__ descs, const int* __restrict__ owner, int count, int blocks_per_clip, Bases b) {
    __shared__ float smem[2 * 2 * MF_KD * (MF_BM + GEMM_PAD)];        // two (A | B) tile pairs: the lean body double-buffers
    float (*As)[MF_BM + GEMM_PAD] = reinterpret_cast<float (*)[MF_BM + GEMM_PAD]>(smem);
    float (*Bs)[MF_BN + GEMM_PAD] = reinterpret_cast<float (*)[MF_BN + GEMM_PAD]>(smem + MF_KD * (MF_BM + GEMM_PAD));
    const int clip = blockIdx.x / blocks_per_clip, lb = blockIdx.x - clip * blocks_per_clip;
    // member of this block: one uniform load from the plan's block -> member table (a binary search over the members' first
    // blocks was log2(members) dependent scalar round trips; the linear probe before it, hundreds)
    const int lo = owner[lb];
    const GemmDesc d = descs[clip * count + lo];
    const int local = lb - d.blk_begin;
    const int ntile = ((d.M + MF_BM - 1) / MF_BM) * ((d.N + MF_BN - 1) / MF_BN);
    const int run = d.run > 0 ? d.run : 1, nruns = (ntile + run - 1) / run;
    const int split = local / nruns, t_begin = (local - split * nruns) * run, t_end = min(ntile, t_begin + run);
    switch (d.variant) {
    case GV_LIN_FWD: gemm_fast_body<OPK_DENSE, OUT_STORE, 1, 1>(d, b, t_begin, t_end, split, smem); break;
    case GV_LIN_DW: gemm_fast_body<OPK_ACTGRAD, OUT_SLAB, 0, 0>(d, b, t_begin, t_end, split, smem); break;
    case GV_LIN_DW_PERM: gemm_fast_body<OPK_ACTGRAD, OUT_PERMW_SLAB, 0, 0>(d, b, t_begin, t_end, split, smem); break;
    case GV_LIN_DA: gemm_fast_body<OPK_ACTGRAD, OUT_ACCUM, 1, 0>(d, b, t_begin, t_end, split, smem); break;
    case GV_HH_DW: gemm_fast_body<OPK_DENSE, OUT_SLAB, 0, 0>(d, b, t_begin, t_end, split, smem); break;
    case GV_LIN_DA_RAW: gemm_fast_body<OPK_DENSE, OUT_ACCUM, 1, 0>(d, b, t_begin, t_end, split, smem); break;
    default:
        for (int tile = t_begin; tile < t_end; ++tile) {
            if (tile > t_begin) __syncthreads();
            if (d.variant == GV_LIN_FWD_PERM) gemm_mfma_body<OPK_DENSE, OPK_PERMW, OUT_STORE, 1, 1>(d, b, tile, split, As, Bs);
            else if (d.variant == GV_CONV_FWD) gemm_mfma_body<OPK_IM2COL, OPK_PERMW, OUT_CONV, 1, 1>(d, b, tile, split, As, Bs);
            else if (d.variant == GV_CONV_DW) {
                if (d.fold_rows) gemm_mfma_body<OPK_CONVGRAD, OPK_IM2COL, OUT_PERMW_SLAB, 1, 0, true>(d, b, tile, split, As, Bs);
                else gemm_mfma_body<OPK_CONVGRAD, OPK_IM2COL, OUT_PERMW_SLAB, 1, 0>(d, b, tile, split, As, Bs);
            }
        }
        break;
    }
}

int gemm_tile_edge(int mfma) { return mfma ? MF_BM : GEMM_BM; }
int gemm_blocks(const GemmDesc& g, int mfma) {
    const int edge = gemm_tile_edge(mfma);
    const int ntile = ((g.M + edge - 1) / edge) * ((g.N + edge - 1) / edge);
    const int run = (mfma && g.run > 1) ? g.run : 1;
    return ((ntile + run - 1) / run) * g.ksplit;
}

// One kernel for every GEMM of the model: blockIdx.y picks the descriptor, the descriptor's
// (workgroup-uniform) variant picks the instantiation.  That lets the scheduler put *independent*
// GEMMs of different kinds — e.g. the weight-gradient and input-gradient GEMMs of one layer, or
// all small Linears of one dependency level — into a single launch.
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(const GemmDesc* __restrict__ descs, const int* __restrict__ owner, int count, int blocks_per_clip, Bases b) {
    // A tile | B tile; the final reduce overlays the whole block with one 32x32 partial tile per wave
    constexpr int TILE_F = GEMM_BK * (GEMM_BM + GEMM_PAD), RED_F = (GEMM_THREADS / 64) * GEMM_BM * GEMM_BN;
    __shared__ float smem[(2 * TILE_F > RED_F) ? 2 * TILE_F : RED_F];
    float (*As)[GEMM_BM + GEMM_PAD] = reinterpret_cast<float (*)[GEMM_BM + GEMM_PAD]>(smem);
    float (*Bs)[GEMM_BN + GEMM_PAD] = reinterpret_cast<float (*)[GEMM_BN + GEMM_PAD]>(smem + TILE_F);
    // flat 1-D grid: clip-major; inside a clip's block range member y owns [blk_begin, blk_begin + tiles * ksplit)
    const int clip = blockIdx.x / blocks_per_clip, lb = blockIdx.x - clip * blocks_per_clip;
    const int lo = owner[lb];                        // member of this block: one uniform load (the plan's block -> member table)
    const GemmDesc d = descs[clip * count + lo];     // by value (scalar loads once): a reference would be re-read after every barrier
    const int local = lb - d.blk_begin;
    const int ntile = ((d.M + GEMM_BM - 1) / GEMM_BM) * ((d.N + GEMM_BN - 1) / GEMM_BN);
    const int tile = local % ntile, split = local / ntile;
#define GEMM_CASE(V, A_, B_, O_, AF_, BF_)                                                          \
    case (V) * 3 + 0: gemm_body<A_, B_, O_, AF_, BF_, 32>(d, b, tile, split, As, Bs); break;           \
    case (V) * 3 + 1: gemm_body<A_, B_, O_, AF_, BF_, 64>(d, b, tile, split, As, Bs); break;           \
    case (V) * 3 + 2: gemm_body<A_, B_, O_, AF_, BF_, 128>(d, b, tile, split, As, Bs); break;
    switch (d.variant * 3 + d.kdsel) {
        GEMM_CASE(GV_LIN_FWD, OPK_DENSE, OPK_DENSE, OUT_STORE, 1, 1)
        GEMM_CASE(GV_LIN_FWD_PERM, OPK_DENSE, OPK_PERMW, OUT_STORE, 1, 1)
        GEMM_CASE(GV_LIN_DW, OPK_ACTGRAD, OPK_DENSE, OUT_SLAB, 0, 0)
        GEMM_CASE(GV_LIN_DW_PERM, OPK_ACTGRAD, OPK_DENSE, OUT_PERMW_SLAB, 0, 0)
        GEMM_CASE(GV_LIN_DA, OPK_ACTGRAD, OPK_DENSE, OUT_ACCUM, 1, 0)
        GEMM_CASE(GV_CONV_FWD, OPK_IM2COL, OPK_PERMW, OUT_CONV, 1, 1)
        GEMM_CASE(GV_CONV_DW, OPK_CONVGRAD, OPK_IM2COL, OUT_PERMW_SLAB, 1, 0)
        GEMM_CASE(GV_HH_DW, OPK_DENSE, OPK_DENSE, OUT_SLAB, 0, 0)
        GEMM_CASE(GV_LIN_DA_RAW, OPK_DENSE, OPK_DENSE, OUT_ACCUM, 1, 0)
    default: break;
    }
#undef GEMM_CASE
}

int gemm_variant(const GemmDesc& g) {
    const int a = g.A.kind, bk = g.B.kind, o = g.out.kind;
    if (a == OPK_DENSE && bk == OPK_DENSE && o == OUT_STORE) return GV_LIN_FWD;
    if (a == OPK_DENSE && bk == OPK_PERMW && o == OUT_STORE) return GV_LIN_FWD_PERM;
    if (a == OPK_ACTGRAD && bk == OPK_DENSE && o == OUT_SLAB) return GV_LIN_DW;
    if (a == OPK_ACTGRAD && bk == OPK_DENSE && o == OUT_PERMW_SLAB) return GV_LIN_DW_PERM;
    if (a == OPK_ACTGRAD && bk == OPK_DENSE && o == OUT_ACCUM) return GV_LIN_DA;
    if (a == OPK_IM2COL && bk == OPK_PERMW && o == OUT_CONV) return GV_CONV_FWD;
    if (a == OPK_CONVGRAD && bk == OPK_IM2COL && o == OUT_PERMW_SLAB) return GV_CONV_DW;
    if (a == OPK_DENSE && bk == OPK_DENSE && o == OUT_SLAB) return GV_HH_DW;
    if (a == OPK_DENSE && bk == OPK_DENSE && o == OUT_ACCUM) return GV_LIN_DA_RAW;
    return -1;
}

int launch_gemm(const GemmDesc* dev_descs, const int* dev_owner, int members, int blocks_per_clip, int clips, int mfma, Bases b, hipStream_t s) {
    if (members <= 0 || blocks_per_clip <= 0 || clips <= 0) return 0;
    if (mfma) hipLaunchKernelGGL(gemm_mfma_kernel, dim3(blocks_per_clip * clips), dim3(MF_THREADS), 0, s, dev_descs, dev_owner, members, blocks_per_clip, b);
    else hipLaunchKernelGGL(gemm_kernel, dim3(blocks_per_clip * clips), dim3(GEMM_THREADS), 0, s, dev_descs, dev_owner, members, blocks_per_clip, b);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Gather: materialise cat_with_broadcast (style/utils/pytorch.py:54-65) once for the Linears that
// consume it — out[row, start_s + w] = seg_s[index_s(row), w], where index_s drops the row-space
// dims the segment is broadcast over.  4 rows per workgroup (one wave each), lanes along the
// concatenated feature axis (coalesced stores, broadcast/coalesced loads).  Its backward is the
// segment reduce below applied to the gradient of the materialised tensor.
__global__ __launch_bounds__(256) void gather_kernel(const GatherDesc* __restrict__ descs, Bases b) {
    const GatherDesc& d = descs[blockIdx.y];       // no barriers here; dynamic seg[] indexing wants it in memory
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wv; row < d.rows; row += gridDim.x * 4) {
        unsigned t = (unsigned)row;
        int rc[4];
        rc[3] = t % (unsigned)d.d[3]; t /= (unsigned)d.d[3];
        rc[2] = t % (unsigned)d.d[2]; t /= (unsigned)d.d[2];
        rc[1] = t % (unsigned)d.d[1]; rc[0] = t / (unsigned)d.d[1];
        float* out = b.p[SP_WS] + d.out_off + (int64_t)row * d.K;
        if (d.sum) {                                    // broadcast sum: out[row, w] = sum over the segments, in segment order
            for (int w = lane; w < d.K; w += 64) {
                float a = 0.f;
                for (int sIdx = 0; sIdx < d.nseg; ++sIdx) {
                    const Seg& sg = d.seg[sIdx];
                    const int srow = rc[0] * sg.s[0] + rc[1] * sg.s[1] + rc[2] * sg.s[2] + rc[3] * sg.s[3];
                    a += b.p[sg.space][sg.off + (int64_t)srow * sg.ld + w];
                }
                out[w] = a;
            }
            continue;
        }
        for (int sIdx = 0; sIdx < d.nseg; ++sIdx) {
            const Seg& sg = d.seg[sIdx];
            const int srow = rc[0] * sg.s[0] + rc[1] * sg.s[1] + rc[2] * sg.s[2] + rc[3] * sg.s[3];
            const float* src = b.p[sg.space] + sg.off + (int64_t)srow * sg.ld;
            for (int w = lane; w < sg.width; w += 64) out[sg.start + w] = src[w];
        }
    }
}

int launch_gather(const GatherDesc* dev, int count, int max_rows, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    int gx = (max_rows + 3) / 4;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(gather_kernel, dim3(gx, count), dim3(256), 0, s, dev, b);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Segment reduce: the gradient of a broadcast segment of a concatenated input is the sum, over
// the broadcast row-space dims, of its columns of dAcat (the backward of cat_with_broadcast's
// expand, style/utils/pytorch.py:60-63).  Stage 1: one workgroup per (destination row, chunk of
// 64 reduced rows) — 4 waves split the rows, 64 lanes walk the segment's columns (coalesced).
// Segments reduced over <= 64 rows finish here (dst +=); wider ones leave per-chunk partials that
// stage 2 sums in chunk order (deterministic; no float atomics).
#define SEGRED_CHUNK 64
// Stage 1, one lane per (destination row, chunk, column): the lane walks its chunk's reduced rows with an
// odometer over the reduced dims (no divisions in the loop), eight loads in flight, and adds them in row
// order.  No LDS, no barriers, and a workgroup always has 256 outputs' worth of work (a workgroup per
// destination row left most lanes idle: the broadcast segments are 5-48 columns wide).
// E consecutive columns per lane (the index arithmetic — a dozen integer divisions — is then paid once per E outputs), R reduced
// rows in flight per round trip.  Every output is still its rows added in row order.
template <int E, int R>
__device__ __forceinline__ void segred_body(const SegRedDesc& d, const int item, const Bases& b) {
    const int wgrp = (d.width + E - 1) / E;
    if (item >= d.nidx * d.nchunk * wgrp) return;
    const int w = (item % wgrp) * E, ne = min(E, d.width - w);
    int t = item / wgrp;
    const int chunk = t % d.nchunk, idx = t / d.nchunk;
    int kc[4];
    t = idx;
#pragma unroll
    for (int q = 3; q >= 0; --q) { kc[q] = t % d.kd[q]; t /= d.kd[q]; }
    int rd[4], nred = 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) { rd[q] = d.kd[q] == 1 ? d.d[q] : 1; nred *= rd[q]; }
    const int r_begin = chunk * SEGRED_CHUNK, r_end = min(nred, r_begin + SEGRED_CHUNK);
    int cr[4];
    t = r_begin;
#pragma unroll
    for (int q = 3; q >= 0; --q) { cr[q] = t % rd[q]; t /= rd[q]; }
    const float* src = b.p[SP_GRAD] + d.src_off + d.start + w;
    const float* ysrc = b.p[SP_WS] + d.y_off + d.start + w;          // d.act: the activation whose derivative scales src
    const int64_t s3 = d.src_ld, s2 = s3 * d.d[3], s1 = s2 * d.d[2], s0 = s1 * d.d[1];
    auto offset = [&]() { return (kc[0] + cr[0]) * s0 + (kc[1] + cr[1]) * s1 + (kc[2] + cr[2]) * s2 + (kc[3] + cr[3]) * s3; };
    auto advance = [&]() {
        if (++cr[3] == rd[3]) { cr[3] = 0; if (++cr[2] == rd[2]) { cr[2] = 0; if (++cr[1] == rd[1]) { cr[1] = 0; ++cr[0]; } } }
    };
    float acc[E];
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] = 0.f;
    // a full group of four is ONE 16-byte load (rows are only 4-byte aligned: the 4-byte-aligned vector type); the ragged last group
    // of a row loads its live columns one by one
    typedef float sr_f4 __attribute__((ext_vector_type(4), aligned(4)));
    auto ldE = [&](const float* p, float (&dst)[E]) {
        if (E == 4 && ne == 4) {
            const sr_f4 t = *reinterpret_cast<const sr_f4*>(p);
#pragma unroll
            for (int j = 0; j < E; ++j) dst[j] = t[j < 4 ? j : 0];
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) dst[j] = p[j < ne ? j : 0];
        }
    };
    int rr = r_begin;
    for (; rr + R <= r_end; rr += R) {                 // R x E loads in flight per round trip, added in row order
        int64_t o[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { o[q] = offset(); advance(); }
        float v[R][E];
#pragma unroll
        for (int q = 0; q < R; ++q) ldE(src + o[q], v[q]);
        if (d.act != ACT_NONE) {
            float y[R][E];
#pragma unroll
            for (int q = 0; q < R; ++q) ldE(ysrc + o[q], y[q]);
#pragma unroll
            for (int q = 0; q < R; ++q)
#pragma unroll
                for (int j = 0; j < E; ++j) v[q][j] *= act_bwd(d.act, y[q][j], d.start + w + j);
        }
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int j = 0; j < E; ++j) acc[j] += v[q][j];
    }
    for (; rr < r_end; ++rr) {
        const int64_t o = offset();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int jj = j < ne ? j : 0;
            float v = src[o + jj];
            if (d.act != ACT_NONE) v *= act_bwd(d.act, ysrc[o + jj], d.start + w + j);
            acc[j] += v;
        }
        advance();
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (j >= ne) break;
        if (d.nchunk == 1) { float* dst = b.p[SP_GRAD] + d.dst_off + (int64_t)idx * d.dst_ld + w + j; *dst = d.first ? acc[j] : *dst + acc[j]; }
        else b.p[SP_TMP][d.part_off + ((int64_t)idx * d.nchunk + chunk) * d.width + w + j] = acc[j];
    }
}

__global__ __launch_bounds__(256) void segred_kernel(const SegRedDesc* __restrict__ descs, int members, int blocks_per_clip, Bases b) {
    // flat grid, clip-major: inside a clip's block range member y owns [blk_begin, blk_begin + ceil(nidx*nchunk*width / 256))
    // (sized for one column per lane; a wide member uses the first quarter of its workgroups, the others retire at once)
    const int clip = blockIdx.x / blocks_per_clip, lb = blockIdx.x - clip * blocks_per_clip;
    int y = 0;
    while (y + 1 < members && lb >= descs[y + 1].blk_begin) ++y;
    const SegRedDesc d = descs[clip * members + y];
    const int e = (lb - d.blk_begin) * 256 + threadIdx.x;
    // four columns per lane only where nothing is reduced (a strided copy: index arithmetic per element was all its time, 68.7 ->
    // 40.6 us for the 456-wide one at 64 clips); a member that sums rows is latency bound and wants every lane it can get
    // (four columns per lane there: 33.6 -> 77.5 us)
    const bool copy = (d.kd[0] == 1 ? d.d[0] : 1) * (d.kd[1] == 1 ? d.d[1] : 1) * (d.kd[2] == 1 ? d.d[2] : 1) * (d.kd[3] == 1 ? d.d[3] : 1) == 1;
    if (copy && d.width >= 64) segred_body<4, 1>(d, e, b);
    else segred_body<1, 8>(d, e, b);
}

__global__ __launch_bounds__(256) void segred2_kernel(const SegRedDesc* __restrict__ descs, Bases b) {
    const SegRedDesc d = descs[blockIdx.y];
    if (d.nchunk == 1) return;
    const int total = d.nidx * d.width;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int idx = e / d.width, w = e - idx * d.width;
        const float* pp = b.p[SP_TMP] + d.part_off + (int64_t)idx * d.nchunk * d.width + w;
        float acc = 0.f;
        for (int c = 0; c < d.nchunk; ++c) acc += pp[(int64_t)c * d.width];
        float* dst = b.p[SP_GRAD] + d.dst_off + (int64_t)idx * d.dst_ld + w;
        *dst = d.first ? acc : *dst + acc;
    }
}

int launch_segred(const SegRedDesc* dev_descs, int members, int blocks_per_clip, int clips, int stage2_blocks, Bases b, hipStream_t s) {
    const int count = members * clips;
    if (count <= 0 || blocks_per_clip <= 0) return 0;
    hipLaunchKernelGGL(segred_kernel, dim3(blocks_per_clip * clips), dim3(256), 0, s, dev_descs, members, blocks_per_clip, b);
    if (stage2_blocks > 0) hipLaunchKernelGGL(segred2_kernel, dim3(stage2_blocks, count), dim3(256), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Deferred weight-gradient reduction: gpar[dst+i] += sum_s slab_s[i], splits summed in index
// order (bitwise reproducible, unlike float atomics).  p.grad accumulates across iterations
// exactly like loss.backward() does in train-model.py:126.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabEntry* __restrict__ ents, const SlabBlock* __restrict__ blocks, Bases b) {
    const SlabBlock blk = blocks[blockIdx.x];       // 64 consecutive elements of one entry
    const SlabEntry e = ents[blk.entry];
    // rows of an entry = (clip, k-split / producer workgroup) pairs, clip-major.  Lanes walk 64 consecutive elements
    // (256-byte rows); the four waves take a contiguous quarter of the rows each, then the quarters are added in order.
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int rows = e.reps * e.splits;
    const int r0 = (int)((int64_t)rows * g / 4), r1 = (int)((int64_t)rows * (g + 1) / 4);
    const int i = blk.start + lane;
    // 32 loads in flight per lane (four batches of the 8 chains): entries with ~1000 rows (per-clip, per-workgroup slabs of the
    // note kernels in a 64-clip plan) are a chain of rows / 4 / 32 memory round trips
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < e.count) {
        const MST_GLOBAL_AS float* src = (const MST_GLOBAL_AS float*)(b.p[SP_TMP] + e.src + i);
        int rep = r0 / e.splits, sp = r0 - rep * e.splits;
        int r = r0;
        for (; r + 32 <= r1; r += 32) {
            int64_t o[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) { o[q] = (int64_t)rep * e.rep_stride + (int64_t)sp * e.stride; if (++sp == e.splits) { sp = 0; ++rep; } }
            float v[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q] = src[o[q]];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] += v[8 * h + q];
            }
        }
        for (; r + 8 <= r1; r += 8) {
            int64_t o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { o[q] = (int64_t)rep * e.rep_stride + (int64_t)sp * e.stride; if (++sp == e.splits) { sp = 0; ++rep; } }
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[o[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += v[q];
        }
        for (; r < r1; ++r) { a[0] += src[(int64_t)rep * e.rep_stride + (int64_t)sp * e.stride]; if (++sp == e.splits) { sp = 0; ++rep; } }
    }
    const float a0 = a[0] + a[4], a1 = a[1] + a[5], a2 = a[2] + a[6], a3 = a[3] + a[7];
    part[g][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < e.count) {
        const int64_t di = e.width > 0 ? (int64_t)(i / e.width) * e.dst_ld + i % e.width : i;      // block of a wider matrix
        b.p[SP_GPAR][e.dst + di] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    }
}

__global__ __launch_bounds__(256) void zero_kernel(const ZeroChunk* __restrict__ ch, float* g, int64_t clip_stride) {
    const ZeroChunk c = ch[blockIdx.x];
    float* p = g + (int64_t)blockIdx.y * clip_stride + c.off;
    for (int i = threadIdx.x; i < c.len; i += 256) p[i] = 0.f;
}

int launch_zero(const ZeroChunk* dev, int nchunks, int clips, float* grad_base, int64_t clip_stride, hipStream_t s) {
    if (nchunks <= 0) return 0;
    hipLaunchKernelGGL(zero_kernel, dim3(nchunks, clips), dim3(256), 0, s, dev, grad_base, clip_stride);
    return (int)hipGetLastError();
}

int launch_slab_reduce(const SlabEntry* dev, const SlabBlock* blocks, int nblocks, Bases b, hipStream_t s) {
    if (nblocks <= 0) return 0;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(nblocks), dim3(256), 0, s, dev, blocks, b);
    return (int)hipGetLastError();
}
