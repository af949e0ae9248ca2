// nn.Linear (+ activation) of BATCHED plans on the 2 x 2-blocked MFMA tile structure of conv.hip / audio.hip:
//   forward   Y = act(X W^T + b)                        (style/model.py: every nn.Linear over a dense activation / input tensor)
//   dX        dX (+)= (dY o act'(Y)) W
//   dW | db   (dY o act'(Y))^T [X | 1], the clips' rows folded into ONE reduction, k-split into parameter-layout slabs
// The generic accessor GEMM (gemm.hip, 64 x 64 tiles, one 32 x 32 MFMA block per wave) stays for everything small, permuted or
// broadcast-decomposed; the plan routes a Linear here when it is dense and large (plan.hip, linear()).  Per workgroup: 256 lanes =
// WM x WN waves of 64 x 64 outputs (four f32 32x32x2 accumulators: each staged operand element feeds two MFMAs), 32-deep
// k-tiles k-major in LDS (single buffer, two LDS-only barriers per k-tile, the next k-tile's global loads in flight under the
// MFMAs), 16-byte loads along every operand's unit-stride direction, ragged ends (K = 82, 514: not multiples of 4) handled by
// loading the row's LAST four elements and shifting — never a byte outside the row, no per-element branches.  All clips are rows
// of one launch: row m = (clip, r), the clip's arena stride added per row.  Every output element is one k-ascending exact-f32
// chain (same association as the accessor GEMM without k-slices); weight-gradient splits meet in split order (slab_reduce).
#include <type_traits>

#include "mst_common.h"

typedef mst_f32x16 ln_f32x16;
typedef float ln_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const MST_GLOBAL_AS float* ln_gp;
#define LN_KT 32

__device__ __forceinline__ float ln_act_fwd(int act, float z, int col) {
    if (act == ACT_LEAKY) return z > 0.f ? z : z * LEAKY;
    if (act == ACT_SIGOUT) { const float s = 1.f / (1.f + expf(-z)); return col == 0 ? 6.f * s : s; }
    if (act == ACT_BPM) { const float s = 1.f / (1.f + expf(-z)); return s * 150.f + 50.f; }
    return z;
}
__device__ __forceinline__ float ln_act_bwd(int act, float y, int col) {
    if (act == ACT_LEAKY) return y > 0.f ? 1.f : LEAKY;
    if (act == ACT_SIGOUT) return col == 0 ? y * (1.f - y * (1.f / 6.f)) : y * (1.f - y);
    if (act == ACT_BPM) { const float s = (y - 50.f) * (1.f / 150.f); return 150.f * s * (1.f - s); }
    return 1.f;
}

// four consecutive elements [i, i + 4) of a row of `len` elements starting at p; elements at or beyond len read as 0.  A group
// that straddles the end loads the row's last four elements and shifts (len >= 4); nothing outside the row is touched.
// Split in two so that the load can fly under a k-tile's MFMAs: ln_ld4 issues it (raw registers), ln_fix4 applies the shift when
// the k-tile is committed to LDS — a select on the loaded value right after the load would park the wave on the load's arrival
// before its MFMAs start.
__device__ __forceinline__ ln_f4 ln_ld4(ln_gp p, int i, int len, bool on) {
    int sft = i + 4 - len;
    sft = sft < 0 ? 0 : (sft > 4 ? 4 : sft);
    const bool ok = on && sft < 4;
    return *reinterpret_cast<const MST_GLOBAL_AS ln_f4*>(p + (ok ? i - sft : 0));
}
// (bit masks, not ?: chains: the compiler turns chained selects on freshly loaded values into nested exec-mask branches with a
// wait on the load in each; masks become v_bfi / v_and.)  A two-stage barrel shift by sft in 0..3, then the validity mask.
__device__ __forceinline__ unsigned ln_bfi(unsigned m, unsigned a, unsigned b) { return (a & m) | (b & ~m); }
__device__ __forceinline__ ln_f4 ln_fix4(const ln_f4 t, int i, int len, bool on) {
    int sft = i + 4 - len;
    sft = sft < 0 ? 0 : (sft > 4 ? 4 : sft);
    const unsigned ok = (on && sft < 4) ? 0xffffffffu : 0u;
    const unsigned m1 = (sft & 1) ? 0xffffffffu : 0u, m2 = (sft & 2) ? 0xffffffffu : 0u;
    const unsigned t0 = __float_as_uint(t[0]), t1 = __float_as_uint(t[1]), t2 = __float_as_uint(t[2]), t3 = __float_as_uint(t[3]);
    const unsigned u0 = ln_bfi(m1, t1, t0), u1 = ln_bfi(m1, t2, t1), u2 = ln_bfi(m1, t3, t2), u3 = t3 & ~m1;
    ln_f4 v;
    v[0] = __uint_as_float(ln_bfi(m2, u2, u0) & ok);
    v[1] = __uint_as_float(ln_bfi(m2, u3, u1) & ok);
    v[2] = __uint_as_float(u2 & ~m2 & ok);
    v[3] = __uint_as_float(u3 & ~m2 & ok);
    return v;
}

// MODE 0: forward   A(m, k) = X[m][k]            (k unit stride), B(k, n) = W[n][k]        (k unit stride)  -> Y[m][n]
// MODE 1: dX        A(m, k) = dYa[m][k = n]      (k unit stride), B(k, j) = W[k = n][j]    (j unit stride)  -> dX[m][j]
template <int WM, int WN, int MODE>
__global__ __launch_bounds__(256, 2) void lin_rows_kernel(const LinDesc d, Bases b, int first) {
    constexpr int BM = 64 * WM, BN = 64 * WN, PA = BM + 1, PB = BN + (MODE == 0 ? 1 : 4);
    constexpr int QA = BM / 32, QB = MODE == 0 ? BN / 32 : BN / 32;      // 16-byte groups per lane and operand
    __shared__ float As[LN_KT][PA];
    __shared__ __attribute__((aligned(16))) float Bs[LN_KT][PB];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wv = MST_UNIFORM(tid >> 6), wm = wv / WN, wn = wv - wm * WN;
    const int KR = MODE == 0 ? d.K : d.N;                     // reduction length
    const int NC = MODE == 0 ? d.N : d.K;                     // output columns
    const int64_t Mtot = (int64_t)d.clips * d.rows;
    const int tiles_n = (NC + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const ln_gp xa = (ln_gp)(MODE == 0 ? b.p[d.x_space] + d.x_off : b.p[SP_GRAD] + d.y_off);       // A rows
    const ln_gp ya = (ln_gp)(b.p[SP_WS] + d.y_off);                                                  // MODE 1: the activations for act'
    const ln_gp w = (ln_gp)(b.p[SP_PAR] + d.w_off);
    const int a_ld = MODE == 0 ? d.x_ld : d.y_ld;
    const int64_t a_cs = MODE == 0 ? d.x_cs : d.y_cs;
    // A: lane -> (row = tid / 8 + 32 q, k group 4 (tid % 8)); the row's element offset is fixed for the whole tile
    unsigned aoff[QA];
    bool aon[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int64_t m = m0 + (tid >> 3) + 32 * q;
        aon[q] = m < Mtot;
        const unsigned mm = aon[q] ? (unsigned)m : 0u;       // (clips x rows < 2^31: checked by the plan)
        const unsigned clip = mm / (unsigned)d.rows, r = mm - clip * (unsigned)d.rows;
        aoff[q] = (unsigned)((int64_t)clip * a_cs + (int64_t)r * a_ld);
    }
    ln_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    ln_f4 ra[QA], ry[MODE == 1 ? QA : 1], rb[QB];            // raw loads of the k-tile in flight
    const int act = d.act;
    auto issue = [&](const int kt) {
        const int ka = kt + 4 * (tid & 7);
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            ra[q] = ln_ld4(xa + aoff[q], ka, KR, aon[q]);
            if (MODE == 1) ry[q] = ln_ld4(ya + aoff[q], ka, KR, aon[q]);
        }
        if (MODE == 0) {                                      // W[n][k]: lane -> (n = tid / 8 + 32 q, k group)
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int n = n0 + (tid >> 3) + 32 * q;
                rb[q] = ln_ld4(w + (unsigned)((n < d.N ? n : 0) * d.K), ka, d.K, n < d.N);
            }
        } else {                                              // W[k = n][j]: lane -> (k = tid / (BN / 4) + (1024 / BN) q, j group)
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int kk = kt + tid / (BN / 4) + (1024 / BN) * q, j = n0 + 4 * (tid % (BN / 4));
                rb[q] = ln_ld4(w + (unsigned)((kk < d.N ? kk : 0) * d.K), j, d.K, kk < d.N);
            }
        }
    };
    issue(0);
    for (int kt = 0; kt < KR; kt += LN_KT) {
        const int ka = kt + 4 * (tid & 7);
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            ln_f4 v = ln_fix4(ra[q], ka, KR, aon[q]);
            if (MODE == 1) {                                  // dYa = dY o act'(Y)
                const ln_f4 y = ln_fix4(ry[q], ka, KR, aon[q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= ln_act_bwd(act, y[j], ka + j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) As[4 * (tid & 7) + j][(tid >> 3) + 32 * q] = v[j];
        }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int n = n0 + (tid >> 3) + 32 * q;
                const ln_f4 v = ln_fix4(rb[q], ka, d.K, n < d.N);
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[4 * (tid & 7) + j][(tid >> 3) + 32 * q] = v[j];
            }
        } else {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int kk = kt + tid / (BN / 4) + (1024 / BN) * q, j = n0 + 4 * (tid % (BN / 4));
                const ln_f4 v = ln_fix4(rb[q], j, d.K, kk < d.N);
                float* dst = &Bs[tid / (BN / 4) + (1024 / BN) * q][4 * (tid % (BN / 4))];
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
        }
        MST_LDS_BARRIER();
        if (kt + LN_KT < KR) issue(kt + LN_KT);
        mst_mfma_ktile_2x2<LN_KT>(As, Bs, wm * 64, wn * 64, l31, kh, acc);
        MST_LDS_BARRIER();
    }
    // epilogue
    const float* bias = b.p[SP_PAR] + d.b_off;
    float* out = MODE == 0 ? b.p[SP_WS] + d.y_off : b.p[SP_GRAD] + d.x_off;
    const int o_ld = MODE == 0 ? d.y_ld : d.x_ld;
    const int64_t o_cs = MODE == 0 ? d.y_cs : d.gx_cs;
    // a lane's 32 rows in ascending order: one division for the first, then (clip, row) advance by the row deltas
    const unsigned mb = (unsigned)m0 + (unsigned)(wm * 64 + 4 * kh);
    unsigned eclip = mb / (unsigned)d.rows, erow = mb - eclip * (unsigned)d.rows, eprev = 0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned delta = 32 * a + (r & 3) + 8 * (r >> 2);      // ascending in (a, r)
            erow += delta - eprev; eprev = delta;
            { const bool wr = erow >= (unsigned)d.rows; erow -= wr ? (unsigned)d.rows : 0u; eclip += wr ? 1u : 0u; }   // rows >= 32: one wrap at most
            if ((int64_t)mb + delta >= Mtot) continue;
            float* orow = out + (int64_t)eclip * o_cs + (int64_t)erow * o_ld;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int n = n0 + wn * 64 + 32 * c + l31;
                if (n < NC) {
                    if (MODE == 0) orow[n] = ln_act_fwd(act, acc[a][c][r] + bias[n], n);
                    else orow[n] = first ? acc[a][c][r] : orow[n] + acc[a][c][r];
                }
            }
        }
}

// dW | db: A(n, k = m) = dYa[m][n] (n unit stride), B(k = m, j) = [X | 1][m][j] (j unit stride); rows m of all clips, k-split
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void lin_dw_kernel(const LinDesc d, Bases b) {
    constexpr int BM = 64 * WM, BN = 64 * WN, PA = BM + 4, PB = BN + 4;
    constexpr int QA = BM / 32, QB = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[LN_KT][PA];
    __shared__ __attribute__((aligned(16))) float Bs[LN_KT][PB];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wv = MST_UNIFORM(tid >> 6), wm = wv / WN, wn = wv - wm * WN;
    const int NB = d.K + 1;                                   // columns: the input features, then the bias column
    const int tiles_n = (NB + BN - 1) / BN, tiles_m = (d.N + BM - 1) / BM, ntile = tiles_m * tiles_n;
    const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int n0 = tm * BM, j0 = tn * BN;
    const int64_t Mtot = (int64_t)d.clips * d.rows;
    const int64_t k0 = (int64_t)split * d.rows_per_split, k1 = min(Mtot, k0 + d.rows_per_split);
    const ln_gp gy = (ln_gp)(b.p[SP_GRAD] + d.y_off), yy = (ln_gp)(b.p[SP_WS] + d.y_off), xx = (ln_gp)(b.p[d.x_space] + d.x_off);
    // A: lane -> (k row = tid / (BM / 4) + (1024 / BM) q, n group 4 (tid % (BM / 4)));  B: the same with BN and j
    const int an = n0 + 4 * (tid % (BM / 4)), bj = j0 + 4 * (tid % (BN / 4));
    const int act = d.act;
    ln_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    ln_f4 ra[QA], ry[QA], rb[QB];                            // raw loads of the k-tile in flight
    // (clip, row inside the clip) of the lane's first A row and first B row of the k-tile, advanced by 32 rows per k-tile: no
    // division in the loop
    unsigned ac, ar, bc, br_;
    {
        const unsigned fa = (unsigned)k0 + (unsigned)(tid / (BM / 4)), fb = (unsigned)k0 + (unsigned)(tid / (BN / 4));
        ac = fa / (unsigned)d.rows; ar = fa - ac * (unsigned)d.rows;
        bc = fb / (unsigned)d.rows; br_ = fb - bc * (unsigned)d.rows;
    }
    auto issue = [&](const int64_t kt) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const int64_t m = kt + tid / (BM / 4) + (1024 / BM) * q;
            const bool on = m < k1;
            unsigned clip = ac, r = ar + (unsigned)((1024 / BM) * q);
            { const bool wr = r >= (unsigned)d.rows; r -= wr ? (unsigned)d.rows : 0u; clip += wr ? 1u : 0u; }           // rows >= 32: one wrap at most
            if (!on) { clip = 0; r = 0; }
            const unsigned o = (unsigned)((int64_t)clip * d.y_cs + (int64_t)r * d.y_ld);
            ra[q] = ln_ld4(gy + o, an, d.N, on);
            ry[q] = ln_ld4(yy + o, an, d.N, on);
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int64_t m = kt + tid / (BN / 4) + (1024 / BN) * q;
            const bool on = m < k1;
            unsigned clip = bc, r = br_ + (unsigned)((1024 / BN) * q);
            { const bool wr = r >= (unsigned)d.rows; r -= wr ? (unsigned)d.rows : 0u; clip += wr ? 1u : 0u; }           // rows >= 32: one wrap at most
            if (!on) { clip = 0; r = 0; }
            rb[q] = ln_ld4(xx + (unsigned)((int64_t)clip * d.x_cs + (int64_t)r * d.x_ld), bj, d.K, on);
        }
        ar += LN_KT; if (ar >= (unsigned)d.rows) { ar -= (unsigned)d.rows; ++ac; }
        br_ += LN_KT; if (br_ >= (unsigned)d.rows) { br_ -= (unsigned)d.rows; ++bc; }
    };
    if (k0 < k1) issue(k0);
    for (int64_t kt = k0; kt < k1; kt += LN_KT) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const bool on = kt + tid / (BM / 4) + (1024 / BM) * q < k1;
            ln_f4 v = ln_fix4(ra[q], an, d.N, on);
            const ln_f4 y = ln_fix4(ry[q], an, d.N, on);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= ln_act_bwd(act, y[j], an + j);
            float* dst = &As[tid / (BM / 4) + (1024 / BM) * q][4 * (tid % (BM / 4))];
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const bool on = kt + tid / (BN / 4) + (1024 / BN) * q < k1;
            ln_f4 v = ln_fix4(rb[q], bj, d.K, on);
            if (on && bj <= d.K && bj + 3 >= d.K) {            // the bias-gradient column: [X | 1]
                const int c1 = d.K - bj;
                v[0] = c1 == 0 ? 1.f : v[0]; v[1] = c1 == 1 ? 1.f : v[1]; v[2] = c1 == 2 ? 1.f : v[2]; v[3] = c1 == 3 ? 1.f : v[3];
            }
            float* dst = &Bs[tid / (BN / 4) + (1024 / BN) * q][4 * (tid % (BN / 4))];
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        MST_LDS_BARRIER();
        if (kt + LN_KT < k1) issue(kt + LN_KT);
        mst_mfma_ktile_2x2<LN_KT>(As, Bs, wm * 64, wn * 64, l31, kh, acc);
        MST_LDS_BARRIER();
    }
    // this split's slab, parameter layout: weight (N x K) then bias (N)
    float* slab = b.p[SP_TMP] + d.slab_off + (int64_t)split * d.slab_stride;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int j = j0 + wn * 64 + 32 * c + l31;
        if (j >= NB) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (n >= d.N) continue;
                if (j < d.K) slab[(int64_t)n * d.K + j] = acc[a][c][r];
                else slab[(int64_t)d.N * d.K + n] = acc[a][c][r];
            }
    }
}

// ---- N <= 16 outputs (the rhythm encoder's 280 -> 16 Linear over every (position, fraction) row of the raw note tensor: 183 MB of
// input per 64-clip launch and 1.5 GFLOP — an HBM stream, not a GEMM).  A 64-column tile would idle 3/4 of its MFMA columns and
// a 2 x 2-blocked wave has nothing to reuse, so these two kernels use v_mfma_f32_16x16x4_f32 (the 16 outputs ARE the tile width),
// read every input row exactly once per pass (forward: 256-row tiles; weight gradient: ONE tile spans all K + 1 <= 320 columns,
// so a k-split reads its rows once) and keep 2-3 workgroups x 32-40 KB of loads in flight per CU.
typedef float ln_f32x4 __attribute__((ext_vector_type(4)));
#define LN16_JMAX 320

__global__ __launch_bounds__(256, 2) void lin16_fwd_kernel(const LinDesc d, Bases b) {
    constexpr int BM = 256, PA = BM + 1, QA = BM / 32;
    __shared__ float As[LN_KT][PA];
    __shared__ float Bs[LN_KT][17];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wv = MST_UNIFORM(tid >> 6);
    const int64_t Mtot = (int64_t)d.clips * d.rows;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const ln_gp xa = (ln_gp)(b.p[d.x_space] + d.x_off);
    const ln_gp w = (ln_gp)(b.p[SP_PAR] + d.w_off);
    unsigned aoff[QA];
    bool aon[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int64_t m = m0 + (tid >> 3) + 32 * q;
        aon[q] = m < Mtot;
        const unsigned mm = aon[q] ? (unsigned)m : 0u;
        const unsigned clip = mm / (unsigned)d.rows, r = mm - clip * (unsigned)d.rows;
        aoff[q] = (unsigned)((int64_t)clip * d.x_cs + (int64_t)r * d.x_ld);
    }
    const int bn = tid >> 3;                                   // lanes 0..127: weight row n = bn, k group tid & 7
    const bool bon = tid < 128 && bn < d.N;
    ln_f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = ln_f32x4{0.f, 0.f, 0.f, 0.f};
    ln_f4 ra[QA], rb;
    auto issue = [&](const int kt) {
        const int ka = kt + 4 * (tid & 7);
#pragma unroll
        for (int q = 0; q < QA; ++q) ra[q] = ln_ld4(xa + aoff[q], ka, d.K, aon[q]);
        rb = ln_ld4(w + (unsigned)((bon ? bn : 0) * d.K), ka, d.K, bon);
    };
    issue(0);
    for (int kt = 0; kt < d.K; kt += LN_KT) {
        const int ka = kt + 4 * (tid & 7);
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const ln_f4 v = ln_fix4(ra[q], ka, d.K, aon[q]);
#pragma unroll
            for (int j = 0; j < 4; ++j) As[4 * (tid & 7) + j][(tid >> 3) + 32 * q] = v[j];
        }
        if (tid < 128) {
            const ln_f4 v = ln_fix4(rb, ka, d.K, bon);
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[4 * (tid & 7) + j][bn] = v[j];
        }
        MST_LDS_BARRIER();
        if (kt + LN_KT < d.K) issue(kt + LN_KT);
#pragma unroll
        for (int ks = 0; ks < LN_KT / 4; ++ks) {
            const int k = 4 * ks + kq;
            const float bv = Bs[k][l15];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[k][wv * 64 + 16 * q + l15], bv, acc[q], 0, 0, 0);
        }
        MST_LDS_BARRIER();
    }
    // epilogue: lane -> column l15, rows 16 q + 4 kq + j of the wave's 64: one division, then (clip, row) advance by the row deltas
    if (l15 >= d.N) return;
    const float bias = b.p[SP_PAR][d.b_off + l15];
    float* out = b.p[SP_WS] + d.y_off;
    const unsigned mb = (unsigned)m0 + (unsigned)(wv * 64 + 4 * kq);
    unsigned eclip = mb / (unsigned)d.rows, erow = mb - eclip * (unsigned)d.rows, eprev = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned delta = 16 * q + j;
            erow += delta - eprev; eprev = delta;
            { const bool wr = erow >= (unsigned)d.rows; erow -= wr ? (unsigned)d.rows : 0u; eclip += wr ? 1u : 0u; }      // rows >= 32: one wrap at most
            if ((int64_t)mb + delta >= Mtot) continue;
            out[(int64_t)eclip * d.y_cs + (int64_t)erow * d.y_ld + l15] = ln_act_fwd(d.act, acc[q][j] + bias, l15);
        }
}

// dW | db for N <= 16: A(n, k = m) = dYa[m][n], B(k = m, j) = [X | 1][m][j]; the tile is N x (K + 1 <= 320): wave w owns column blocks
// 5 w .. 5 w + 4 (16 columns each).  Row m of a k-tile belongs to the eight lanes tid / 8 = m: one (clip, row) pair per lane.
__global__ __launch_bounds__(256, 2) void lin16_dw_kernel(const LinDesc d, Bases b) {
    constexpr int PB = LN16_JMAX + 16;
    __shared__ float As[LN_KT][17];
    __shared__ __attribute__((aligned(16))) float Bs[LN_KT][PB];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wv = MST_UNIFORM(tid >> 6);
    const int split = blockIdx.x;
    const int64_t Mtot = (int64_t)d.clips * d.rows;
    const int64_t k0 = (int64_t)split * d.rows_per_split, k1 = min(Mtot, k0 + d.rows_per_split);
    const ln_gp gy = (ln_gp)(b.p[SP_GRAD] + d.y_off), yy = (ln_gp)(b.p[SP_WS] + d.y_off), xx = (ln_gp)(b.p[d.x_space] + d.x_off);
    const int mrow = tid >> 3, t7 = tid & 7;                   // row of the k-tile, column-group phase
    const int an = 4 * (t7 & 3);                               // lanes with t7 < 4 also carry the row's four groups of dY / Y
    const int act = d.act;
    ln_f32x4 acc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = ln_f32x4{0.f, 0.f, 0.f, 0.f};
    ln_f4 ra, ry, rb[10];
    unsigned rc, rr;                                           // (clip, row inside the clip) of this lane's row of the next k-tile
    {
        const unsigned f = (unsigned)k0 + (unsigned)mrow;
        rc = f / (unsigned)d.rows; rr = f - rc * (unsigned)d.rows;
    }
    auto issue = [&](const int64_t kt) {
        const bool on = kt + mrow < k1;
        const unsigned clip = on ? rc : 0u, r = on ? rr : 0u;
        const unsigned oy = (unsigned)((int64_t)clip * d.y_cs + (int64_t)r * d.y_ld);
        const unsigned ox = (unsigned)((int64_t)clip * d.x_cs + (int64_t)r * d.x_ld);
        if (t7 < 4) { ra = ln_ld4(gy + oy, an, d.N, on); ry = ln_ld4(yy + oy, an, d.N, on); }
#pragma unroll
        for (int q = 0; q < 10; ++q) rb[q] = ln_ld4(xx + ox, 4 * (t7 + 8 * q), d.K, on);
        rr += LN_KT; if (rr >= (unsigned)d.rows) { rr -= (unsigned)d.rows; ++rc; }
    };
    if (k0 < k1) issue(k0);
    for (int64_t kt = k0; kt < k1; kt += LN_KT) {
        const bool on = kt + mrow < k1;
        if (t7 < 4) {
            ln_f4 v = ln_fix4(ra, an, d.N, on);
            const ln_f4 y = ln_fix4(ry, an, d.N, on);
#pragma unroll
            for (int j = 0; j < 4; ++j) As[mrow][an + j] = v[j] * ln_act_bwd(act, y[j], an + j);
        }
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const int bj = 4 * (t7 + 8 * q);
            ln_f4 v = ln_fix4(rb[q], bj, d.K, on);
            if (on && bj <= d.K && bj + 3 >= d.K) {            // the bias-gradient column: [X | 1]
                const int c1 = d.K - bj;
                v[0] = c1 == 0 ? 1.f : v[0]; v[1] = c1 == 1 ? 1.f : v[1]; v[2] = c1 == 2 ? 1.f : v[2]; v[3] = c1 == 3 ? 1.f : v[3];
            }
            float* dst = &Bs[mrow][bj];
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        MST_LDS_BARRIER();
        if (kt + LN_KT < k1) issue(kt + LN_KT);
#pragma unroll
        for (int ks = 0; ks < LN_KT / 4; ++ks) {
            const int k = 4 * ks + kq;
            const float av = As[k][l15];
#pragma unroll
            for (int q = 0; q < 5; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bs[k][(wv * 5 + q) * 16 + l15], acc[q], 0, 0, 0);
        }
        MST_LDS_BARRIER();
    }
    // this split's slab, parameter layout: weight (N x K) then bias (N); lane -> column j, outputs n = 4 kq + i
    float* slab = b.p[SP_TMP] + d.slab_off + (int64_t)split * d.slab_stride;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int j = (wv * 5 + q) * 16 + l15;
        if (j > d.K) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 4 * kq + i;
            if (n >= d.N) continue;
            if (j < d.K) slab[(int64_t)n * d.K + j] = acc[q][i];
            else slab[(int64_t)d.N * d.K + n] = acc[q][i];
        }
    }
}

// tile shape by the output's column count: <= 64 columns -> four waves stacked over the rows (256 x 64), else 2 x 2 (128 x 128)
int lin_rows_blocks(const LinDesc& d, int mode) {
    const int nc = mode == 0 ? d.N : d.K;
    const int64_t mtot = (int64_t)d.clips * d.rows;
    if (nc <= 64) return (int)((mtot + 255) / 256);
    return (int)((mtot + 127) / 128) * ((nc + 127) / 128);
}
static bool lin_is16(const LinDesc& d) { return d.N <= 16 && d.K + 1 <= LN16_JMAX; }
int launch_lin_fwd(const LinDesc& d, Bases b, hipStream_t s) {
    if (lin_is16(d)) {
        hipLaunchKernelGGL(lin16_fwd_kernel, dim3((unsigned)(((int64_t)d.clips * d.rows + 255) / 256)), dim3(256), 0, s, d, b);
        return (int)hipGetLastError();
    }
    const dim3 grid(lin_rows_blocks(d, 0));
    if (d.N <= 64) hipLaunchKernelGGL((lin_rows_kernel<4, 1, 0>), grid, dim3(256), 0, s, d, b, 0);
    else hipLaunchKernelGGL((lin_rows_kernel<2, 2, 0>), grid, dim3(256), 0, s, d, b, 0);
    return (int)hipGetLastError();
}
int launch_lin_dx(const LinDesc& d, Bases b, int first, hipStream_t s) {
    const dim3 grid(lin_rows_blocks(d, 1));
    if (d.K <= 64) hipLaunchKernelGGL((lin_rows_kernel<4, 1, 1>), grid, dim3(256), 0, s, d, b, first);
    else hipLaunchKernelGGL((lin_rows_kernel<2, 2, 1>), grid, dim3(256), 0, s, d, b, first);
    return (int)hipGetLastError();
}
// weight-gradient tile: <= 64 outputs -> 64 x 256 (four waves side by side), else 128 x 128
int lin_dw_tiles(const LinDesc& d) {
    if (lin_is16(d)) return 1;
    if (d.N <= 64) return (d.K + 1 + 255) / 256;
    return ((d.N + 127) / 128) * ((d.K + 1 + 127) / 128);
}
int launch_lin_dw(const LinDesc& d, Bases b, hipStream_t s) {
    if (lin_is16(d)) {
        hipLaunchKernelGGL(lin16_dw_kernel, dim3(d.splits), dim3(256), 0, s, d, b);
        return (int)hipGetLastError();
    }
    const dim3 grid(lin_dw_tiles(d) * d.splits);
    if (d.N <= 64) hipLaunchKernelGGL((lin_dw_kernel<1, 4>), grid, dim3(256), 0, s, d, b);
    else hipLaunchKernelGGL((lin_dw_kernel<2, 2>), grid, dim3(256), 0, s, d, b);
    return (int)hipGetLastError();
}
