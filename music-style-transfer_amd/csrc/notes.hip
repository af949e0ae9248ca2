// Note-level fused stages for gfx950.
//
// MelodyEncoder tail (style/model.py:270-296) and PitchedStyleApplier tail (:644-675) share one
// shape: an "octave (+) scale-degree" outer sum  h[o*7+d] = leaky(oct[o] + deg[d])  per note,
// concatenated with a small per-note vector and pushed through a tiny Linear.  The reference
// materialises the (positions x 56 x 50) concat in memory (358 MB at the training cap); here
// every note is one lane, the octave/degree rows of the position sit in LDS, the tiny weights
// are LDS-broadcast, and nothing but the final 8 (or 5) outputs per note touches HBM.
//
// Backward kernels recompute the cheap per-note activations instead of saving them, stage the
// per-note gradient vectors of one (position, fraction) in LDS, and then run "role" threads:
// one lane per weight-gradient element (kept in a register across the workgroup's whole
// grid-stride loop, one slab row per workgroup at the end) and one lane per octave / degree
// gradient element (7- or 8-term LDS sums).  No float atomics; summation order is fixed.
#include "mst_common.h"

__device__ __forceinline__ float lrelu(float z) { return z > 0.f ? z : z * LEAKY; }
__device__ __forceinline__ float dlrelu(float y) { return y > 0.f ? 1.f : LEAKY; }

// ============================================================================ MelodyEncoder
template <int W, int CW>
__global__ __launch_bounds__(256) void me_notes_fwd_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    constexpr int KL = W + CW;
    __shared__ float wc_s[CW * NPF], bc_s[CW], wl_s[W * KL], bl_s[W];
    __shared__ float oct_s[NOCT * W], deg_s[NDEG * W];
    const int tid = threadIdx.x;
    const float* par = b.p[SP_PAR];
    for (int i = tid; i < CW * NPF; i += 256) wc_s[i] = par[d.wc_off + i];
    for (int i = tid; i < CW; i += 256) bc_s[i] = par[d.bc_off + i];
    for (int i = tid; i < W * KL; i += 256) wl_s[i] = par[d.wl_off + i];
    for (int i = tid; i < W; i += 256) bl_s[i] = par[d.bl_off + i];
    const float* x = b.p[d.x_space] + d.x_off;
    float* ws = b.p[SP_WS];
    const int P = d.C * d.Q;
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < NOCT * W; i += 256) oct_s[i] = ws[d.oct_off + (int64_t)p * NOCT * W + i];
        for (int i = tid; i < NDEG * W; i += 256) deg_s[i] = ws[d.deg_off + (int64_t)p * NDEG * W + i];
        __syncthreads();
        for (int item = tid; item < NF * NPN; item += 256) {
            const int n = item % NPN, o = n / NDEG, dg = n - o * NDEG;
            const float* xi = x + ((int64_t)p * NF * NPN + item) * NPF;
            float x5[NPF];
#pragma unroll
            for (int i = 0; i < NPF; ++i) x5[i] = xi[i];
            float cat[KL];
#pragma unroll
            for (int j = 0; j < W; ++j) cat[j] = lrelu(oct_s[o * W + j] + deg_s[dg * W + j]);
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                float z = bc_s[k];
#pragma unroll
                for (int i = 0; i < NPF; ++i) z = fmaf(wc_s[k * NPF + i], x5[i], z);
                cat[W + k] = lrelu(z);
            }
            float* out = ws + d.out_off + ((int64_t)p * NF * NPN + item) * W;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                float z = bl_s[j];
#pragma unroll
                for (int i = 0; i < KL; ++i) z = fmaf(wl_s[j * KL + i], cat[i], z);
                out[j] = lrelu(z);
            }
        }
    }
}

template <int W, int CW>
__global__ __launch_bounds__(256) void me_notes_bwd_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    constexpr int KL = W + CW;
    constexpr int R_WC = 0, R_BC = CW * NPF, R_WL = R_BC + CW, R_BL = R_WL + W * KL, NW = R_BL + W;
    static_assert(NW <= 256 && (NOCT + NDEG) * W <= 256, "role count exceeds the workgroup");
    __shared__ float wc_s[CW * NPF], bc_s[CW], wl_s[W * KL];
    __shared__ float oct_s[NOCT * W], deg_s[NDEG * W];
    // per-note vectors staged transposed ([feature][note], 16-byte aligned rows): lane-contiguous writes in the note phase,
    // two 16-byte LDS reads per four FMAs in the role sums (see psa_notes_bwd_kernel)
    constexpr int NTP = NPN + 4;
    static_assert(NPN % 4 == 0, "notes per group");
    __shared__ __attribute__((aligned(16))) float gm_t[W][NTP], gc_t[CW][NTP], cat_t[KL][NTP], x_t[NPF][NTP];
    __shared__ float god_t[W][NTP];
    const int tid = threadIdx.x;
    const float* par = b.p[SP_PAR];
    for (int i = tid; i < CW * NPF; i += 256) wc_s[i] = par[d.wc_off + i];
    for (int i = tid; i < CW; i += 256) bc_s[i] = par[d.bc_off + i];
    for (int i = tid; i < W * KL; i += 256) wl_s[i] = par[d.wl_off + i];
    const float* x = b.p[d.x_space] + d.x_off;
    float* ws = b.p[SP_WS];
    const int P = d.C * d.Q;
    float wacc = 0.f;                       // this lane's weight-gradient element
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < NOCT * W; i += 256) oct_s[i] = ws[d.oct_off + (int64_t)p * NOCT * W + i];
        for (int i = tid; i < NDEG * W; i += 256) deg_s[i] = ws[d.deg_off + (int64_t)p * NDEG * W + i];
        float odacc = 0.f;                  // this lane's octave / degree gradient element
        for (int f = 0; f < NF; ++f) {
            __syncthreads();
            // note phase on all four waves: wave w handles every note (lane = note) for the cat elements
            // i in [w*IQ, w*IQ + IQ); the note's 8 output gradients are re-read by each wave (L1 hits)
            if ((tid & 63) < NPN) {
                constexpr int IQ = (KL + 3) / 4;
                const int wv = tid >> 6, n = tid & 63, o = n / NDEG, dg = n - o * NDEG;
                const int64_t pos = (int64_t)p * NF * NPN + f * NPN + n;
                float x5[NPF];
#pragma unroll
                for (int i = 0; i < NPF; ++i) x5[i] = x[pos * NPF + i];
                float gm[W];
#pragma unroll
                for (int j = 0; j < W; ++j) gm[j] = b.p[SP_GRAD][d.g_out_off + pos * W + j] * dlrelu(ws[d.out_off + pos * W + j]);
                if (wv == 0) {
#pragma unroll
                    for (int i = 0; i < NPF; ++i) x_t[i][n] = x5[i];
#pragma unroll
                    for (int j = 0; j < W; ++j) gm_t[j][n] = gm[j];
                }
#pragma unroll
                for (int ii = 0; ii < IQ; ++ii) {
                    const int i = wv * IQ + ii;
                    if (i < KL) {
                        float c;
                        if (i < W) {
                            c = lrelu(oct_s[o * W + i] + deg_s[dg * W + i]);
                        } else {
                            float z = bc_s[i - W];
#pragma unroll
                            for (int q = 0; q < NPF; ++q) z = fmaf(wc_s[(i - W) * NPF + q], x5[q], z);
                            c = lrelu(z);
                        }
                        float g = 0.f;
#pragma unroll
                        for (int j = 0; j < W; ++j) g = fmaf(gm[j], wl_s[j * KL + i], g);
                        g *= dlrelu(c);
                        cat_t[i][n] = c;
                        if (i < W) god_t[i][n] = g; else gc_t[i - W][n] = g;
                    }
                }
            }
            __syncthreads();
            if (tid < NW) {
                // a role lane sums pa[n] (* pb[n]) over the 56 notes, four notes per pair of 16-byte reads
                const float4* pa;
                const float4* pb = nullptr;
                if (tid >= R_BL) {
                    pa = reinterpret_cast<const float4*>(gm_t[tid - R_BL]);
                } else if (tid >= R_WL) {
                    pa = reinterpret_cast<const float4*>(gm_t[(tid - R_WL) / KL]);
                    pb = reinterpret_cast<const float4*>(cat_t[(tid - R_WL) % KL]);
                } else if (tid >= R_BC) {
                    pa = reinterpret_cast<const float4*>(gc_t[tid - R_BC]);
                } else {
                    pa = reinterpret_cast<const float4*>(gc_t[tid / NPF]);
                    pb = reinterpret_cast<const float4*>(x_t[tid % NPF]);
                }
                float a = 0.f;
                if (pb) {
#pragma unroll
                    for (int m4 = 0; m4 < NPN / 4; ++m4) {
                        const float4 u = pa[m4], v = pb[m4];
                        a = fmaf(u.x, v.x, a); a = fmaf(u.y, v.y, a); a = fmaf(u.z, v.z, a); a = fmaf(u.w, v.w, a);
                    }
                } else {
#pragma unroll
                    for (int m4 = 0; m4 < NPN / 4; ++m4) { const float4 u = pa[m4]; a += u.x; a += u.y; a += u.z; a += u.w; }
                }
                wacc += a;
            }
            if (tid < NOCT * W) {
                const int o = tid / W, j = tid % W;
                float a = 0.f;
#pragma unroll
                for (int dg = 0; dg < NDEG; ++dg) a += god_t[j][o * NDEG + dg];
                odacc += a;
            } else if (tid < (NOCT + NDEG) * W) {
                const int dg = (tid - NOCT * W) / W, j = tid % W;
                float a = 0.f;
#pragma unroll
                for (int o = 0; o < NOCT; ++o) a += god_t[j][o * NDEG + dg];
                odacc += a;
            }
        }
        // sole writer of these rows: store, not read-modify-write
        if (tid < NOCT * W) b.p[SP_GRAD][d.g_oct_off + (int64_t)p * NOCT * W + tid] = odacc;
        else if (tid < (NOCT + NDEG) * W) b.p[SP_GRAD][d.g_deg_off + (int64_t)p * NDEG * W + (tid - NOCT * W)] = odacc;
    }
    if (tid < NW) b.p[SP_TMP][d.slab_off + (int64_t)blockIdx.x * d.slab_stride + tid] = wacc;
}

// ============================================================================ PitchedStyleApplier
#define PSA_HW 30
template <int ML>
__global__ __launch_bounds__(64) void psa_notes_fwd_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    constexpr int KL = PSA_HW + ML;
    __shared__ float w_s[NPF * KL], b_s[NPF];
    __shared__ float lo_s[NOCT * PSA_HW], ld_s[NDEG * PSA_HW];
    const int tid = threadIdx.x;
    const float* par = b.p[SP_PAR];
    for (int i = tid; i < NPF * KL; i += 64) w_s[i] = par[d.wl_off + i];
    if (tid < NPF) b_s[tid] = par[d.bl_off + tid];
    float* ws = b.p[SP_WS];
    const int QF = d.Q * NF;
    const int n = tid, o = n / NDEG, dg = n - o * NDEG;
    // the octave | degree rows of the NEXT (qf, c) sit in 8 registers per lane while the current one is processed
    constexpr int NLO = NOCT * PSA_HW, NOD = (NOCT + NDEG) * PSA_HW, NPRE = (NOD + 63) / 64;
    float od[NPRE];
    auto fetch_od = [&](int64_t row) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int i = tid + 64 * q;
            od[q] = i < NLO ? ws[d.oct_off + row * NLO + i] : (i < NOD ? ws[d.deg_off + row * (NOD - NLO) + (i - NLO)] : 0.f);
        }
    };
    if ((int)blockIdx.x < QF) fetch_od((int64_t)blockIdx.x);
    for (int qf = blockIdx.x; qf < QF; qf += gridDim.x) {
        __syncthreads();
        float zm[NPF];
#pragma unroll
        for (int i = 0; i < NPF; ++i) zm[i] = 0.f;
        if (n < NPN) {
            const float* ml = ws + d.ml_off + ((int64_t)qf * NPN + n) * ML;
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                float z = b_s[i];
#pragma unroll
                for (int k = 0; k < ML; ++k) z = fmaf(w_s[i * KL + PSA_HW + k], ml[k], z);
                zm[i] = z;
            }
        }
        for (int c = 0; c < d.C; ++c) {
            const int64_t row = (int64_t)c * QF + qf;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NPRE; ++q) {
                const int i = tid + 64 * q;
                if (i < NLO) lo_s[i] = od[q]; else if (i < NOD) ld_s[i - NLO] = od[q];
            }
            __syncthreads();
            if (c + 1 < d.C) fetch_od((int64_t)(c + 1) * QF + qf);
            else if (qf + (int)gridDim.x < QF) fetch_od((int64_t)(qf + gridDim.x));
            if (n < NPN) {
                float z[NPF];
#pragma unroll
                for (int i = 0; i < NPF; ++i) z[i] = zm[i];
#pragma unroll
                for (int j = 0; j < PSA_HW; ++j) {
                    const float h = lrelu(lo_s[o * PSA_HW + j] + ld_s[dg * PSA_HW + j]);
#pragma unroll
                    for (int i = 0; i < NPF; ++i) z[i] = fmaf(w_s[i * KL + j], h, z[i]);
                }
                float* out = ws + d.out_off + (row * NPN + n) * NPF;
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    const float s = 1.f / (1.f + expf(-z[i]));
                    out[i] = i == 0 ? 6.f * s : s;
                }
            }
        }
    }
}

template <int ML>
__global__ __launch_bounds__(256) void psa_notes_bwd_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    constexpr int KL = PSA_HW + ML;
    constexpr int NW = NPF * KL + NPF;                 // linear.weight (5 x KL) then linear.bias (5)
    constexpr int NLO = NOCT * PSA_HW, NLD = NDEG * PSA_HW;
    constexpr int NTP = NPN + 4;                       // note-major rows, 16-byte aligned, 8 lanes of a b128 read span all banks
    static_assert(NW <= 256 && NPN % 4 == 0, "role count exceeds the workgroup");
    __shared__ float w_s[NPF * KL];
    __shared__ float lo_s[NLO], ld_s[NLD];
    // per-note vectors are staged TRANSPOSED ([feature][note]): the note phase writes lane-contiguous, and a role lane
    // reads four notes of its two operands with two 16-byte LDS reads per four FMAs (it was two 4-byte reads per FMA:
    // PMC showed one LDS instruction per 1.8 VALU instructions)
    __shared__ __attribute__((aligned(16))) float dz_t[NPF][NTP];
    __shared__ __attribute__((aligned(16))) float h_t[PSA_HW][NTP];
    __shared__ __attribute__((aligned(16))) float ml_t[ML][NTP];
    __shared__ float dh_t[PSA_HW][NTP];
    const int tid = threadIdx.x;
    const float* par = b.p[SP_PAR];
    for (int i = tid; i < NPF * KL; i += 256) w_s[i] = par[d.wl_off + i];
    float* ws = b.p[SP_WS];
    const int QF = d.Q * NF;
    // note phase on all four waves: lane = note, wave w handles hidden features j in [8w, 8w + 8) and the melody-linear
    // gradients k in [w*KQ, w*KQ + KQ); the note's 5 output gradients are re-read by each wave (L1 hits)
    constexpr int JQ = (PSA_HW + 3) / 4, KQ = (ML + 3) / 4;
    const int wv = tid >> 6, n = tid & 63, o = n / NDEG, dg = n - o * NDEG;
    float wacc = 0.f;
    // the octave | degree rows of the NEXT (qf, c) are fetched into two registers per lane while the current one is
    // processed, so their global-load latency is not paid at every barrier
    constexpr int NOD = NLO + NLD;                      // 450 <= 2 * 256
    float od0 = 0.f, od1 = 0.f;
    auto fetch_od = [&](int64_t row) {
        od0 = tid < NLO ? ws[d.oct_off + row * NLO + tid] : ws[d.deg_off + row * NLD + (tid - NLO)];
        if (tid + 256 < NOD) od1 = ws[d.deg_off + row * NLD + (tid + 256 - NLO)];
    };
    if ((int)blockIdx.x < QF) fetch_od((int64_t)blockIdx.x);
    // this lane's role: element (ri, rj) of linear.weight, or bias element ri
    const int ri = tid < NPF * KL ? tid / KL : tid - NPF * KL;
    const int rj = tid < NPF * KL ? tid % KL : -1;
    for (int qf = blockIdx.x; qf < QF; qf += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < NPN * ML; i += 256) ml_t[i % ML][i / ML] = ws[d.ml_off + (int64_t)qf * NPN * ML + i];
        float gml[KQ];
#pragma unroll
        for (int k = 0; k < KQ; ++k) gml[k] = 0.f;
        for (int c = 0; c < d.C; ++c) {
            const int64_t row = (int64_t)c * QF + qf;
            __syncthreads();
            if (tid < NLO) lo_s[tid] = od0; else ld_s[tid - NLO] = od0;
            if (tid + 256 < NOD) ld_s[tid + 256 - NLO] = od1;
            __syncthreads();
            if (c + 1 < d.C) fetch_od((int64_t)(c + 1) * QF + qf);
            else if (qf + (int)gridDim.x < QF) fetch_od((int64_t)(qf + gridDim.x));
            if (n < NPN) {
                const int64_t pos = row * NPN + n;
                float dz[NPF];
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    const float y = ws[d.out_off + pos * NPF + i];
                    const float dy = b.p[SP_GRAD][d.g_out_off + pos * NPF + i];
                    dz[i] = dy * (i == 0 ? y * (1.f - y * (1.f / 6.f)) : y * (1.f - y));
                    if (wv == 0) dz_t[i][n] = dz[i];
                }
#pragma unroll
                for (int jj = 0; jj < JQ; ++jj) {
                    const int j = wv * JQ + jj;
                    if (j < PSA_HW) {
                        const float h = lrelu(lo_s[o * PSA_HW + j] + ld_s[dg * PSA_HW + j]);
                        float g = 0.f;
#pragma unroll
                        for (int i = 0; i < NPF; ++i) g = fmaf(dz[i], w_s[i * KL + j], g);
                        h_t[j][n] = h;
                        dh_t[j][n] = g * dlrelu(h);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) {
                    const int k = wv * KQ + kk;
                    if (k < ML) {
                        float g = 0.f;
#pragma unroll
                        for (int i = 0; i < NPF; ++i) g = fmaf(dz[i], w_s[i * KL + PSA_HW + k], g);
                        gml[kk] += g;
                    }
                }
            }
            __syncthreads();
            if (tid < NW) {
                const float4* pa = reinterpret_cast<const float4*>(dz_t[ri]);
                float a = 0.f;
                if (rj < 0) {
#pragma unroll
                    for (int m4 = 0; m4 < NPN / 4; ++m4) { const float4 x = pa[m4]; a += x.x; a += x.y; a += x.z; a += x.w; }
                } else {
                    const float4* pb = reinterpret_cast<const float4*>(rj < PSA_HW ? h_t[rj] : ml_t[rj - PSA_HW]);
#pragma unroll
                    for (int m4 = 0; m4 < NPN / 4; ++m4) {
                        const float4 x = pa[m4], y = pb[m4];
                        a = fmaf(x.x, y.x, a); a = fmaf(x.y, y.y, a); a = fmaf(x.z, y.z, a); a = fmaf(x.w, y.w, a);
                    }
                }
                wacc += a;
            }
            for (int r = tid; r < NLO + NLD; r += 256) {
                float a = 0.f;
                if (r < NLO) {
                    const int oo = r / PSA_HW, j = r % PSA_HW;
#pragma unroll
                    for (int q = 0; q < NDEG; ++q) a += dh_t[j][oo * NDEG + q];
                    b.p[SP_GRAD][d.g_oct_off + row * NLO + r] = a;      // sole writer of this row: store, not read-modify-write
                } else {
                    const int q = (r - NLO) / PSA_HW, j = (r - NLO) % PSA_HW;
#pragma unroll
                    for (int oo = 0; oo < NOCT; ++oo) a += dh_t[j][oo * NDEG + q];
                    b.p[SP_GRAD][d.g_deg_off + row * NLD + (r - NLO)] = a;
                }
            }
        }
        if (n < NPN) {
            float* g = b.p[SP_GRAD] + d.g_ml_off + ((int64_t)qf * NPN + n) * ML;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int k = wv * KQ + kk;
                if (k < ML) g[k] = gml[kk];                            // sole writer (all channels summed above)
            }
        }
    }
    if (tid < NW) b.p[SP_TMP][d.slab_off + (int64_t)blockIdx.x * d.slab_stride + tid] = wacc;
}

// ============================================================================ dispatch
bool notes_widths_supported(int W, int CW, int ML) {
    bool me = (W == 8 && CW == 7) || (W == 4 && CW == 5);
    bool psa = ML == 20 || ML == 14;
    return me && psa;
}

#define ME_DISPATCH(KERN, GRID, BLOCK)                                                              \
    if (h.W == 8 && h.CW == 7) hipLaunchKernelGGL((KERN<8, 7>), GRID, BLOCK, 0, s, dev, b);           \
    else if (h.W == 4 && h.CW == 5) hipLaunchKernelGGL((KERN<4, 5>), GRID, BLOCK, 0, s, dev, b);      \
    else return MST_ERR_UNSUPPORTED;
#define PSA_DISPATCH(KERN, GRID, BLOCK)                                                             \
    if (h.ML == 20) hipLaunchKernelGGL((KERN<20>), GRID, BLOCK, 0, s, dev, b);                        \
    else if (h.ML == 14) hipLaunchKernelGGL((KERN<14>), GRID, BLOCK, 0, s, dev, b);                   \
    else return MST_ERR_UNSUPPORTED;

int launch_me_notes_fwd(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    int P = h.C * h.Q;
    ME_DISPATCH(me_notes_fwd_kernel, dim3(P < 2048 ? P : 2048, count), dim3(256));
    return (int)hipGetLastError();
}
int launch_me_notes_bwd(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    ME_DISPATCH(me_notes_bwd_kernel, dim3(h.nblk, count), dim3(256));
    return (int)hipGetLastError();
}
int launch_psa_notes_fwd(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    int QF = h.Q * NF;
    PSA_DISPATCH(psa_notes_fwd_kernel, dim3(QF < 4096 ? QF : 4096, count), dim3(64));
    return (int)hipGetLastError();
}
int launch_psa_notes_bwd(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    PSA_DISPATCH(psa_notes_bwd_kernel, dim3(h.nblk, count), dim3(256));
    return (int)hipGetLastError();
}

// ============================================================================ row-wise tiny Linear
// One lane per row: the K_in inputs and N_out outputs of a row live in registers, the weights are
// LDS-broadcast, rows are read / written with the widest aligned vector the widths allow.  Backward
// does input gradient and weight gradient in one pass: every lane stages its row's dZ and x in LDS
// (transposed, [feature][row]), then "role" lanes (one per weight / bias element) sum the 256 staged rows in row order and keep
// their element in a register across the grid-stride loop — one slab row per workgroup, no shuffles,
// no float atomics.
__device__ __forceinline__ float rl_act(int act, float z, int col) {
    if (act == ACT_LEAKY) return lrelu(z);
    if (act == ACT_SIGOUT) { const float s = 1.f / (1.f + expf(-z)); return col == 0 ? 6.f * s : s; }
    return z;
}
__device__ __forceinline__ float rl_dact(int act, float y, int col) {
    if (act == ACT_LEAKY) return dlrelu(y);
    if (act == ACT_SIGOUT) return col == 0 ? y * (1.f - y * (1.f / 6.f)) : y * (1.f - y);
    return 1.f;
}
template <int N>
__device__ __forceinline__ void rl_load(const float* p, float* v) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { const float4 t = reinterpret_cast<const float4*>(p)[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { const float2 t = reinterpret_cast<const float2*>(p)[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
}
template <int N>
__device__ __forceinline__ void rl_store(float* p, const float* v) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) reinterpret_cast<float2*>(p)[i] = make_float2(v[2 * i], v[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = v[i];
    }
}

template <int KIN, int NOUT>
__global__ __launch_bounds__(256) void rowlin_fwd_kernel(const RowLinDesc* __restrict__ dp, Bases b) {
    const RowLinDesc d = dp[blockIdx.y];
    __shared__ float w_s[NOUT * KIN], b_s[NOUT];
    const int tid = threadIdx.x;
    for (int i = tid; i < NOUT * KIN; i += 256) w_s[i] = b.p[SP_PAR][d.w_off + i];
    if (tid < NOUT) b_s[tid] = b.p[SP_PAR][d.b_off + tid];
    __syncthreads();
    const float* x0 = b.p[SP_WS] + d.x_off;
    float* y0 = b.p[SP_WS] + d.y_off;
    for (int row = blockIdx.x * 256 + tid; row < d.rows; row += gridDim.x * 256) {
        float x[KIN], y[NOUT];
        rl_load<KIN>(x0 + (int64_t)row * KIN, x);
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            float z = b_s[n];
#pragma unroll
            for (int k = 0; k < KIN; ++k) z = fmaf(w_s[n * KIN + k], x[k], z);
            y[n] = rl_act(d.act, z, n);
        }
        rl_store<NOUT>(y0 + (int64_t)row * NOUT, y);
    }
}

template <int KIN, int NOUT>
__global__ __launch_bounds__(256) void rowlin_bwd_kernel(const RowLinDesc* __restrict__ dp, Bases b) {
    const RowLinDesc d = dp[blockIdx.y];
    constexpr int NACC = NOUT * KIN + NOUT;            // weight (NOUT x KIN) then bias (NOUT), the parameters' own order
    constexpr int RW = NOUT + KIN;                     // staged per row: dZ[NOUT] | x[KIN]
    constexpr int RP = 256 + 4;                        // transposed staging [feature][row]: lane-contiguous writes, 16-byte role reads
    static_assert(NACC <= 256, "role count exceeds the workgroup");
    __shared__ float w_s[NOUT * KIN];
    __shared__ __attribute__((aligned(16))) float st[RW][RP];
    const int tid = threadIdx.x;
    for (int i = tid; i < NOUT * KIN; i += 256) w_s[i] = b.p[SP_PAR][d.w_off + i];
    const float* x0 = b.p[SP_WS] + d.x_off;
    const float* y0 = b.p[SP_WS] + d.y_off;
    const float* gy0 = b.p[SP_GRAD] + d.y_off;
    float* gx0 = b.p[SP_GRAD] + d.x_off;
    // role lanes: lane t < NOUT*KIN owns dW[n][k], the next NOUT lanes own db[n]; each sums the 256 staged rows of a
    // pass in row order and keeps its element in a register across the grid-stride loop (one slab row per workgroup)
    const int rn = tid < NOUT * KIN ? tid / KIN : tid - NOUT * KIN;
    const int rk = tid < NOUT * KIN ? NOUT + tid % KIN : -1;
    float wacc = 0.f;
    for (int base = blockIdx.x * 256; base < d.rows; base += gridDim.x * 256) {
        const int row = base + tid;
        __syncthreads();                                // previous pass's role reads are done (and w_s is loaded)
        if (row < d.rows) {
            float x[KIN], y[NOUT], g[NOUT];
            rl_load<KIN>(x0 + (int64_t)row * KIN, x);
            rl_load<NOUT>(y0 + (int64_t)row * NOUT, y);
            rl_load<NOUT>(gy0 + (int64_t)row * NOUT, g);
#pragma unroll
            for (int n = 0; n < NOUT; ++n) { g[n] *= rl_dact(d.act, y[n], n); st[n][tid] = g[n]; }
#pragma unroll
            for (int k = 0; k < KIN; ++k) st[NOUT + k][tid] = x[k];
            if (d.xgrad) {
                float dx[KIN];
                rl_load<KIN>(gx0 + (int64_t)row * KIN, dx);
#pragma unroll
                for (int k = 0; k < KIN; ++k) {
                    float a = 0.f;
#pragma unroll
                    for (int n = 0; n < NOUT; ++n) a = fmaf(g[n], w_s[n * KIN + k], a);
                    dx[k] += a;
                }
                rl_store<KIN>(gx0 + (int64_t)row * KIN, dx);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NOUT + KIN; ++j) st[j][tid] = 0.f;
        }
        __syncthreads();
        if (tid < NACC) {
            float a = 0.f;
            const float4* pa = reinterpret_cast<const float4*>(st[rn]);
            if (rk >= 0) {
                const float4* pb = reinterpret_cast<const float4*>(st[rk]);
#pragma unroll 4
                for (int r4 = 0; r4 < 64; ++r4) {
                    const float4 u = pa[r4], v = pb[r4];
                    a = fmaf(u.x, v.x, a); a = fmaf(u.y, v.y, a); a = fmaf(u.z, v.z, a); a = fmaf(u.w, v.w, a);
                }
            } else {
#pragma unroll 4
                for (int r4 = 0; r4 < 64; ++r4) { const float4 u = pa[r4]; a += u.x; a += u.y; a += u.z; a += u.w; }
            }
            wacc += a;
        }
    }
    if (tid < NACC) b.p[SP_TMP][d.slab_off + (int64_t)blockIdx.x * d.slab_stride + tid] = wacc;
}

bool rowlin_supported(int kin, int nout) { return (kin == 8 && nout == 20) || (kin == 4 && nout == 14) || (kin == 8 && nout == 2); }

#define ROWLIN_DISPATCH(KERN, GRID)                                                                         \
    if (h.kin == 8 && h.nout == 20) hipLaunchKernelGGL((KERN<8, 20>), GRID, dim3(256), 0, s, dev, b);         \
    else if (h.kin == 4 && h.nout == 14) hipLaunchKernelGGL((KERN<4, 14>), GRID, dim3(256), 0, s, dev, b);    \
    else if (h.kin == 8 && h.nout == 2) hipLaunchKernelGGL((KERN<8, 2>), GRID, dim3(256), 0, s, dev, b);      \
    else return MST_ERR_UNSUPPORTED;

int launch_rowlin_fwd(const RowLinDesc* dev, const RowLinDesc& h, int count, Bases b, hipStream_t s) {
    int nb = (h.rows + 255) / 256;
    if (nb > 1024) nb = 1024;
    ROWLIN_DISPATCH(rowlin_fwd_kernel, dim3(nb, count));
    return (int)hipGetLastError();
}
int launch_rowlin_bwd(const RowLinDesc* dev, const RowLinDesc& h, int count, Bases b, hipStream_t s) {
    ROWLIN_DISPATCH(rowlin_bwd_kernel, dim3(h.nblk, count));
    return (int)hipGetLastError();
}
