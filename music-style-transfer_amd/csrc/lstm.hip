// LSTM recurrence for gfx950 (nn.LSTM as wrapped at style/utils/pytorch.py:19-25; torch gate
// order i,f,g,o; zero initial state; one layer).  The input projection x W_ih^T + b_ih is a
// plain GEMM (gemm.hip); only the strictly sequential part lives here.
//
// One workgroup per sequence.  H <= 64: one lane per gate row, its W_hh row (forward) or W_hh
// column slice (backward) lives in VGPRs for the whole sequence, h_{t-1} / dz_t are broadcast
// from LDS, and the streamed per-step operands (zx row, saved gates) are prefetched one step
// ahead so a step costs two barriers and no exposed memory latency.  H > 64 (StyleEncoder,
// H = 192, 590 KB of W_hh — more than one CU's registers + LDS): one lane per gate row over W_hh^T (
// transposed once per forward so the per-step L2 reads are lane-contiguous); every in-loop barrier
// is LDS-only (MST_LDS_BARRIER), so streamed stores / prefetches never stall a step.
#include "mst_common.h"

#define LSTM_ZS 8192        // floats of LDS holding staged per-step operands in the register flavours
__device__ __forceinline__ float sigm(float x) { return MST_FAST_RCP(1.f + MST_FAST_EXP(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.f * sigm(2.f * x) - 1.f; }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// REG (H <= 64) launches at most 256 lanes, so it may use the whole register file of one wave per SIMD
template <bool REG>
__global__ __launch_bounds__(REG ? 256 : 1024) void lstm_fwd_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];        // by value: no descriptor re-reads after the per-step barriers
    const int bi = blockIdx.x;
    if (bi >= d.B) return;
    const int H = d.H, G = 4 * d.H, tid = threadIdx.x;
    __shared__ float h_s[256];
    __shared__ float z_s[1024];
    // REG: the input-projection rows of the next LSTM_ZS / 4H steps wait in LDS, fetched together.  (Prefetching one step
    // ahead left every step waiting out most of an L2 / HBM round trip: 1.0-1.4 us per step for ~0.4 us of work.)
    __shared__ float zx_s[REG ? LSTM_ZS : 1];
    const int zchunk = REG ? LSTM_ZS / G : 1;
    const float* whh = b.p[SP_PAR] + d.whh_off;
    const float* zx = b.p[SP_WS] + d.zx_off;
    float* ws = b.p[SP_WS];
    float* tmp = b.p[SP_TMP];
    float w[REG ? 64 : 1];
    if (REG && tid < G) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {       // unconditional (clamped) loads: all 64 in flight, no branches
            const float v = whh[(int64_t)tid * H + min(k, H - 1)];
            w[k] = k < H ? v : 0.f;
        }
    }
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, zq[4] = {0.f, 0.f, 0.f, 0.f};
    const int s0 = d.reverse ? d.S - 1 : 0;
    if (tid < H) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bias[q] = b.p[SP_PAR][d.bhh_off + q * H + tid];
            if (!REG) zq[q] = zx[((int64_t)bi * d.S + s0) * G + q * H + tid];
        }
    }
    if (REG) {               // landed before the loop: the step loop then holds no load wait that would also drain its stores
#pragma unroll
        for (int q = 0; q < 4; ++q) MST_PIN(bias[q]);
    }
    if (tid < 256) h_s[tid] = 0.f;
    float c = 0.f;
    for (int step = 0; step < d.S; ++step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int64_t row = (int64_t)bi * d.S + s;
        float zn[4] = {0.f, 0.f, 0.f, 0.f};
        if (REG) {
            if (step % zchunk == 0) {
                if (step) MST_LDS_BARRIER();                 // the gate lanes are done with the previous chunk
                const int cnt = min(zchunk, d.S - step);
                if (tid < G) {
                    // global-address-space pointer: a generic one may alias LDS, and every load then waits for the previous
                    // iteration's LDS store
                    const MST_GLOBAL_AS float* zg = (const MST_GLOBAL_AS float*)zx + (int64_t)bi * d.S * G + tid;
                    const int sstep = d.reverse ? -G : G;
                    zg += (int64_t)(d.reverse ? d.S - 1 - step : step) * G;
#pragma unroll 8
                    for (int i = 0; i < cnt; ++i) zx_s[i * G + tid] = zg[(int64_t)i * sstep];      // independent loads, all in flight
                }
            }
        } else if (tid < H && step + 1 < d.S) {             // next step's zx row: in flight under this step's matvec
            const int sn = d.reverse ? s - 1 : s + 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) zn[q] = zx[((int64_t)bi * d.S + sn) * G + q * H + tid];
        }
        MST_LDS_BARRIER();
        if (REG) {
            if (tid < G) {
                // all of h_{t-1} into registers first (sixteen 16-byte broadcast reads, ONE wait), then the FMAs: left to itself the
                // compiler feeds the chains two reads at a time with a wait in front of every second FMA — eight LDS round trips per
                // step on the sequential path.  (One wave per SIMD: registers are not what this kernel is short of.)
                float hv[64];
#pragma unroll
                for (int k = 0; k < 64; ++k) hv[k] = h_s[k];
                __builtin_amdgcn_sched_barrier(0);
                float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;       // 4 independent FMA chains
#pragma unroll
                for (int k = 0; k < 64; k += 4) {
                    z0 = fmaf(w[k], hv[k], z0); z1 = fmaf(w[k + 1], hv[k + 1], z1);
                    z2 = fmaf(w[k + 2], hv[k + 2], z2); z3 = fmaf(w[k + 3], hv[k + 3], z3);
                }
                z_s[tid] = (z0 + z1) + (z2 + z3);
            }
        } else if (tid < G) {
            // thread per gate row over the transposed copy: lane-contiguous (coalesced) reads, 4 chains
            // summed as four quarters of k, each four interleaved chains, then (q0 + q1) + (q2 + q3): the association of the
            // multi-workgroup flavour below, so a clip's activations are bit-identical in one-clip and batched plans.  The
            // four quarters advance together: sixteen independent loads per trip
            const float* wt = tmp + d.whht_off + tid;
            float qs[4] = {0.f, 0.f, 0.f, 0.f};
            if (H % 16 == 0) {
                const int KQ = H / 4;
                float a[4][4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) { a[qq][0] = 0.f; a[qq][1] = 0.f; a[qq][2] = 0.f; a[qq][3] = 0.f; }
                for (int k = 0; k < KQ; k += 4) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[qq][i] = fmaf(wt[(int64_t)(qq * KQ + k + i) * G], h_s[qq * KQ + k + i], a[qq][i]);
                    }
                }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) qs[qq] = (a[qq][0] + a[qq][1]) + (a[qq][2] + a[qq][3]);
            } else {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int k = 0;
                for (; k + 4 <= H; k += 4) {
                    a0 = fmaf(wt[(int64_t)k * G], h_s[k], a0);
                    a1 = fmaf(wt[(int64_t)(k + 1) * G], h_s[k + 1], a1);
                    a2 = fmaf(wt[(int64_t)(k + 2) * G], h_s[k + 2], a2);
                    a3 = fmaf(wt[(int64_t)(k + 3) * G], h_s[k + 3], a3);
                }
                for (; k < H; ++k) a0 = fmaf(wt[(int64_t)k * G], h_s[k], a0);
                qs[0] = (a0 + a1) + (a2 + a3);
            }
            z_s[tid] = (qs[0] + qs[1]) + (qs[2] + qs[3]);
        }
        MST_LDS_BARRIER();
        if (tid < H) {
            if (REG) {
#pragma unroll
                for (int q = 0; q < 4; ++q) zq[q] = zx_s[(step % zchunk) * G + q * H + tid];
            }
            const float ig = sigm(z_s[tid] + zq[0] + bias[0]), fg = sigm(z_s[H + tid] + zq[1] + bias[1]);
            const float gg = tanh_fast(z_s[2 * H + tid] + zq[2] + bias[2]), og = sigm(z_s[3 * H + tid] + zq[3] + bias[3]);
            tmp[d.hprev_off + row * H + tid] = h_s[tid];
            c = fg * c + ig * gg;
            const float tc = tanh_fast(c);
            const float h = og * tc;
            tmp[d.tc_off + row * H + tid] = tc;
            float* g = tmp + d.gates_off + row * G;
            g[tid] = ig; g[H + tid] = fg; g[2 * H + tid] = gg; g[3 * H + tid] = og;
            tmp[d.c_off + row * H + tid] = c;
            ws[d.out_off + row * d.out_ld + tid] = h;
            h_s[tid] = h;
            if (!REG) {
#pragma unroll
                for (int q = 0; q < 4; ++q) zq[q] = zn[q];
            }
        }
    }
}

template <bool REG>
__global__ __launch_bounds__(REG ? 256 : 1024) void lstm_bwd_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];        // by value: no descriptor re-reads after the per-step barriers
    const int bi = blockIdx.x;
    if (bi >= d.B) return;
    const int H = d.H, G = 4 * d.H, tid = threadIdx.x;
    __shared__ float dz_s[1024];
    // partial sums of dh_{t-1} per hidden unit, consumed by the next step: four (one per gate) in the register flavour,
    // sixteen 48-row chunks in the L2 flavour (the association of the multi-workgroup kernel: bit-identical results)
    __shared__ float red_s[REG ? 1024 : 16 * 256];
    const bool chunked = !REG && (d.H % 16 == 0) && d.H <= 256;
    const float* whh = b.p[SP_PAR] + d.whh_off;
    const float* tmp = b.p[SP_TMP];
    float* gr = b.p[SP_GRAD];
    const int kk = tid % H, part = tid / H;                 // lane (part, kk) sums W_hh[part*H + jj, kk] dz[part*H + jj]
    float w[REG ? 64 : 1];
    if (REG && tid < G) {
#pragma unroll
        for (int jj = 0; jj < 64; ++jj) {
            const float v = whh[((int64_t)part * H + min(jj, H - 1)) * H + kk];
            w[jj] = jj < H ? v : 0.f;
        }
    }
    if (REG) red_s[tid] = 0.f;
    else for (int i = tid; i < 16 * 256; i += blockDim.x) red_s[i] = 0.f;
    float dc_next = 0.f;
    // streamed operands of a step (saved gates, cell states, incoming gradient), prefetched one step ahead
    float sv[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const MST_GLOBAL_AS float* tmpg = (const MST_GLOBAL_AS float*)tmp;      // not generic: cannot alias the LDS staging
    const MST_GLOBAL_AS float* grg = (const MST_GLOBAL_AS float*)gr;
#define LSTM_LOAD(STEP, DST)                                                                            \
    {                                                                                                   \
        const int s_ = d.reverse ? d.S - 1 - (STEP) : (STEP);                                           \
        const int sp_ = d.reverse ? s_ + 1 : s_ - 1;                                                    \
        const int64_t row_ = (int64_t)bi * d.S + s_;                                                    \
        const MST_GLOBAL_AS float* g_ = tmpg + d.gates_off + row_ * G;                                  \
        DST[0] = g_[tid]; DST[1] = g_[H + tid]; DST[2] = g_[2 * H + tid]; DST[3] = g_[3 * H + tid];     \
        DST[4] = tmpg[d.tc_off + row_ * H + tid];                                                       \
        DST[5] = (STEP) > 0 ? tmpg[d.c_off + ((int64_t)bi * d.S + sp_) * H + tid] : 0.f;                \
        DST[6] = grg[d.gout_off + row_ * d.out_ld + tid];                                               \
    }
    // REG: the streamed operands of the next LSTM_ZS / 7H steps are fetched together and parked in LDS, each lane in slots
    // of its own (no barrier): one step of prefetch left a step waiting out most of a memory round trip
    __shared__ float sv_s[REG ? LSTM_ZS : 1];
    const int schunk = REG ? LSTM_ZS / (7 * H) : 1;
    if (!REG && tid < H) LSTM_LOAD(d.S - 1, sv)
    __syncthreads();
    for (int step = d.S - 1; step >= 0; --step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int64_t row = (int64_t)bi * d.S + s;
        float nx[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tid < H) {
            if (REG) {
                const int pos = d.S - 1 - step, slot = pos % schunk;
                if (slot == 0) {
                    const int cnt = min(schunk, step + 1);
#pragma unroll 4
                    for (int i = 0; i < cnt; ++i) {          // independent loads, all in flight
                        float t[7];
                        LSTM_LOAD(step - i, t)
#pragma unroll
                        for (int q = 0; q < 7; ++q) sv_s[(i * 7 + q) * H + tid] = t[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < 7; ++q) sv[q] = sv_s[(slot * 7 + q) * H + tid];
            } else if (step > 0) LSTM_LOAD(step - 1, nx)
            const float ig = sv[0], fg = sv[1], gg = sv[2], og = sv[3], tc = sv[4], cprev = sv[5];
            float dhr;
            if (chunked) {
                dhr = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) dhr += red_s[q * H + tid];
            } else dhr = (red_s[tid] + red_s[H + tid]) + (red_s[2 * H + tid] + red_s[3 * H + tid]);
            const float dh = sv[6] + dhr;
            const float dc = dc_next + dh * og * (1.f - tc * tc);
            const float dzi = dc * gg * ig * (1.f - ig);
            const float dzf = dc * cprev * fg * (1.f - fg);
            const float dzg = dc * ig * (1.f - gg * gg);
            const float dzo = dh * tc * og * (1.f - og);
            dc_next = dc * fg;
            dz_s[tid] = dzi; dz_s[H + tid] = dzf; dz_s[2 * H + tid] = dzg; dz_s[3 * H + tid] = dzo;
            float* gz = gr + d.gzx_off + row * G;
            gz[tid] = dzi; gz[H + tid] = dzf; gz[2 * H + tid] = dzg; gz[3 * H + tid] = dzo;
        }
        MST_LDS_BARRIER();
        float acc = 0.f;
        float cacc[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < G) {   // dh_{t-1}[k] = sum_j W_hh[j,k] dz[j], four partial sums per k (summed by the next step)
            if (REG) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const int base = part * H;
                float dv[64];                                 // the gate's dz slice in registers first, then the FMAs (see lstm_fwd_kernel)
#pragma unroll
                for (int jj = 0; jj < 64; ++jj) dv[jj] = dz_s[min(base + jj, G - 1)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < 64; jj += 4) {
                    a0 = fmaf(w[jj], dv[jj], a0);
                    a1 = fmaf(w[jj + 1], dv[jj + 1], a1);
                    a2 = fmaf(w[jj + 2], dv[jj + 2], a2);
                    a3 = fmaf(w[jj + 3], dv[jj + 3], a3);
                }
                acc = (a0 + a1) + (a2 + a3);
            } else {
                const float* wc = whh + (int64_t)part * H * H + kk;      // coalesced along kk
                const float* dzp = dz_s + part * H;
                if (chunked) {                                            // four 48-row chunks advance together: 16 loads per trip
                    const int CH = H / 4;
                    float a[4][4];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) { a[ci][0] = 0.f; a[ci][1] = 0.f; a[ci][2] = 0.f; a[ci][3] = 0.f; }
                    for (int jj = 0; jj < CH; jj += 4) {
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) a[ci][i] = fmaf(wc[(int64_t)(ci * CH + jj + i) * H], dzp[ci * CH + jj + i], a[ci][i]);
                        }
                    }
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) cacc[ci] = (a[ci][0] + a[ci][1]) + (a[ci][2] + a[ci][3]);
                } else {
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    int jj = 0;
                    for (; jj + 4 <= H; jj += 4) {
                        a0 = fmaf(wc[(int64_t)jj * H], dzp[jj], a0);
                        a1 = fmaf(wc[(int64_t)(jj + 1) * H], dzp[jj + 1], a1);
                        a2 = fmaf(wc[(int64_t)(jj + 2) * H], dzp[jj + 2], a2);
                        a3 = fmaf(wc[(int64_t)(jj + 3) * H], dzp[jj + 3], a3);
                    }
                    for (; jj < H; ++jj) a0 = fmaf(wc[(int64_t)jj * H], dzp[jj], a0);
                    acc = (a0 + a1) + (a2 + a3);
                }
            }
        }
        if (!REG && tid < H) {
#pragma unroll
            for (int q = 0; q < 7; ++q) sv[q] = nx[q];     // the prefetch landed under the matvec above
        }
        if (tid < G) {                                    // phase A's reads of red_s ended before the barrier above
            if (chunked) {
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) red_s[(4 * part + ci) * H + kk] = cacc[ci];
            } else red_s[tid] = acc;
        }
        MST_LDS_BARRIER();
    }
#undef LSTM_LOAD
}

// ---- grouped flavour ---------------------------------------------------------------------------
// Batched plans run thousands of short sequences through the same small LSTM (the beats LSTM: 64 sequences of 4 steps per
// direction and clip, H = 64): with one sequence per workgroup, fetching the 64 KB of W_hh into registers cost more than the
// four steps that used it.  Here a workgroup carries LSTM_NS sequences of the same LSTM: the weights are fetched once, every
// step is a [4H x H] x [H x NS] product out of registers and LDS, and the gate math of the NS sequences spreads over
// NS * H lanes.  Per sequence the arithmetic (and its order) is that of lstm_fwd_kernel<true> / lstm_bwd_kernel<true>.
#define LSTM_NS 4
__global__ __launch_bounds__(256) void lstm_fwd_group_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];
    const int b0 = blockIdx.x * LSTM_NS;
    if (b0 >= d.B) return;
    const int H = d.H, G = 4 * d.H, tid = threadIdx.x;
    __shared__ float h_s[LSTM_NS][64];
    __shared__ float z_s[LSTM_NS][256];
    const float* whh = b.p[SP_PAR] + d.whh_off;
    const float* zx = b.p[SP_WS] + d.zx_off;
    float* ws = b.p[SP_WS];
    float* tmp = b.p[SP_TMP];
    float w[64];
    if (tid < G) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float v = whh[(int64_t)tid * H + min(k, H - 1)];
            w[k] = k < H ? v : 0.f;
        }
    }
    // gate lane (sq, hh): hidden unit hh of the workgroup's sequence sq
    const int sq = tid / H, hh = tid - sq * H, bi = b0 + sq;
    const bool gate = sq < LSTM_NS && bi < d.B;
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, zq[4] = {0.f, 0.f, 0.f, 0.f};
    const int s0 = d.reverse ? d.S - 1 : 0;
    if (gate) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bias[q] = b.p[SP_PAR][d.bhh_off + q * H + hh];
            zq[q] = zx[((int64_t)bi * d.S + s0) * G + q * H + hh];
        }
    }
    (&h_s[0][0])[tid] = 0.f;
    float c = 0.f;
    for (int step = 0; step < d.S; ++step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int64_t row = (int64_t)bi * d.S + s;
        float zn[4] = {0.f, 0.f, 0.f, 0.f};
        if (gate && step + 1 < d.S) {
            const int sn = d.reverse ? s - 1 : s + 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) zn[q] = zx[((int64_t)bi * d.S + sn) * G + q * H + hh];
        }
        MST_LDS_BARRIER();
        if (tid < G) {
            float z[LSTM_NS][4];
#pragma unroll
            for (int j = 0; j < LSTM_NS; ++j) { z[j][0] = 0.f; z[j][1] = 0.f; z[j][2] = 0.f; z[j][3] = 0.f; }
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
#pragma unroll
                for (int j = 0; j < LSTM_NS; ++j) {
                    z[j][0] = fmaf(w[k], h_s[j][k], z[j][0]); z[j][1] = fmaf(w[k + 1], h_s[j][k + 1], z[j][1]);
                    z[j][2] = fmaf(w[k + 2], h_s[j][k + 2], z[j][2]); z[j][3] = fmaf(w[k + 3], h_s[j][k + 3], z[j][3]);
                }
            }
#pragma unroll
            for (int j = 0; j < LSTM_NS; ++j) z_s[j][tid] = (z[j][0] + z[j][1]) + (z[j][2] + z[j][3]);
        }
        MST_LDS_BARRIER();
        if (gate) {
            const float* zr = z_s[sq];
            const float ig = sigm(zr[hh] + zq[0] + bias[0]), fg = sigm(zr[H + hh] + zq[1] + bias[1]);
            const float gg = tanh_fast(zr[2 * H + hh] + zq[2] + bias[2]), og = sigm(zr[3 * H + hh] + zq[3] + bias[3]);
            tmp[d.hprev_off + row * H + hh] = h_s[sq][hh];
            c = fg * c + ig * gg;
            const float tc = tanh_fast(c);
            const float h = og * tc;
            tmp[d.tc_off + row * H + hh] = tc;
            float* g = tmp + d.gates_off + row * G;
            g[hh] = ig; g[H + hh] = fg; g[2 * H + hh] = gg; g[3 * H + hh] = og;
            tmp[d.c_off + row * H + hh] = c;
            ws[d.out_off + row * d.out_ld + hh] = h;
            h_s[sq][hh] = h;
#pragma unroll
            for (int q = 0; q < 4; ++q) zq[q] = zn[q];
        }
    }
}

__global__ __launch_bounds__(256, 2) void lstm_bwd_group_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];
    const int b0 = blockIdx.x * LSTM_NS;
    if (b0 >= d.B) return;
    const int H = d.H, G = 4 * d.H, tid = threadIdx.x;
    __shared__ float dz_s[LSTM_NS][256 + 64];               // zero tail: lane (part, kk) reads rows part*H .. part*H + 63 unclamped
    __shared__ float red_s[LSTM_NS][256];
    const float* whh = b.p[SP_PAR] + d.whh_off;
    const float* tmp = b.p[SP_TMP];
    float* gr = b.p[SP_GRAD];
    const int kk = tid % H, part = tid / H;                 // matvec lane (part, kk), as in lstm_bwd_kernel<true>
    float w[64];
    if (tid < G) {
        // 32-bit lane offsets off the uniform base (4 H^2 floats): half the address registers of 64 flat pointers
        const MST_GLOBAL_AS float* wb = (const MST_GLOBAL_AS float*)whh;
        const unsigned o0 = (unsigned)(part * H) * (unsigned)H + (unsigned)kk;
#pragma unroll
        for (int jj = 0; jj < 64; ++jj) {
            const float v = wb[o0 + (unsigned)(min(jj, H - 1) * H)];
            w[jj] = jj < H ? v : 0.f;
        }
    }
    const int sq = part, hh = kk, bi = b0 + sq;             // gate lane (sq, hh)
    const bool gate = sq < LSTM_NS && bi < d.B;
#pragma unroll
    for (int j = 0; j < LSTM_NS; ++j) { red_s[j][tid] = 0.f; dz_s[j][tid] = 0.f; if (tid < 64) dz_s[j][256 + tid] = 0.f; }
    float dc_next = 0.f;
    float sv[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load = [&](const int step, float* dst) {
        const int s_ = d.reverse ? d.S - 1 - step : step;
        const int sp_ = d.reverse ? s_ + 1 : s_ - 1;
        const int64_t row_ = (int64_t)bi * d.S + s_;
        const float* g_ = tmp + d.gates_off + row_ * G;
        dst[0] = g_[hh]; dst[1] = g_[H + hh]; dst[2] = g_[2 * H + hh]; dst[3] = g_[3 * H + hh];
        dst[4] = tmp[d.tc_off + row_ * H + hh];
        dst[5] = step > 0 ? tmp[d.c_off + ((int64_t)bi * d.S + sp_) * H + hh] : 0.f;
        dst[6] = gr[d.gout_off + row_ * d.out_ld + hh];
    };
    if (gate) load(d.S - 1, sv);
    __syncthreads();
    for (int step = d.S - 1; step >= 0; --step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int64_t row = (int64_t)bi * d.S + s;
        float nx[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (gate) {
            if (step > 0) load(step - 1, nx);
            const float ig = sv[0], fg = sv[1], gg = sv[2], og = sv[3], tc = sv[4], cprev = sv[5];
            const float* rr = red_s[sq];
            const float dhr = (rr[hh] + rr[H + hh]) + (rr[2 * H + hh] + rr[3 * H + hh]);
            const float dh = sv[6] + dhr;
            const float dc = dc_next + dh * og * (1.f - tc * tc);
            const float dzi = dc * gg * ig * (1.f - ig);
            const float dzf = dc * cprev * fg * (1.f - fg);
            const float dzg = dc * ig * (1.f - gg * gg);
            const float dzo = dh * tc * og * (1.f - og);
            dc_next = dc * fg;
            float* dzr = dz_s[sq];
            dzr[hh] = dzi; dzr[H + hh] = dzf; dzr[2 * H + hh] = dzg; dzr[3 * H + hh] = dzo;
            float* gz = gr + d.gzx_off + row * G;
            gz[hh] = dzi; gz[H + hh] = dzf; gz[2 * H + hh] = dzg; gz[3 * H + hh] = dzo;
        }
        MST_LDS_BARRIER();
        if (tid < G) {          // the gate lanes' reads of red_s ended before the barrier above
            // one sequence at a time (a real loop): fully unrolled over the sequences, the compiler fetched all 256 LDS
            // operands of the step before the first FMA and spilled
            const float* dzp = &dz_s[0][0] + part * H;
#pragma unroll 1
            for (int j = 0; j < LSTM_NS; ++j) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int jj = 0; jj < 64; jj += 4) {
                    a0 = fmaf(w[jj], dzp[jj], a0); a1 = fmaf(w[jj + 1], dzp[jj + 1], a1);
                    a2 = fmaf(w[jj + 2], dzp[jj + 2], a2); a3 = fmaf(w[jj + 3], dzp[jj + 3], a3);
                }
                red_s[j][tid] = (a0 + a1) + (a2 + a3);
                dzp += 256 + 64;
            }
        }
        if (gate) {
#pragma unroll
            for (int q = 0; q < 7; ++q) sv[q] = nx[q];
        }
        MST_LDS_BARRIER();
    }
}

// ---- multi-workgroup flavour -------------------------------------------------------------------
// One clip, H = 192 (StyleEncoder.bars_lstm), batch 1: a single workgroup streamed all 590 KB of W_hh from L2 every step
// (one CU's L2 bandwidth: 5.4 us per step, 2 x 82 us per iteration — a fifth of the whole iteration).  Here LSTM_NB = 12
// workgroups own 16 hidden units each: their 64 gate rows (forward) / 16 columns (backward) of W_hh live in registers for
// the whole sequence (48 floats per lane), and per step they exchange only h_t (192 floats) or dz_t (768 floats) through
// tagged 8-byte granules {epoch, value}: one relaxed agent-scope (sc1) store publishes a value, consumers poll the granule
// until its tag is this step's epoch — the data is its own flag, no fence, no separate counter
// (cdna_hip_programming.md Guideline 16, R2).  Buffers alternate with the step's parity: a workgroup can publish step
// t + 1 only after it has consumed every workgroup's step t, so the slot it overwrites (t - 1) is no longer needed by
// anyone.  Tags are cleared before every forward (lstm_transpose_kernel): epochs restart at 1 in every launch.
// The 12 x clips workgroups of a launch must all be resident at once (the plan sizes the launch from the occupancy the
// runtime reports for these kernels, lstm_multi_blocks_per_cu(), and keeps one workgroup slot per CU free); dispatch order
// is undefined and nothing here depends on which workgroup starts first.
// FAILURE IS LOUD AND BOUNDED: a lane that does not see its tag within LSTM_WAIT_TICKS of wall clock (0.2 s; a healthy wait is
// microseconds) ORs MST_DEV_LSTM_TIMEOUT into the plan's device status word (LstmDesc.status_off), returns NaN and makes every
// later wait of that lane return NaN at once; every other waiting lane of the launch looks at the status word every 256 polls
// and gives up as soon as it is set, so the whole launch drains within a millisecond of the first timeout instead of
// S x 0.2 s.  The status word is sticky until the host reads and clears it (mst_plan_status); while it is set, every wait of
// a later launch gives up after 256 polls — nothing computed after a failure looks healthy.
#ifdef HIPSIM
#define LSTM_WAIT_TICKS 8            /* the interpreter's clock counts calls: 8 x 256 polls */
#else
#define LSTM_WAIT_TICKS 20000000     /* wall_clock64() runs at 100 MHz: 0.2 s */
#endif
typedef unsigned long long lstm_gran_t;
__device__ __forceinline__ void gran_store(lstm_gran_t* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((lstm_gran_t)tag << 32) | (lstm_gran_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float gran_wait(const lstm_gran_t* g, unsigned tag, int* status, bool& dead) {
    if (dead) return __builtin_nanf("");
    long long t0 = 0;
    for (unsigned spin = 0;; ++spin) {
        const lstm_gran_t x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == tag) return __uint_as_float((unsigned)x);
        __builtin_amdgcn_s_sleep(1);
        if ((spin & 255u) == 255u) {                     // off the fast path: a healthy wait ends within a few polls
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;      // somebody gave up
            const long long now = wall_clock64();
            if (spin == 255u) t0 = now;
            else if (now - t0 > LSTM_WAIT_TICKS) { __hip_atomic_fetch_or(status, MST_DEV_LSTM_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    dead = true;
    return __builtin_nanf("");
}
// LDS of a kernel whose workgroups are co-resident: the interpreter keeps one copy per block (its `__shared__` is a static)
#ifdef HIPSIM
#define MST_COOP_LDS(name, n) static float name##_all[hipsim::COOP_MAXB][n]; float* name = name##_all[hipsim::coop_block()]
#define MST_COOP_LDS2(name, n, m) static float name##_all[hipsim::COOP_MAXB][n][m]; float (*name)[m] = name##_all[hipsim::coop_block()]
#define MST_LAUNCH_CORESIDENT(k, g, b, s, ...) hipsim::launch_coop((k), (g), (b), (s), __VA_ARGS__)
#else
#define MST_COOP_LDS(name, n) __shared__ __attribute__((aligned(16))) float name[n]
#define MST_COOP_LDS2(name, n, m) __shared__ float name[n][m]
#define MST_LAUNCH_CORESIDENT(k, g, b, s, ...) hipLaunchKernelGGL((k), (g), (b), 0, (s), __VA_ARGS__)
#endif

__global__ __launch_bounds__(256) void lstm_multi_fwd_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];
    constexpr int H = LSTM_MH, G = 4 * H, HU = H / LSTM_NB, KQ = H / 4;       // 16 units per workgroup, 48 k per lane
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int row_l = tid >> 2, kq = tid & 3;                                  // gate row of this workgroup / quarter of k
    const int gate = row_l / HU, u = row_l - gate * HU;
    const int j = gate * H + wg * HU + u;                                      // row of W_hh
    MST_COOP_LDS(h_s, H);
    MST_COOP_LDS(z_s, 4 * HU);
    const float* whh = b.p[SP_PAR] + d.whh_off;
    const float* zx = b.p[SP_WS] + d.zx_off;
    float* ws = b.p[SP_WS];
    float* tmp = b.p[SP_TMP];
    lstm_gran_t* xch = reinterpret_cast<lstm_gran_t*>(tmp + d.xch_off);        // [2][H]
    int* status = reinterpret_cast<int*>(b.p[SP_WS] + d.status_off);
    bool dead = false;
    // multi == 2 (mst_plan_options.lstm_flavour = 2, tests only): workgroup 0 publishes its first step under a wrong epoch, so
    // that every consumer of it runs into the timeout path
    const unsigned fault = (d.multi == 2 && wg == 0) ? 0x10000u : 0u;
    float w[KQ];
#pragma unroll
    for (int i = 0; i < KQ / 4; ++i) {
        const float4 t4 = *reinterpret_cast<const float4*>(whh + (int64_t)j * H + kq * KQ + 4 * i);
        w[4 * i] = t4.x; w[4 * i + 1] = t4.y; w[4 * i + 2] = t4.z; w[4 * i + 3] = t4.w;
    }
    // unit lanes: tid < HU owns hidden unit k = wg * HU + tid
    const int k = wg * HU + (tid < HU ? tid : 0);
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, zq[4] = {0.f, 0.f, 0.f, 0.f};
    const int s0 = d.reverse ? d.S - 1 : 0;
    if (tid < HU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { bias[q] = b.p[SP_PAR][d.bhh_off + q * H + k]; zq[q] = zx[(int64_t)s0 * G + q * H + k]; }
    }
    float c = 0.f, hprev = 0.f;
    for (int step = 0; step < d.S; ++step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        float zn[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < HU && step + 1 < d.S) {                    // next step's zx row: in flight under this step
            const int sn = d.reverse ? s - 1 : s + 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) zn[q] = zx[(int64_t)sn * G + q * H + k];
        }
        // h_{t-1} of every workgroup (tag = step: published with epoch (step - 1) + 1)
        if (tid < H) h_s[tid] = step == 0 ? 0.f : gran_wait(xch + ((step - 1) & 1) * H + tid, (unsigned)step, status, dead);
        __syncthreads();
        if (tid < HU) hprev = h_s[k];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < KQ; i += 4) {
            const float4 h4 = *reinterpret_cast<const float4*>(h_s + kq * KQ + i);
            a0 = fmaf(w[i], h4.x, a0); a1 = fmaf(w[i + 1], h4.y, a1); a2 = fmaf(w[i + 2], h4.z, a2); a3 = fmaf(w[i + 3], h4.w, a3);
        }
        float zsum = (a0 + a1) + (a2 + a3);
        zsum += __shfl_xor(zsum, 1);
        zsum += __shfl_xor(zsum, 2);
        if (kq == 0) z_s[row_l] = zsum;
        __syncthreads();
        if (tid < HU) {
            const int64_t row = s;
            const float ig = sigm(z_s[tid] + zq[0] + bias[0]), fg = sigm(z_s[HU + tid] + zq[1] + bias[1]);
            const float gg = tanh_fast(z_s[2 * HU + tid] + zq[2] + bias[2]), og = sigm(z_s[3 * HU + tid] + zq[3] + bias[3]);
            c = fg * c + ig * gg;
            const float tc = tanh_fast(c);
            const float h = og * tc;
            gran_store(xch + (step & 1) * H + k, (unsigned)(step + 1) ^ (step == 0 ? fault : 0u), h);      // first: the other workgroups wait for it
            tmp[d.hprev_off + row * H + k] = hprev;
            tmp[d.tc_off + row * H + k] = tc;
            float* g = tmp + d.gates_off + row * G;
            g[k] = ig; g[H + k] = fg; g[2 * H + k] = gg; g[3 * H + k] = og;
            tmp[d.c_off + row * H + k] = c;
            ws[d.out_off + row * d.out_ld + k] = h;
#pragma unroll
            for (int q = 0; q < 4; ++q) zq[q] = zn[q];
        }
    }
}

__global__ __launch_bounds__(256) void lstm_multi_bwd_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];
    constexpr int H = LSTM_MH, G = 4 * H, HU = H / LSTM_NB, JQ = 16, JW = G / JQ;   // 16 j-chunks of 48 gate rows
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int kk = tid & (HU - 1), jq = tid / HU;             // lane (jq, kk) sums W_hh[jq*48 + i][wg*16 + kk] dz[jq*48 + i]
    MST_COOP_LDS(dz_s, G);
    MST_COOP_LDS2(part_s, JQ, HU + 1);
    const float* whh = b.p[SP_PAR] + d.whh_off;
    const float* tmp = b.p[SP_TMP];
    float* gr = b.p[SP_GRAD];
    int* status = reinterpret_cast<int*>(b.p[SP_WS] + d.status_off);
    bool dead = false;
    lstm_gran_t* xch = reinterpret_cast<lstm_gran_t*>(b.p[SP_TMP] + d.xch_off) + 2 * H;     // [2][G], behind the forward's
    float w[JW];
#pragma unroll
    for (int i = 0; i < JW; ++i) w[i] = whh[(int64_t)(jq * JW + i) * H + wg * HU + kk];
    const int k = wg * HU + (tid < HU ? tid : 0);
    float dc_next = 0.f, dh_rec = 0.f;
    float sv[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define LSTM_MLOAD(STEP, DST)                                                                           \
    {                                                                                                   \
        const int s_ = d.reverse ? d.S - 1 - (STEP) : (STEP);                                           \
        const int sp_ = d.reverse ? s_ + 1 : s_ - 1;                                                    \
        const float* g_ = tmp + d.gates_off + (int64_t)s_ * G;                                          \
        DST[0] = g_[k]; DST[1] = g_[H + k]; DST[2] = g_[2 * H + k]; DST[3] = g_[3 * H + k];             \
        DST[4] = tmp[d.tc_off + (int64_t)s_ * H + k];                                                   \
        DST[5] = (STEP) > 0 ? tmp[d.c_off + (int64_t)sp_ * H + k] : 0.f;                                \
        DST[6] = gr[d.gout_off + (int64_t)s_ * d.out_ld + k];                                           \
    }
    if (tid < HU) LSTM_MLOAD(d.S - 1, sv)
    for (int step = d.S - 1; step >= 0; --step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const unsigned epoch = (unsigned)(d.S - step);        // 1, 2, ... in execution order
        const int par = (int)(epoch & 1);
        float nx[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tid < HU) {
            if (step > 0) LSTM_MLOAD(step - 1, nx)
            const float ig = sv[0], fg = sv[1], gg = sv[2], og = sv[3], tc = sv[4], cprev = sv[5];
            const float dh = sv[6] + dh_rec;
            const float dc = dc_next + dh * og * (1.f - tc * tc);
            const float dzi = dc * gg * ig * (1.f - ig);
            const float dzf = dc * cprev * fg * (1.f - fg);
            const float dzg = dc * ig * (1.f - gg * gg);
            const float dzo = dh * tc * og * (1.f - og);
            dc_next = dc * fg;
            if (step > 0) {                                   // the last step's dz feeds no further recurrence
                lstm_gran_t* x = xch + par * G;
                gran_store(x + k, epoch, dzi); gran_store(x + H + k, epoch, dzf);
                gran_store(x + 2 * H + k, epoch, dzg); gran_store(x + 3 * H + k, epoch, dzo);
            }
            float* gz = gr + d.gzx_off + (int64_t)s * G;
            gz[k] = dzi; gz[H + k] = dzf; gz[2 * H + k] = dzg; gz[3 * H + k] = dzo;
#pragma unroll
            for (int q = 0; q < 7; ++q) sv[q] = nx[q];
        }
        if (step == 0) break;
#pragma unroll
        for (int i = 0; i < G / 256; ++i) dz_s[tid + 256 * i] = gran_wait(xch + par * G + tid + 256 * i, epoch, status, dead);
        __syncthreads();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < JW; i += 4) {
            const float4 z4 = *reinterpret_cast<const float4*>(dz_s + jq * JW + i);
            a0 = fmaf(w[i], z4.x, a0); a1 = fmaf(w[i + 1], z4.y, a1); a2 = fmaf(w[i + 2], z4.z, a2); a3 = fmaf(w[i + 3], z4.w, a3);
        }
        part_s[jq][kk] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tid < HU) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < JQ; ++q) a += part_s[q][tid];
            dh_rec = a;
        }
        __syncthreads();                                      // part_s / dz_s are rewritten by the next step
    }
#undef LSTM_MLOAD
}

__global__ __launch_bounds__(256) void lstm_multi_clear_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];
    if (!d.multi) return;
    float* x = b.p[SP_TMP] + d.xch_off;
    const int n = 2 * (2 * d.H + 8 * d.H);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = 0.f;
}

// workgroups of the multi-workgroup kernels one CU holds at once, as the runtime reports it for THIS build of them (the
// smaller of the two kernels' answers); the plan sizes co-resident launches from it
int lstm_multi_blocks_per_cu() {
#ifdef HIPSIM
    return 4;
#else
    int f = 0, w = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, lstm_multi_fwd_kernel, 256, 0) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, lstm_multi_bwd_kernel, 256, 0) != hipSuccess) return 0;
    return f < w ? f : w;
#endif
}

// W_hh (4H x H) -> W_hh^T (H x 4H) so that the H > 64 forward reads it lane-contiguously
__global__ __launch_bounds__(256) void lstm_transpose_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];        // by value: no descriptor re-reads after the per-step barriers
    if (d.H <= 64 || d.multi) return;
    const int G = 4 * d.H, n = G * d.H;
    const float* w = b.p[SP_PAR] + d.whh_off;
    float* wt = b.p[SP_TMP] + d.whht_off;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int k = e / G, j = e - k * G;       // consecutive lanes write consecutive j
        wt[e] = w[(int64_t)j * d.H + k];
    }
}

int launch_lstm_transpose(const LstmDesc* dev_descs, int count, int maxH, int multi, Bases b, hipStream_t s) {
    if (count <= 0 || maxH <= 64) return 0;
    if (multi) {       // the multi-workgroup flavour needs no transposed copy; its exchange tags restart from zero
        hipLaunchKernelGGL(lstm_multi_clear_kernel, dim3(4, count), dim3(256), 0, s, dev_descs, b);
        return (int)hipGetLastError();
    }
    int nb = (4 * maxH * maxH + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(lstm_transpose_kernel, dim3(nb, count), dim3(256), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}

static int block_for(int maxH) {
    int t = (4 * maxH + 63) / 64 * 64;
    return t < 64 ? 64 : t;
}

// one workgroup per LSTM_NS sequences once a launch holds enough sequences to fill the chip several times over anyway
#ifdef HIPSIM
static bool lstm_grouped(int, int maxB) { return maxB >= 2 * LSTM_NS; }      // the interpreter's small cases take it too
#else
static bool lstm_grouped(int count, int maxB) { return maxB >= 2 * LSTM_NS && (int64_t)count * maxB >= 2048; }
#endif

int launch_lstm_fwd(const LstmDesc* dev_descs, int count, int maxB, int maxH, int multi, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    if (multi) {
        MST_LAUNCH_CORESIDENT(lstm_multi_fwd_kernel, dim3(LSTM_NB, count), dim3(256), s, dev_descs, b);
        return (int)hipGetLastError();
    }
    if (maxH <= 64 && lstm_grouped(count, maxB))
        hipLaunchKernelGGL(lstm_fwd_group_kernel, dim3((maxB + LSTM_NS - 1) / LSTM_NS, count), dim3(256), 0, s, dev_descs, b);
    else if (maxH <= 64)
        hipLaunchKernelGGL((lstm_fwd_kernel<true>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    else
        hipLaunchKernelGGL((lstm_fwd_kernel<false>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}

int launch_lstm_bwd(const LstmDesc* dev_descs, int count, int maxB, int maxH, int multi, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    if (multi) {
        MST_LAUNCH_CORESIDENT(lstm_multi_bwd_kernel, dim3(LSTM_NB, count), dim3(256), s, dev_descs, b);
        return (int)hipGetLastError();
    }
    if (maxH <= 64 && lstm_grouped(count, maxB))
        hipLaunchKernelGGL(lstm_bwd_group_kernel, dim3((maxB + LSTM_NS - 1) / LSTM_NS, count), dim3(256), 0, s, dev_descs, b);
    else if (maxH <= 64)
        hipLaunchKernelGGL((lstm_bwd_kernel<true>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    else
        hipLaunchKernelGGL((lstm_bwd_kernel<false>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}
