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

__device__ __forceinline__ float sigm(float x) { return __fdividef(1.f, 1.f + MST_FAST_EXP(-x)); }
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
            zq[q] = zx[((int64_t)bi * d.S + s0) * G + q * H + tid];
        }
    }
    if (tid < 256) h_s[tid] = 0.f;
    float c = 0.f;
    for (int step = 0; step < d.S; ++step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int64_t row = (int64_t)bi * d.S + s;
        float zn[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < H && step + 1 < d.S) {                    // next step's zx row: in flight under this step's matvec
            const int sn = d.reverse ? s - 1 : s + 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) zn[q] = zx[((int64_t)bi * d.S + sn) * G + q * H + tid];
        }
        MST_LDS_BARRIER();
        if (REG) {
            if (tid < G) {
                float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;       // 4 independent FMA chains
#pragma unroll
                for (int k = 0; k < 64; k += 4) {
                    z0 = fmaf(w[k], h_s[k], z0); z1 = fmaf(w[k + 1], h_s[k + 1], z1);
                    z2 = fmaf(w[k + 2], h_s[k + 2], z2); z3 = fmaf(w[k + 3], h_s[k + 3], z3);
                }
                z_s[tid] = (z0 + z1) + (z2 + z3);
            }
        } else if (tid < G) {
            // thread per gate row over the transposed copy: lane-contiguous (coalesced) reads, 4 chains
            const float* wt = tmp + d.whht_off + tid;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int k = 0;
            for (; k + 4 <= H; k += 4) {
                a0 = fmaf(wt[(int64_t)k * G], h_s[k], a0);
                a1 = fmaf(wt[(int64_t)(k + 1) * G], h_s[k + 1], a1);
                a2 = fmaf(wt[(int64_t)(k + 2) * G], h_s[k + 2], a2);
                a3 = fmaf(wt[(int64_t)(k + 3) * G], h_s[k + 3], a3);
            }
            for (; k < H; ++k) a0 = fmaf(wt[(int64_t)k * G], h_s[k], a0);
            z_s[tid] = (a0 + a1) + (a2 + a3);
        }
        MST_LDS_BARRIER();
        if (tid < H) {
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
#pragma unroll
            for (int q = 0; q < 4; ++q) zq[q] = zn[q];
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
    __shared__ float red_s[1024];      // four partial sums of dh_{t-1} per hidden unit, consumed by the next step
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
    red_s[tid] = 0.f;
    float dc_next = 0.f;
    // streamed operands of a step (saved gates, cell states, incoming gradient), prefetched one step ahead
    float sv[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define LSTM_LOAD(STEP, DST)                                                                            \
    {                                                                                                   \
        const int s_ = d.reverse ? d.S - 1 - (STEP) : (STEP);                                           \
        const int sp_ = d.reverse ? s_ + 1 : s_ - 1;                                                    \
        const int64_t row_ = (int64_t)bi * d.S + s_;                                                    \
        const float* g_ = tmp + d.gates_off + row_ * G;                                                 \
        DST[0] = g_[tid]; DST[1] = g_[H + tid]; DST[2] = g_[2 * H + tid]; DST[3] = g_[3 * H + tid];     \
        DST[4] = tmp[d.tc_off + row_ * H + tid];                                                        \
        DST[5] = (STEP) > 0 ? tmp[d.c_off + ((int64_t)bi * d.S + sp_) * H + tid] : 0.f;                 \
        DST[6] = gr[d.gout_off + row_ * d.out_ld + tid];                                                \
    }
    if (tid < H) LSTM_LOAD(d.S - 1, sv)
    __syncthreads();
    for (int step = d.S - 1; step >= 0; --step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int64_t row = (int64_t)bi * d.S + s;
        float nx[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tid < H) {
            if (step > 0) LSTM_LOAD(step - 1, nx)
            const float ig = sv[0], fg = sv[1], gg = sv[2], og = sv[3], tc = sv[4], cprev = sv[5];
            const float dh = sv[6] + ((red_s[tid] + red_s[H + tid]) + (red_s[2 * H + tid] + red_s[3 * H + tid]));
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
        if (tid < G) {   // dh_{t-1}[k] = sum_j W_hh[j,k] dz[j], four partial sums per k (summed by the next step)
            if (REG) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const int base = part * H;
#pragma unroll
                for (int jj = 0; jj < 64; jj += 4) {
                    a0 = fmaf(w[jj], dz_s[min(base + jj, G - 1)], a0);
                    a1 = fmaf(w[jj + 1], dz_s[min(base + jj + 1, G - 1)], a1);
                    a2 = fmaf(w[jj + 2], dz_s[min(base + jj + 2, G - 1)], a2);
                    a3 = fmaf(w[jj + 3], dz_s[min(base + jj + 3, G - 1)], a3);
                }
                acc = (a0 + a1) + (a2 + a3);
            } else {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const float* wc = whh + (int64_t)part * H * H + kk;      // coalesced along kk
                const float* dzp = dz_s + part * H;
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
        if (tid < H) {
#pragma unroll
            for (int q = 0; q < 7; ++q) sv[q] = nx[q];     // the prefetch landed under the matvec above
        }
        if (tid < G) red_s[tid] = acc;                    // phase A's reads of red_s ended before the barrier above
        MST_LDS_BARRIER();
    }
#undef LSTM_LOAD
}

// W_hh (4H x H) -> W_hh^T (H x 4H) so that the H > 64 forward reads it lane-contiguously
__global__ __launch_bounds__(256) void lstm_transpose_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc d = descs[blockIdx.y];        // by value: no descriptor re-reads after the per-step barriers
    if (d.H <= 64) return;
    const int G = 4 * d.H, n = G * d.H;
    const float* w = b.p[SP_PAR] + d.whh_off;
    float* wt = b.p[SP_TMP] + d.whht_off;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int k = e / G, j = e - k * G;       // consecutive lanes write consecutive j
        wt[e] = w[(int64_t)j * d.H + k];
    }
}

int launch_lstm_transpose(const LstmDesc* dev_descs, int count, int maxH, Bases b, hipStream_t s) {
    if (count <= 0 || maxH <= 64) return 0;
    int nb = (4 * maxH * maxH + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(lstm_transpose_kernel, dim3(nb, count), dim3(256), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}

static int block_for(int maxH) {
    int t = (4 * maxH + 63) / 64 * 64;
    return t < 64 ? 64 : t;
}

int launch_lstm_fwd(const LstmDesc* dev_descs, int count, int maxB, int maxH, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    if (maxH <= 64)
        hipLaunchKernelGGL((lstm_fwd_kernel<true>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    else
        hipLaunchKernelGGL((lstm_fwd_kernel<false>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}

int launch_lstm_bwd(const LstmDesc* dev_descs, int count, int maxB, int maxH, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    if (maxH <= 64)
        hipLaunchKernelGGL((lstm_bwd_kernel<true>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    else
        hipLaunchKernelGGL((lstm_bwd_kernel<false>), dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}
