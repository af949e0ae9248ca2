// LSTM recurrence for gfx950 (nn.LSTM as wrapped at style/utils/pytorch.py:19-25; torch gate
// order i,f,g,o; zero initial state; one layer).  The input projection x W_ih^T + b_ih is a
// plain GEMM (gemm.hip); only the strictly sequential part lives here.
//
// One workgroup per sequence, one lane per gate row (4H <= 1024).  For H <= 64 every lane keeps
// its W_hh row in VGPRs for the whole sequence and h_{t-1} is broadcast from LDS, so a step is
// 64 FMAs + two barriers with no global traffic except the streamed zx/gate rows.  Wider
// hidden states (StyleEncoder, H = 192) re-read W_hh from L2 each step.
// The backward kernel runs BPTT with W_hh read coalesced along the hidden index.
#include "mst_common.h"

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

template <bool REG>
__global__ __launch_bounds__(1024) void lstm_fwd_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc& d = descs[blockIdx.y];
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
    float bias = 0.f;
    if (tid < G) {
        bias = b.p[SP_PAR][d.bhh_off + tid];
        if (REG) {
#pragma unroll
            for (int k = 0; k < 64; ++k) w[k] = k < H ? whh[(int64_t)tid * H + k] : 0.f;
        }
    }
    if (tid < 256) h_s[tid] = 0.f;
    float c = 0.f;
    for (int step = 0; step < d.S; ++step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int64_t row = (int64_t)bi * d.S + s;
        __syncthreads();
        if (tid < G) {
            float z = zx[row * G + tid] + bias;
            if (REG) {
#pragma unroll
                for (int k = 0; k < 64; ++k) z = fmaf(w[k], h_s[k], z);
            } else {
                const float* wr = whh + (int64_t)tid * H;
                for (int k = 0; k < H; ++k) z = fmaf(wr[k], h_s[k], z);
            }
            z_s[tid] = z;
        }
        __syncthreads();
        if (tid < H) {
            float ig = sigm(z_s[tid]), fg = sigm(z_s[H + tid]);
            float gg = tanhf(z_s[2 * H + tid]), og = sigm(z_s[3 * H + tid]);
            tmp[d.hprev_off + row * H + tid] = h_s[tid];
            c = fg * c + ig * gg;
            float h = og * tanhf(c);
            float* g = tmp + d.gates_off + row * G;
            g[tid] = ig; g[H + tid] = fg; g[2 * H + tid] = gg; g[3 * H + tid] = og;
            tmp[d.c_off + row * H + tid] = c;
            ws[d.out_off + row * d.out_ld + tid] = h;
            h_s[tid] = h;
        }
    }
}

__global__ __launch_bounds__(1024) void lstm_bwd_kernel(const LstmDesc* __restrict__ descs, Bases b) {
    const LstmDesc& d = descs[blockIdx.y];
    const int bi = blockIdx.x;
    if (bi >= d.B) return;
    const int H = d.H, G = 4 * d.H, tid = threadIdx.x;
    __shared__ float dh_next[256];
    __shared__ float dz_s[1024];
    __shared__ float red_s[1024];
    const float* whh = b.p[SP_PAR] + d.whh_off;
    const float* tmp = b.p[SP_TMP];
    float* gr = b.p[SP_GRAD];
    if (tid < 256) dh_next[tid] = 0.f;
    float dc_next = 0.f;
    __syncthreads();
    for (int step = d.S - 1; step >= 0; --step) {
        const int s = d.reverse ? d.S - 1 - step : step;
        const int sp = d.reverse ? s + 1 : s - 1;          // sequence position of the previous step
        const int64_t row = (int64_t)bi * d.S + s;
        if (tid < H) {
            const float* g = tmp + d.gates_off + row * G;
            float ig = g[tid], fg = g[H + tid], gg = g[2 * H + tid], og = g[3 * H + tid];
            float c = tmp[d.c_off + row * H + tid];
            float cprev = step > 0 ? tmp[d.c_off + ((int64_t)bi * d.S + sp) * H + tid] : 0.f;
            float dh = gr[d.gout_off + row * d.out_ld + tid] + dh_next[tid];
            float tc = tanhf(c);
            float dc = dc_next + dh * og * (1.f - tc * tc);
            float dzi = dc * gg * ig * (1.f - ig);
            float dzf = dc * cprev * fg * (1.f - fg);
            float dzg = dc * ig * (1.f - gg * gg);
            float dzo = dh * tc * og * (1.f - og);
            dc_next = dc * fg;
            dz_s[tid] = dzi; dz_s[H + tid] = dzf; dz_s[2 * H + tid] = dzg; dz_s[3 * H + tid] = dzo;
            float* gz = gr + d.gzx_off + row * G;
            gz[tid] = dzi; gz[H + tid] = dzf; gz[2 * H + tid] = dzg; gz[3 * H + tid] = dzo;
        }
        __syncthreads();
        if (tid < G) {   // dh_{t-1}[k] = sum_j W_hh[j,k] dz[j], four partial sums per k
            const int k = tid % H, part = tid / H;
            float acc = 0.f;
            for (int j = part * H; j < (part + 1) * H; ++j) acc = fmaf(whh[(int64_t)j * H + k], dz_s[j], acc);
            red_s[tid] = acc;
        }
        __syncthreads();
        if (tid < H) dh_next[tid] = (red_s[tid] + red_s[H + tid]) + (red_s[2 * H + tid] + red_s[3 * H + tid]);
        __syncthreads();
    }
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
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(maxB, count), dim3(block_for(maxH)), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}
