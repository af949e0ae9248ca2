// combine() of style/model.py:796-815 for gfx950: a norm-weighted merge over the channel axis,
//   n_c = sqrt(1 + sum(x_c^2)),  out = sum_c x_c n_c / sum_c n_c,
// and its exact backward (the gradient flows through the norms as well):
//   dx_c = g n_c / S + ((a_c - b) / S) x_c / n_c,  a_c = sum(g x_c),  b = sum(g out),  S = sum_c n_c.
// Both directions are a global reduction followed by an elementwise pass, so each is two
// launches; partial sums are written per workgroup and re-summed in index order by every
// consumer workgroup (deterministic, no float atomics).  HBM-bound: x is read twice per
// direction, coalesced along the feature axis.
#include "mst_common.h"

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ int64_t elem_off(const CombineDesc& d, int64_t e) {
    int64_t r = e / d.cols;
    return r * d.ld + (e - r * d.cols);
}

// grid (nblk, Cn): part[c*MAXBLK + blk] = partial sum of squares of slice c
__global__ __launch_bounds__(256) void combine_sumsq_kernel(const CombineDesc* __restrict__ dp, Bases b) {
    const CombineDesc& d = *dp;
    __shared__ float red[4];
    const int c = blockIdx.y;
    const float* x = b.p[SP_WS] + d.x_off + (int64_t)c * d.cs;
    const int64_t n = (int64_t)d.rows * d.cols;
    float acc = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        float v = x[elem_off(d, e)];
        acc = fmaf(v, v, acc);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) b.p[SP_TMP][d.part_off + c * COMBINE_MAXBLK + blockIdx.x] = acc;
}

// grid (nblk): norms from the partials, then out = sum_c x_c n_c / S
__global__ __launch_bounds__(256) void combine_apply_kernel(const CombineDesc* __restrict__ dp, Bases b) {
    const CombineDesc& d = *dp;
    __shared__ float nrm[COMBINE_MAXC + 1];
    float* ws = b.p[SP_WS];
    float* tmp = b.p[SP_TMP];
    if (threadIdx.x < d.Cn) {
        float s = 0.f;
        for (int k = 0; k < d.nblk; ++k) s += tmp[d.part_off + threadIdx.x * COMBINE_MAXBLK + k];
        nrm[threadIdx.x] = sqrtf(1.f + s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f;
        for (int c = 0; c < d.Cn; ++c) S += nrm[c];
        nrm[COMBINE_MAXC] = S;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x <= d.Cn)
        tmp[d.stats_off + threadIdx.x] = threadIdx.x < d.Cn ? nrm[threadIdx.x] : nrm[COMBINE_MAXC];
    const float S = nrm[COMBINE_MAXC];
    const int64_t n = (int64_t)d.rows * d.cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t eo = elem_off(d, e);
        float acc = 0.f;
        for (int c = 0; c < d.Cn; ++c) acc += ws[d.x_off + (int64_t)c * d.cs + eo] * nrm[c];
        ws[d.out_off + e] = acc / S;
    }
}

// grid (nblk): part[blk*(Cn+1) + c] = partial a_c, part[blk*(Cn+1) + Cn] = partial b
__global__ __launch_bounds__(256) void combine_bwd_reduce_kernel(const CombineDesc* __restrict__ dp, Bases b) {
    const CombineDesc& d = *dp;
    __shared__ float red[4];
    float* ws = b.p[SP_WS];
    const float* gr = b.p[SP_GRAD];
    const int64_t n = (int64_t)d.rows * d.cols;
    for (int c = 0; c <= d.Cn; ++c) {
        float acc = 0.f;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
            float g = gr[d.gout_off + e];
            float v = c < d.Cn ? ws[d.x_off + (int64_t)c * d.cs + elem_off(d, e)] : ws[d.out_off + e];
            acc = fmaf(g, v, acc);
        }
        acc = block_sum(acc, red);
        if (threadIdx.x == 0) b.p[SP_TMP][d.part_off + blockIdx.x * (COMBINE_MAXC + 1) + c] = acc;
    }
}

__global__ __launch_bounds__(256) void combine_bwd_apply_kernel(const CombineDesc* __restrict__ dp, Bases b) {
    const CombineDesc& d = *dp;
    __shared__ float coef[COMBINE_MAXC + 1];
    float* ws = b.p[SP_WS];
    float* gr = b.p[SP_GRAD];
    const float* tmp = b.p[SP_TMP];
    if (threadIdx.x <= d.Cn) {
        float s = 0.f;
        for (int k = 0; k < d.nblk; ++k) s += tmp[d.part_off + k * (COMBINE_MAXC + 1) + threadIdx.x];
        coef[threadIdx.x] = s;
    }
    __syncthreads();
    const float S = tmp[d.stats_off + d.Cn];
    const float bsum = coef[d.Cn];
    const int64_t n = (int64_t)d.rows * d.cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t eo = elem_off(d, e);
        const float g = gr[d.gout_off + e];
        for (int c = 0; c < d.Cn; ++c) {
            const float nc = tmp[d.stats_off + c];
            const float x = ws[d.x_off + (int64_t)c * d.cs + eo];
            gr[d.gx_off + (int64_t)c * d.cs + eo] += g * nc / S + ((coef[c] - bsum) / S) * (x / nc);
        }
    }
}

int launch_combine_fwd(const CombineDesc* dev, const CombineDesc& h, Bases b, hipStream_t s) {
    hipLaunchKernelGGL(combine_sumsq_kernel, dim3(h.nblk, h.Cn), dim3(256), 0, s, dev, b);
    hipLaunchKernelGGL(combine_apply_kernel, dim3(h.nblk), dim3(256), 0, s, dev, b);
    return (int)hipGetLastError();
}

int launch_combine_bwd(const CombineDesc* dev, const CombineDesc& h, Bases b, hipStream_t s) {
    hipLaunchKernelGGL(combine_bwd_reduce_kernel, dim3(h.nblk), dim3(256), 0, s, dev, b);
    hipLaunchKernelGGL(combine_bwd_apply_kernel, dim3(h.nblk), dim3(256), 0, s, dev, b);
    return (int)hipGetLastError();
}
