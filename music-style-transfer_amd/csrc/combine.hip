// combine() of style/model.py:796-815 for gfx950: a norm-weighted merge over the channel axis,
//   n_c = sqrt(1 + sum(x_c^2)),  out = sum_c x_c n_c / sum_c n_c,
// and its exact backward (the gradient flows through the norms as well):
//   dx_c = g n_c / S + ((a_c - b) / S) x_c / n_c,  a_c = sum(g x_c),  b = sum(g out),  S = sum_c n_c.
// Both directions are a global reduction followed by an elementwise pass, so each is two
// launches; partial sums are written per workgroup and re-summed in index order by every
// consumer workgroup (deterministic, no float atomics).  blockIdx.y selects the call site, so
// independent sites of one dependency level share launches.  HBM/L2-bound: x is read twice per
// direction, coalesced along the feature axis; the backward reduce reads g once for all channels.
#include "mst_common.h"

// sums `nv` per-lane values across the 256-lane workgroup; result for value v lands in red[v]
template <int MC = COMBINE_MAXC>
__device__ __forceinline__ void block_sum_multi(float* vals, int nv, float (*part)[COMBINE_MAXC + 1], float* red) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v <= MC; ++v) {                 // static indices keep vals[] in registers; nv is workgroup-uniform
        if (v < nv) {
            float x = vals[v];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
            if (lane == 0) part[wv][v] = x;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) red[threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    __syncthreads();
}

// element e of a rows x cols slice with row stride ld (32-bit index math: a slice has < 2^31 elements)
__device__ __forceinline__ int64_t elem_off(const CombineDesc& d, int e) {
    const unsigned r = (unsigned)e / (unsigned)d.cols;
    return (int64_t)r * d.ld + (int)((unsigned)e - r * (unsigned)d.cols);
}

// part[blk*(MAXC+1) + c] = partial sum of squares of slice c over this workgroup's elements
__global__ __launch_bounds__(256) void combine_sumsq_kernel(const CombineDesc* __restrict__ descs, Bases b) {
    const CombineDesc d = descs[blockIdx.y];     // by value: fields stay in registers across barriers
    if ((int)blockIdx.x >= d.nblk) return;
    __shared__ float part[4][COMBINE_MAXC + 1];
    __shared__ float red[COMBINE_MAXC + 1];
    const float* x = b.p[SP_WS] + d.x_off;
    const int64_t n = (int64_t)d.rows * d.cols;
    float acc[COMBINE_MAXC + 1];
#pragma unroll
    for (int c = 0; c <= COMBINE_MAXC; ++c) acc[c] = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < (int)n; e += d.nblk * 256) {
        const int64_t eo = elem_off(d, e);
#pragma unroll
        for (int c = 0; c < COMBINE_MAXC; ++c) {
            if (c < d.Cn) { const float v = x[(int64_t)c * d.cs + eo]; acc[c] = fmaf(v, v, acc[c]); }
        }
    }
    block_sum_multi(acc, d.Cn, part, red);
    if ((int)threadIdx.x < d.Cn) b.p[SP_TMP][d.part_off + blockIdx.x * (COMBINE_MAXC + 1) + threadIdx.x] = red[threadIdx.x];
}

// norms from the partials, then out = sum_c x_c n_c / S
__global__ __launch_bounds__(256) void combine_apply_kernel(const CombineDesc* __restrict__ descs, Bases b) {
    const CombineDesc d = descs[blockIdx.y];     // by value: fields stay in registers across barriers
    if ((int)blockIdx.x >= d.nblk) return;
    __shared__ float nrm[COMBINE_MAXC + 1];
    __shared__ float part[4][COMBINE_MAXC + 1];
    __shared__ float red[COMBINE_MAXC + 1];
    float* ws = b.p[SP_WS];
    float* tmp = b.p[SP_TMP];
    {   // re-sum the per-workgroup partials: lane k takes partial k, then a fixed-order tree (nblk <= 256)
        float vals[COMBINE_MAXC + 1];
#pragma unroll
        for (int c = 0; c <= COMBINE_MAXC; ++c)
            vals[c] = (c < d.Cn && (int)threadIdx.x < d.nblk) ? tmp[d.part_off + threadIdx.x * (COMBINE_MAXC + 1) + c] : 0.f;
        block_sum_multi(vals, d.Cn, part, red);
    }
    if ((int)threadIdx.x < d.Cn) nrm[threadIdx.x] = sqrtf(1.f + red[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f;
        for (int c = 0; c < d.Cn; ++c) S += nrm[c];
        nrm[COMBINE_MAXC] = S;
    }
    __syncthreads();
    if (blockIdx.x == 0 && (int)threadIdx.x <= d.Cn)
        tmp[d.stats_off + threadIdx.x] = (int)threadIdx.x < d.Cn ? nrm[threadIdx.x] : nrm[COMBINE_MAXC];
    const float S = nrm[COMBINE_MAXC];
    const int64_t n = (int64_t)d.rows * d.cols;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < (int)n; e += d.nblk * 256) {
        const int64_t eo = elem_off(d, e);
        float acc = 0.f;
        for (int c = 0; c < d.Cn; ++c) acc += ws[d.x_off + (int64_t)c * d.cs + eo] * nrm[c];
        ws[d.out_off + e] = acc / S;
    }
}

// part[blk*(MAXC+1) + c] = partial a_c, part[blk*(MAXC+1) + Cn] = partial b
__global__ __launch_bounds__(256) void combine_bwd_reduce_kernel(const CombineDesc* __restrict__ descs, Bases b) {
    const CombineDesc d = descs[blockIdx.y];     // by value: fields stay in registers across barriers
    if ((int)blockIdx.x >= d.nblk) return;
    __shared__ float part[4][COMBINE_MAXC + 1];
    __shared__ float red[COMBINE_MAXC + 1];
    const float* ws = b.p[SP_WS];
    const float* gr = b.p[SP_GRAD];
    const int64_t n = (int64_t)d.rows * d.cols;
    float acc[COMBINE_MAXC + 1];
#pragma unroll
    for (int c = 0; c <= COMBINE_MAXC; ++c) acc[c] = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < (int)n; e += d.nblk * 256) {
        const int64_t eo = elem_off(d, e);
        const float g = gr[d.gout_off + e];
#pragma unroll
        for (int c = 0; c < COMBINE_MAXC; ++c) {
            if (c < d.Cn) acc[c] = fmaf(g, ws[d.x_off + (int64_t)c * d.cs + eo], acc[c]);
        }
        acc[COMBINE_MAXC] = fmaf(g, ws[d.out_off + e], acc[COMBINE_MAXC]);
    }
    // move b next to the a_c so one multi-value reduction covers all Cn+1 sums
    float vals[COMBINE_MAXC + 1];
#pragma unroll
    for (int c = 0; c < COMBINE_MAXC; ++c) vals[c] = acc[c];
    vals[COMBINE_MAXC] = 0.f;
#pragma unroll
    for (int c = 0; c <= COMBINE_MAXC; ++c) if (c == d.Cn) vals[c] = acc[COMBINE_MAXC];
    block_sum_multi(vals, d.Cn + 1, part, red);
    if ((int)threadIdx.x <= d.Cn) b.p[SP_TMP][d.part_off + blockIdx.x * (COMBINE_MAXC + 1) + threadIdx.x] = red[threadIdx.x];
}

__global__ __launch_bounds__(256) void combine_bwd_apply_kernel(const CombineDesc* __restrict__ descs, Bases b) {
    const CombineDesc d = descs[blockIdx.y];     // by value: fields stay in registers across barriers
    if ((int)blockIdx.x >= d.nblk) return;
    __shared__ float coef[COMBINE_MAXC + 1];
    __shared__ float nc_s[COMBINE_MAXC + 1];
    const float* ws = b.p[SP_WS];
    float* gr = b.p[SP_GRAD];
    const float* tmp = b.p[SP_TMP];
    __shared__ float part[4][COMBINE_MAXC + 1];
    {
        float vals[COMBINE_MAXC + 1];
#pragma unroll
        for (int c = 0; c <= COMBINE_MAXC; ++c)
            vals[c] = (c <= d.Cn && (int)threadIdx.x < d.nblk) ? tmp[d.part_off + threadIdx.x * (COMBINE_MAXC + 1) + c] : 0.f;
        block_sum_multi(vals, d.Cn + 1, part, coef);
    }
    if ((int)threadIdx.x <= d.Cn) nc_s[threadIdx.x] = tmp[d.stats_off + threadIdx.x];       // n_c ..., S
    __syncthreads();
    const float S = nc_s[d.Cn];
    const float bsum = coef[d.Cn];
    __shared__ float fa_s[COMBINE_MAXC], fb_s[COMBINE_MAXC];      // per-channel factors: dx_c = g fa_c + fb_c x_c
    if ((int)threadIdx.x < d.Cn) {
        const float nc = nc_s[threadIdx.x];
        fa_s[threadIdx.x] = nc / S; fb_s[threadIdx.x] = ((coef[threadIdx.x] - bsum) / S) / nc;
    }
    __syncthreads();
    const int64_t n = (int64_t)d.rows * d.cols;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < (int)n; e += d.nblk * 256) {
        const int64_t eo = elem_off(d, e);
        const float g = gr[d.gout_off + e];
        for (int c = 0; c < d.Cn; ++c) {
            const float x = ws[d.x_off + (int64_t)c * d.cs + eo];
            float* gx = gr + d.gx_off + (int64_t)c * d.cs + eo;
            const float v = fmaf(g, fa_s[c], fb_s[c] * x);
            *gx = d.first ? v : *gx + v;
        }
    }
}

// Small sites (<= COMBINE_SMALL elements per slice: the per-bar tensors, the style vector) need no cross-workgroup
// re-sum: one workgroup per site does reduction and elementwise pass in ONE launch instead of two.
// MC: compile-time bound on the channel count (4: every site of the model up to 4 channels — unrolled loops and shuffle
// reductions over 32 mostly absent channels were most of these kernels' ~10 us; 32: the general case)
template <int MC>
__global__ __launch_bounds__(256) void combine_small_fwd_kernel(const CombineDesc* __restrict__ descs, Bases b) {
    const CombineDesc d = descs[blockIdx.x];
    __shared__ float nrm[COMBINE_MAXC + 1];
    __shared__ float part[4][COMBINE_MAXC + 1];
    __shared__ float red[COMBINE_MAXC + 1];
    float* ws = b.p[SP_WS];
    float* tmp = b.p[SP_TMP];
    const int n = d.rows * d.cols;
    float acc[MC + 1];
#pragma unroll
    for (int c = 0; c <= MC; ++c) acc[c] = 0.f;
    // U elements per lane and trip, all their loads issued before the first use: one workgroup walks the whole site, and a
    // trip per element was a chain of n / 256 dependent L2 round trips (16 us for 4096 elements).  Per lane the elements
    // are still accumulated in ascending order.
    constexpr int U = MC <= 4 ? 4 : 1;
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * U) {
        float v[U][MC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            const bool ok = e < n;
            const int64_t eo = elem_off(d, ok ? e : 0);
#pragma unroll
            for (int c = 0; c < MC; ++c) v[u][c] = (ok && c < d.Cn) ? ws[d.x_off + (int64_t)c * d.cs + eo] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int c = 0; c < MC; ++c) acc[c] = fmaf(v[u][c], v[u][c], acc[c]);
        }
    }
    block_sum_multi<MC>(acc, d.Cn, part, red);
    if ((int)threadIdx.x < d.Cn) nrm[threadIdx.x] = sqrtf(1.f + red[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f;
        for (int c = 0; c < d.Cn; ++c) S += nrm[c];
        nrm[COMBINE_MAXC] = S;
    }
    __syncthreads();
    if ((int)threadIdx.x <= d.Cn) tmp[d.stats_off + threadIdx.x] = (int)threadIdx.x < d.Cn ? nrm[threadIdx.x] : nrm[COMBINE_MAXC];
    const float S = nrm[COMBINE_MAXC];
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * U) {
        float v[U][MC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            const bool ok = e < n;
            const int64_t eo = elem_off(d, ok ? e : 0);
#pragma unroll
            for (int c = 0; c < MC; ++c) v[u][c] = (ok && c < d.Cn) ? ws[d.x_off + (int64_t)c * d.cs + eo] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) if (c < d.Cn) a += v[u][c] * nrm[c];
            if (e < n) ws[d.out_off + e] = a / S;
        }
    }
}

template <int MC>
__global__ __launch_bounds__(256) void combine_small_bwd_kernel(const CombineDesc* __restrict__ descs, Bases b) {
    const CombineDesc d = descs[blockIdx.x];
    __shared__ float coef[COMBINE_MAXC + 1];
    __shared__ float nc_s[COMBINE_MAXC + 1];
    __shared__ float part[4][COMBINE_MAXC + 1];
    const float* ws = b.p[SP_WS];
    float* gr = b.p[SP_GRAD];
    const float* tmp = b.p[SP_TMP];
    const int n = d.rows * d.cols;
    float acc[MC + 1];
#pragma unroll
    for (int c = 0; c <= MC; ++c) acc[c] = 0.f;
    constexpr int U = MC <= 4 ? 4 : 1;                 // loads of U elements in flight per trip (see combine_small_fwd_kernel)
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * U) {
        float v[U][MC], g[U], o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            const bool ok = e < n;
            const int64_t eo = elem_off(d, ok ? e : 0);
            g[u] = ok ? gr[d.gout_off + e] : 0.f;
            o[u] = ok ? ws[d.out_off + e] : 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) v[u][c] = (ok && c < d.Cn) ? ws[d.x_off + (int64_t)c * d.cs + eo] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + 256 * u < n) {
#pragma unroll
                for (int c = 0; c < MC; ++c) if (c < d.Cn) acc[c] = fmaf(g[u], v[u][c], acc[c]);
                acc[MC] = fmaf(g[u], o[u], acc[MC]);
            }
        }
    }
    float vals[MC + 1];
#pragma unroll
    for (int c = 0; c < MC; ++c) vals[c] = acc[c];
    vals[MC] = 0.f;
#pragma unroll
    for (int c = 0; c <= MC; ++c) if (c == d.Cn) vals[c] = acc[MC];
    block_sum_multi<MC>(vals, d.Cn + 1, part, coef);
    if ((int)threadIdx.x <= d.Cn) nc_s[threadIdx.x] = tmp[d.stats_off + threadIdx.x];       // n_c ..., S
    __syncthreads();
    const float S = nc_s[d.Cn];
    const float bsum = coef[d.Cn];
    // per-channel factors once per lane (two IEEE divisions per element and channel were most of the elementwise pass):
    // dx_c = g n_c / S + ((b_c - b) / S) x_c / n_c
    float fa[MC], fb[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const float nc = c < d.Cn ? nc_s[c] : 1.f;
        fa[c] = nc / S; fb[c] = c < d.Cn ? ((coef[c] - bsum) / S) / nc : 0.f;
    }
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * U) {
        float x[U][MC], old[U][MC], g[U];
        int64_t eo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            const bool ok = e < n;
            eo[u] = elem_off(d, ok ? e : 0);
            g[u] = ok ? gr[d.gout_off + e] : 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                const bool okc = ok && c < d.Cn;
                x[u][c] = okc ? ws[d.x_off + (int64_t)c * d.cs + eo[u]] : 0.f;
                old[u][c] = (okc && !d.first) ? gr[d.gx_off + (int64_t)c * d.cs + eo[u]] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + 256 * u < n) {
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    if (c < d.Cn) {
                        const float v = fmaf(g[u], fa[c], fb[c] * x[u][c]);
                        gr[d.gx_off + (int64_t)c * d.cs + eo[u]] = d.first ? v : old[u][c] + v;
                    }
                }
            }
        }
    }
}

// all_small: every site of the (merged) launch has <= COMBINE_SMALL elements per slice (2: and at most 4 channels)
int launch_combine_fwd(const CombineDesc* dev, int count, int max_nblk, int all_small, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    if (all_small) {
        if (all_small == 2) hipLaunchKernelGGL(combine_small_fwd_kernel<4>, dim3(count), dim3(256), 0, s, dev, b);
        else hipLaunchKernelGGL(combine_small_fwd_kernel<COMBINE_MAXC>, dim3(count), dim3(256), 0, s, dev, b);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(combine_sumsq_kernel, dim3(max_nblk, count), dim3(256), 0, s, dev, b);
    hipLaunchKernelGGL(combine_apply_kernel, dim3(max_nblk, count), dim3(256), 0, s, dev, b);
    return (int)hipGetLastError();
}

int launch_combine_bwd(const CombineDesc* dev, int count, int max_nblk, int all_small, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    if (all_small) {
        if (all_small == 2) hipLaunchKernelGGL(combine_small_bwd_kernel<4>, dim3(count), dim3(256), 0, s, dev, b);
        else hipLaunchKernelGGL(combine_small_bwd_kernel<COMBINE_MAXC>, dim3(count), dim3(256), 0, s, dev, b);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(combine_bwd_reduce_kernel, dim3(max_nblk, count), dim3(256), 0, s, dev, b);
    hipLaunchKernelGGL(combine_bwd_apply_kernel, dim3(max_nblk, count), dim3(256), 0, s, dev, b);
    return (int)hipGetLastError();
}

// ---- tiled plans: the halves of a combine on their own (the ranks' partial sums meet in between), strided row copies
// and the fold / spread pair around an exchange (mst_common.h)
int launch_combine_phase(const CombineDesc* dev, int nblk, int which, Bases b, hipStream_t s) {
    if (which == 0) hipLaunchKernelGGL(combine_sumsq_kernel, dim3(nblk, 1), dim3(256), 0, s, dev, b);
    else if (which == 1) hipLaunchKernelGGL(combine_apply_kernel, dim3(nblk, 1), dim3(256), 0, s, dev, b);
    else if (which == 2) hipLaunchKernelGGL(combine_bwd_reduce_kernel, dim3(nblk, 1), dim3(256), 0, s, dev, b);
    else hipLaunchKernelGGL(combine_bwd_apply_kernel, dim3(nblk, 1), dim3(256), 0, s, dev, b);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const CopyDesc* __restrict__ dp, int backward, Bases b) {
    const CopyDesc d = dp[0];
    const int total = d.na * d.nb * d.cols;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e % d.cols, ab = e / d.cols, bb = ab % d.nb, a = ab / d.nb;
        const int64_t so = d.src_off + (int64_t)a * d.src_sa + (int64_t)bb * d.src_sb + j;
        const int64_t dof = d.dst_off + (int64_t)a * d.dst_sa + (int64_t)bb * d.dst_sb + j;
        if (!backward) b.p[SP_WS][dof] = b.p[SP_WS][so];
        else { float* g = b.p[SP_GRAD] + so; const float v = b.p[SP_GRAD][dof]; *g = d.first ? v : *g + v; }
    }
}

int launch_copy_rows(const CopyDesc* dev, const CopyDesc& host, int backward, Bases b, hipStream_t s) {
    const int total = host.na * host.nb * host.cols;
    int nb = (total + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) return 0;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(nb), dim3(256), 0, s, dev, backward, b);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(64) void fold_kernel(const FoldDesc* __restrict__ dp, int spread, Bases b) {
    const FoldDesc d = dp[0];
    float* part = b.p[d.space] + d.part_off;
    float* sum = b.p[SP_TMP] + d.sum_off;
    for (int c = threadIdx.x; c < d.ncols; c += 64) {
        if (!spread) {
            float a = 0.f;
            for (int r = 0; r < d.nrows; ++r) a += part[(int64_t)r * d.row_stride + (int64_t)c * d.col_stride];
            sum[c] = a;
        } else {
            part[(int64_t)c * d.col_stride] = sum[c];
            for (int r = 1; r < d.nrows; ++r) part[(int64_t)r * d.row_stride + (int64_t)c * d.col_stride] = 0.f;
        }
    }
}

int launch_fold(const FoldDesc* dev, int spread, Bases b, hipStream_t s) {
    hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(64), 0, s, dev, spread, b);
    return (int)hipGetLastError();
}
