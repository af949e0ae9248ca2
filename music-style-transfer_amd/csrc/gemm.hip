// Generic descriptor-driven fp32 GEMM for gfx950: C[m,n] = epi(sum_k A(m,k) B(k,n)).
//
// Every dense contraction of the model goes through this kernel: the broadcast-concat
// Linears (cat_with_broadcast + nn.Linear, style/utils/pytorch.py:54-65), the note-axis
// Conv1d as an implicit-im2col GEMM (style/model.py:46-53,82), the LSTM input projections,
// and all their weight/input gradients.  Operands are *accessors*: the concatenated,
// broadcast input row is never materialised in HBM — each LDS tile is gathered straight from
// the source tensors — and bias/activation (or the activation derivative on the backward
// side) are fused into the tile load / epilogue.
//
// Tiling: 64x64x16 per 256-thread workgroup (4 waves of 64), 4x4 register micro-tile per
// lane, LDS tiles stored k-major with a +4 pad so both the float4 fragment reads and the
// transposed tile writes are (at worst 2-way) bank-conflict free.  blockIdx.y selects the
// descriptor, so independent small GEMMs share one launch; blockIdx.z is the split of the
// reduction dimension for weight gradients (deterministic slabs, reduced later in order).
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

__device__ __forceinline__ float fetch(const Operand& o, const Bases& b, int i, int j) {
    switch (o.kind) {
    case OPK_DENSE:
        if (j == o.ones_at) return 1.f;
        return b.p[o.space][o.off + (int64_t)i * o.si + (int64_t)j * o.sj];
    case OPK_CAT: {
        if (j == o.ones_at) return 1.f;
        int r3 = i % o.d3; int t = i / o.d3;
        int r2 = t % o.d2; t /= o.d2;
        int r1 = t % o.d1; int r0 = t / o.d1;
        int sidx = 0;
#pragma unroll
        for (int q = 1; q < MAX_SEG; ++q)
            if (q < o.nseg && j >= o.seg[q].start) sidx = q;
        const Seg& sg = o.seg[sidx];
        int64_t row = (int64_t)r0 * sg.s[0] + (int64_t)r1 * sg.s[1] + (int64_t)r2 * sg.s[2] + (int64_t)r3 * sg.s[3];
        return b.p[sg.space][sg.off + row * sg.ld + (j - sg.start)];
    }
    case OPK_ACTGRAD: {
        int row = o.transposed ? j : i;
        int col = o.transposed ? i : j;
        int64_t idx = (int64_t)row * o.ld + col;
        return b.p[o.space][o.off + idx] * act_bwd(o.act, b.p[o.space2][o.off2 + idx], col);
    }
    case OPK_IM2COL: {   // i = (p, octave), j = (fraction, tap, feature)
        if (j == o.ones_at) return 1.f;
        int p = i >> 3, oc = i & 7;
        int f = j / (CONV_K * NPF), rem = j - f * (CONV_K * NPF);
        int e = (NDEG * oc - CONV_PAD) * NPF + rem;
        if (e < 0 || e >= NPN * NPF) return 0.f;
        return b.p[o.space][o.off + (int64_t)p * (NF * NPN * NPF) + f * (NPN * NPF) + e];
    }
    case OPK_PERMW: {    // i = (a, b, c) in memory order of the activations, j = out feature.
        // conv: (fraction, tap, feature) reads W[oc, fraction*5+feature, tap]  (pb=14, pc=5)
        // unpitched linear: (fraction, note, feature) reads W[j, fraction*94 + feature*47 + note]
        const int bc = o.pb * o.pc;
        int a = i / bc, rem = i - a * bc;
        int bb = rem / o.pc, c = rem - bb * o.pc;
        return b.p[o.space][o.off + (int64_t)j * o.ld + a * bc + c * o.pb + bb];
    }
    case OPK_CONVGRAD: { // i = out channel, j = (p, octave)
        int p = j >> 3, oc = j & 7;
        int64_t idx = (int64_t)p * (o.oc * NOCT) + i * NOCT + oc;
        return b.p[o.space][o.off + idx] * act_bwd(ACT_LEAKY, b.p[o.space2][o.off2 + idx], 0);
    }
    }
    return 0.f;
}

__device__ __forceinline__ void store_out(const GemmDesc& d, const Bases& b, int m, int n, int split, float acc) {
    const OutSpec& o = d.out;
    switch (o.kind) {
    case OUT_STORE: {
        float v = acc;
        if (o.bias_space >= 0) v += b.p[o.bias_space][o.bias_off + n];
        b.p[o.space][o.off + (int64_t)m * o.ldc + n] = act_fwd(o.act, v, n);
        break;
    }
    case OUT_ACCUM:
        b.p[o.space][o.off + (int64_t)m * o.ldc + n] += acc;
        break;
    case OUT_CONV: {     // m = (p, octave), n = out channel -> x1[p, n*8 + octave]
        float v = acc + b.p[o.bias_space][o.bias_off + n];
        b.p[o.space][o.off + (int64_t)(m >> 3) * o.ldc + n * NOCT + (m & 7)] = act_fwd(ACT_LEAKY, v, 0);
        break;
    }
    case OUT_SLAB: {
        int64_t base = o.off + (int64_t)split * o.slab_stride;
        int64_t idx = n < o.wcols ? (int64_t)m * o.wcols + n : (int64_t)d.M * o.wcols + m;
        b.p[o.space][base + idx] = acc;
        break;
    }
    case OUT_PERMW_SLAB: {   // m = out feature, n = (a, b, c) | bias column
        int64_t base = o.off + (int64_t)split * o.slab_stride;
        int64_t idx;
        if (n < o.wcols) {
            const int bc = o.pb * o.pc;
            int a = n / bc, rem = n - a * bc;
            int bb = rem / o.pc, c = rem - bb * o.pc;
            idx = (int64_t)m * o.wcols + a * bc + c * o.pb + bb;
        } else {
            idx = (int64_t)d.M * o.wcols + m;
        }
        b.p[o.space][base + idx] = acc;
        break;
    }
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(const GemmDesc* __restrict__ descs, Bases b) {
    const GemmDesc& d = descs[blockIdx.y];
    const int tiles_n = (d.N + GEMM_BN - 1) / GEMM_BN;
    const int tiles_m = (d.M + GEMM_BM - 1) / GEMM_BM;
    const int tile = blockIdx.x;
    const int split = blockIdx.z;
    if (tile >= tiles_m * tiles_n || split >= d.ksplit) return;   // block-uniform
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    int kchunk = (d.K + d.ksplit - 1) / d.ksplit;
    kchunk = (kchunk + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    const int k0 = split * kchunk;
    const int k1 = min(d.K, k0 + kchunk);

    __shared__ float As[GEMM_BK][GEMM_BM + 4];
    __shared__ float Bs[GEMM_BK][GEMM_BN + 4];
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int kt = k0; kt < k1; kt += GEMM_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + i * 256;
            int ml, kl;
            if (d.A.kfast) { ml = e / GEMM_BK; kl = e % GEMM_BK; } else { ml = e % GEMM_BM; kl = e / GEMM_BM; }
            int m = tm * GEMM_BM + ml, k = kt + kl;
            As[kl][ml] = (m < d.M && k < k1) ? fetch(d.A, b, m, k) : 0.f;
            int nl;
            if (d.B.kfast) { nl = e / GEMM_BK; kl = e % GEMM_BK; } else { nl = e % GEMM_BN; kl = e / GEMM_BN; }
            int n = tn * GEMM_BN + nl;
            k = kt + kl;
            Bs[kl][nl] = (n < d.N && k < k1) ? fetch(d.B, b, k, n) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GEMM_BK; ++kk) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; bb[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = tm * GEMM_BM + ty * 4 + i;
        if (m >= d.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = tn * GEMM_BN + tx * 4 + j;
            if (n < d.N) store_out(d, b, m, n, split, acc[i][j]);
        }
    }
}

int launch_gemm(const GemmDesc* dev_descs, int count, int max_tiles, int max_split, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(gemm_kernel, dim3(max_tiles, count, max_split), dim3(256), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Segment reduce: the gradient of a broadcast segment of a concatenated input is the sum, over
// the broadcast row-space dims, of its columns of dAcat (the backward of cat_with_broadcast's
// expand, style/utils/pytorch.py:60-63).  One workgroup per destination row; 4 waves split the
// reduced rows, 64 lanes walk the segment's columns (coalesced), fixed summation order.
__global__ __launch_bounds__(256) void segred_kernel(const SegRedDesc* __restrict__ descs, Bases b) {
    const SegRedDesc& d = descs[blockIdx.y];
    const int idx = blockIdx.x;
    if (idx >= d.nidx) return;
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    // kept coordinates of this destination row
    int kc[4], t = idx;
#pragma unroll
    for (int q = 3; q >= 0; --q) { kc[q] = t % d.kd[q]; t /= d.kd[q]; }
    int rd[4], nred = 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) { rd[q] = d.kd[q] == 1 ? d.d[q] : 1; nred *= rd[q]; }
    const float* src = b.p[SP_TMP] + d.src_off;
    float* dst = b.p[SP_GRAD] + d.dst_off + (int64_t)idx * d.dst_ld;
    for (int w0 = 0; w0 < d.width; w0 += 64) {
        int w = w0 + lane;
        float acc = 0.f;
        if (w < d.width) {
            for (int rr = grp; rr < nred; rr += 4) {
                int c[4], u = rr;
#pragma unroll
                for (int q = 3; q >= 0; --q) { c[q] = kc[q] + u % rd[q]; u /= rd[q]; }
                int64_t row = (((int64_t)c[0] * d.d[1] + c[1]) * d.d[2] + c[2]) * d.d[3] + c[3];
                acc += src[row * d.src_ld + d.start + w];
            }
        }
        part[grp][lane] = acc;
        __syncthreads();
        if (grp == 0 && w < d.width) dst[w] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        __syncthreads();
    }
}

int launch_segred(const SegRedDesc* dev_descs, int count, int max_idx, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(segred_kernel, dim3(max_idx, count), dim3(256), 0, s, dev_descs, b);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Deferred weight-gradient reduction: gpar[dst+i] += sum_s slab_s[i], splits summed in index
// order (bitwise reproducible, unlike float atomics).  p.grad accumulates across iterations
// exactly like loss.backward() does in train-model.py:126.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabEntry* __restrict__ ents, Bases b) {
    const SlabEntry& e = ents[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < e.count; i += gridDim.x * 256) {
        const float* src = b.p[SP_TMP] + e.src + i;
        float acc = 0.f;
        for (int s = 0; s < e.splits; ++s) acc += src[(int64_t)s * e.stride];
        b.p[SP_GPAR][e.dst + i] += acc;
    }
}

int launch_slab_reduce(const SlabEntry* dev, int count, int max_count, Bases b, hipStream_t s) {
    if (count <= 0) return 0;
    int gx = (max_count + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(gx, count), dim3(256), 0, s, dev, b);
    return (int)hipGetLastError();
}
