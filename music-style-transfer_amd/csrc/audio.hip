// AUDIO EXTENSION — NOT REFERENCE PARITY.  The reference (marcinp7/music-style-transfer) is a symbolic, piano-roll model and has
// no audio path at all (latex/music-style-transfer.tex:79-80 lists it as future work; requirements.txt has no audio library).
// BASELINE.json's metric text, however, speaks of "30 s @ 44.1 kHz, STFT(1024/256)" clips, a spectrogram featuriser and a
// feature-Gram style loss; SURVEY.md §8(f4) keeps that as an optional extension with a build-defined oracle.  This file is that
// extension, kept apart from the hot path (nothing in plan.hip or the model uses it, and bench.py reports it under its own key):
//
//   mst_audio_stft            Hann-windowed STFT (n_fft 1024 / 2048, centre-padded by reflection like torch.stft's defaults):
//                             two real frames ride one complex radix-4 Stockham FFT in LDS; complex and / or magnitude output
//   mst_audio_gram            feature Gram  G = S^T S / T  of a (frames x bins) magnitude matrix on the f32 matrix cores
//   mst_audio_style_iteration one optimisation iteration of the classic spectrogram style transfer: loss = || G(x) - G_style ||_F^2,
//                             dL/dx = (4 / T) x (G(x) - G_style), Adam on x — two MFMA GEMMs, a Gram finalise / loss kernel, Adam
//
// Oracle: oracle/audio_oracle.py (torch.stft / matmul / autograd / torch.optim.Adam on the CPU) — build-defined, PARITY UNPINNED.
//
// GEMM design (ag_gemm_kernel): 128 x 128 output tile per 256-lane workgroup, 2 x 2 waves of 64 x 64 (four 32x32x2 f32 MFMA
// accumulators per wave: 4 MFMAs per 4 LDS fragment reads), 32-deep k-tiles k-major in LDS, double-buffered with ONE LDS-only
// barrier per k-tile, 16-byte global loads along each operand's unit-stride direction, the next k-tile's loads in flight under the
// current tile's 64 MFMAs per wave.  Edge 32 x 32 blocks that lie outside the matrix are skipped (bins = 513 = 16 x 32 + 1).
#include <cmath>
#include <type_traits>
#include <vector>

#include "mst_common.h"

struct mst_audio_plan {
    int n_fft = 0, hop = 0, frames = 0, bins = 0, ld = 0, splits = 0, ntile = 0;
    int64_t n_samples = 0;
    float* d_win = nullptr;      // periodic Hann window (torch.hann_window(n_fft))
    float2* d_tw = nullptr;      // exp(-2 pi i m / n_fft), m < n_fft
    int* d_tiles = nullptr;      // lower-triangle tiles of the Gram: (ti, tj) pairs, ti >= tj
};

// ------------------------------------------------------------------------------------------ STFT
__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// One workgroup = TWO frames: z = a + i b of the two windowed real frames goes through one N-point complex FFT (radix-4 Stockham
// autosort passes in LDS, ping-pong buffers, one radix-2 pass first when N = 2 * 4^k), and the two spectra come apart by symmetry:
//   A[k] = (Z[k] + conj Z[N - k]) / 2,   B[k] = (Z[k] - conj Z[N - k]) / (2 i).
// Frame t covers samples [t hop - N / 2, t hop + N / 2) with reflection at both ends (torch.stft centre = True, pad_mode = reflect).
template <int N>
__global__ __launch_bounds__(N / 4) void stft_kernel(const float* __restrict__ audio, int64_t n, int hop, int frames,
                                                     const float* __restrict__ win, const float2* __restrict__ tw,
                                                     float2* __restrict__ spec, float* __restrict__ mag, int bins, int ld) {
    constexpr int NT = N / 4;
    __shared__ float2 buf[2][N];
    const int tid = threadIdx.x;
    const int fa = 2 * blockIdx.x, fb = fa + 1;
    const bool hb = fb < frames;
    for (int i = tid; i < N; i += NT) {
        int64_t ia = (int64_t)fa * hop - N / 2 + i, ib = (int64_t)fb * hop - N / 2 + i;
        ia = ia < 0 ? -ia : (ia >= n ? 2 * (n - 1) - ia : ia);
        ib = ib < 0 ? -ib : (ib >= n ? 2 * (n - 1) - ib : ib);
        const float w = win[i];
        buf[0][i] = make_float2(audio[ia] * w, hb ? audio[ib] * w : 0.f);
    }
    __syncthreads();
    int src = 0, p = 1;
    if ((N & 0x55555555) == 0) {                           // N = 2 * 4^k: one radix-2 pass (p = 1: no twiddles)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * NT;
            const float2 u0 = buf[src][i], u1 = buf[src][i + N / 2];
            buf[src ^ 1][2 * i] = cadd(u0, u1);
            buf[src ^ 1][2 * i + 1] = csub(u0, u1);
        }
        __syncthreads();
        src ^= 1; p = 2;
    }
    for (; p < N; p *= 4) {
        const int i = tid, k = i & (p - 1), j = ((i - k) << 2) + k;
        const int m = k * (N / (4 * p));
        const float2 u0 = buf[src][i];
        const float2 u1 = cmul(buf[src][i + NT], tw[m]);
        const float2 u2 = cmul(buf[src][i + 2 * NT], tw[2 * m]);
        const float2 u3 = cmul(buf[src][i + 3 * NT], tw[3 * m]);
        const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), d = csub(u1, u3);
        const float2 v3 = make_float2(d.y, -d.x);          // (u1 - u3) * (-i)
        buf[src ^ 1][j] = cadd(v0, v2);
        buf[src ^ 1][j + p] = cadd(v1, v3);
        buf[src ^ 1][j + 2 * p] = csub(v0, v2);
        buf[src ^ 1][j + 3 * p] = csub(v1, v3);
        __syncthreads();
        src ^= 1;
    }
    for (int k = tid; k < bins; k += NT) {
        const float2 z = buf[src][k], y = buf[src][(N - k) & (N - 1)];
        const float2 a = make_float2(0.5f * (z.x + y.x), 0.5f * (z.y - y.y));          // (Z[k] + conj Z[N-k]) / 2
        const float2 bb = make_float2(0.5f * (z.y + y.y), 0.5f * (y.x - z.x));         // (Z[k] - conj Z[N-k]) / (2 i)
        if (spec) {
            spec[(int64_t)fa * bins + k] = a;
            if (hb) spec[(int64_t)fb * bins + k] = bb;
        }
        if (mag) {
            mag[(int64_t)fa * ld + k] = sqrtf(a.x * a.x + a.y * a.y);
            if (hb) mag[(int64_t)fb * ld + k] = sqrtf(bb.x * bb.x + bb.y * bb.y);
        }
    }
    if (mag) {                                             // the pad columns [bins, ld) are part of the matrix: zeros
        for (int k = bins + tid; k < ld; k += NT) {
            mag[(int64_t)fa * ld + k] = 0.f;
            if (hb) mag[(int64_t)fb * ld + k] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------ GEMM
#define AG_BM 128
#define AG_BN 128
#define AG_KT 32
typedef float ag_f32x16 __attribute__((ext_vector_type(16)));
typedef float ag_f4 __attribute__((ext_vector_type(4), aligned(4)));
struct AgGemm {
    const float* A; const float* B; float* C;
    int M, N, K;                 // C is M x N, the reduction runs over K
    int Ma, Na, Ka;              // allocated extents of the unit-stride directions (multiples of 4, pads hold zeros)
    int64_t sA, sB;              // the non-unit stride of A and of B (floats)
    int ldc;
    int ksplit; int64_t slab_stride;      // ksplit > 1: split s writes its partial tile at C + s * slab_stride
    const int* tiles; int ntile;          // (tm, tn) pairs; nullptr: all tiles, row-major
};

// AKF / BKF = 1: the reduction index is the operand's unit-stride direction (A(m, k) = A[m * sA + k], B(k, n) = B[n * sB + k]);
//            0: the row / column index is (A(m, k) = A[k * sA + m], B(k, n) = B[k * sB + n]).
template <int AKF, int BKF>
__global__ __launch_bounds__(256, 2) void ag_gemm_kernel(const AgGemm g) {
    constexpr int PA = AG_BM + (AKF ? 1 : 4), PB = AG_BN + (BKF ? 1 : 4);     // k-major row pitch: odd for transposing stores, 16-byte for b128
    __shared__ __attribute__((aligned(16))) float As[2][AG_KT][PA];
    __shared__ __attribute__((aligned(16))) float Bs[2][AG_KT][PB];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    // the wave index as a SCALAR: everything derived from it (which 32 x 32 blocks are live) must be a uniform branch — as a
    // per-lane value the compiler predicated every MFMA separately and shuttled the accumulators through one register set
    const int wv = MST_UNIFORM(tid >> 6), wm = wv >> 1, wn = wv & 1;
    const int tiles_n = (g.N + AG_BN - 1) / AG_BN;
    const int tile = blockIdx.x % g.ntile, split = blockIdx.x / g.ntile;
    const int tm = g.tiles ? g.tiles[2 * tile] : tile / tiles_n, tn = g.tiles ? g.tiles[2 * tile + 1] : tile % tiles_n;
    const int m0 = tm * AG_BM, n0 = tn * AG_BN;
    int kchunk = (g.K + g.ksplit - 1) / g.ksplit;
    kchunk = (kchunk + AG_KT - 1) / AG_KT * AG_KT;
    const int k0 = split * kchunk, k1 = min(g.K, k0 + kchunk);
    ag_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    // which of the wave's four 32 x 32 blocks hold anything (wave-uniform)
    const bool lm[2] = {m0 + wm * 64 < g.M, m0 + wm * 64 + 32 < g.M}, ln[2] = {n0 + wn * 64 < g.N, n0 + wn * 64 + 32 < g.N};
    typedef const MST_GLOBAL_AS float* gp;
    const gp A = (gp)g.A, B = (gp)g.B;
    ag_f4 ra[4], rb[4];
    // KF = 0: lane -> (k = tid / 32 + 8 q, 4 consecutive rows from 4 (tid % 32));  KF = 1: lane -> (row = tid / 8 + 32 q, 4 consecutive k)
    auto issue = [&](const int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ag_f4 z = {0.f, 0.f, 0.f, 0.f};
            if (AKF) {
                const int m = m0 + (tid >> 3) + 32 * q, k = kt + 4 * (tid & 7);
                ra[q] = (m < g.M && k < g.Ka && k < k1) ? *reinterpret_cast<const MST_GLOBAL_AS ag_f4*>(A + ((int64_t)m * g.sA + k)) : z;
                if (k + 3 >= k1) {                          // a group that straddles the split's end keeps only its own k
#pragma unroll
                    for (int j = 0; j < 4; ++j) ra[q][j] = k + j < k1 ? ra[q][j] : 0.f;
                }
            } else {
                const int k = kt + (tid >> 5) + 8 * q, m = m0 + 4 * (tid & 31);
                ra[q] = (k < k1 && m < g.Ma) ? *reinterpret_cast<const MST_GLOBAL_AS ag_f4*>(A + ((int64_t)k * g.sA + m)) : z;
            }
            if (BKF) {
                const int n = n0 + (tid >> 3) + 32 * q, k = kt + 4 * (tid & 7);
                rb[q] = (n < g.N && k < g.Ka && k < k1) ? *reinterpret_cast<const MST_GLOBAL_AS ag_f4*>(B + ((int64_t)n * g.sB + k)) : z;
                if (k + 3 >= k1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rb[q][j] = k + j < k1 ? rb[q][j] : 0.f;
                }
            } else {
                const int k = kt + (tid >> 5) + 8 * q, n = n0 + 4 * (tid & 31);
                rb[q] = (k < k1 && n < g.Na) ? *reinterpret_cast<const MST_GLOBAL_AS ag_f4*>(B + ((int64_t)k * g.sB + n)) : z;
            }
        }
    };
    // The whole k-loop exists once per pattern of live 32 x 32 blocks of the wave (uniform: chosen once, outside the loop — inside
    // it the compiler merged the paths' accumulators through register copies at every k-tile).  Every path runs the same loads
    // and the same barriers; a wave without any live block only helps to stage the tiles.
    auto run = [&](auto LM0, auto LM1, auto LN1) {
        if (k0 < k1) issue(k0);
        int buf = 0;
        for (int kt = k0; kt < k1; kt += AG_KT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (AKF) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) As[buf][4 * (tid & 7) + j][(tid >> 3) + 32 * q] = ra[q][j];
                } else {
                    float* dst = &As[buf][(tid >> 5) + 8 * q][4 * (tid & 31)];
                    dst[0] = ra[q][0]; dst[1] = ra[q][1]; dst[2] = ra[q][2]; dst[3] = ra[q][3];
                }
                if (BKF) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) Bs[buf][4 * (tid & 7) + j][(tid >> 3) + 32 * q] = rb[q][j];
                } else {
                    float* dst = &Bs[buf][(tid >> 5) + 8 * q][4 * (tid & 31)];
                    dst[0] = rb[q][0]; dst[1] = rb[q][1]; dst[2] = rb[q][2]; dst[3] = rb[q][3];
                }
            }
            MST_LDS_BARRIER();                              // (a wave past this barrier has finished reading the other buffer)
            if (kt + AG_KT < k1) issue(kt + AG_KT);        // flies under this k-tile's MFMAs and across the next barrier
            if constexpr (decltype(LM0)::value)
                mst_mfma_ktile_2x2<AG_KT, decltype(LM1)::value, decltype(LN1)::value>(As[buf], Bs[buf], wm * 64, wn * 64, l31, kh, acc);
            buf ^= 1;
        }
    };
    typedef std::true_type Y; typedef std::false_type NO;
    if (!(lm[0] && ln[0])) run(NO{}, NO{}, NO{});
    else if (lm[1] && ln[1]) run(Y{}, Y{}, Y{});
    else if (lm[1]) run(Y{}, Y{}, NO{});
    else if (ln[1]) run(Y{}, NO{}, Y{});
    else run(Y{}, NO{}, NO{});
    float* C = g.C + (g.ksplit > 1 ? (int64_t)split * g.slab_stride : 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!(lm[a] && ln[c])) continue;
            const int n = n0 + wn * 64 + 32 * c + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < g.M && n < g.N) C[(int64_t)m * g.ldc + n] = acc[a][c][r];
            }
        }
}

// Gram finalise: G = (1 / T) sum over the k-splits' slabs (lower-triangle tiles, mirrored), D = cd (G - G_style) for the gradient
// GEMM, per-workgroup partial sums of (G - G_style)^2 (the loss), pad rows / columns of both matrices zero.  Splits and loss
// partials are summed in a fixed order.
__global__ __launch_bounds__(256) void gram_finalize_kernel(const float* __restrict__ slabs, int splits, int64_t slab_stride, int nb, int ld,
                                                            float inv_t, const float* __restrict__ gs, float cd,
                                                            float* __restrict__ G, float* __restrict__ D, float* __restrict__ loss_part) {
    __shared__ float red[4];
    float sq = 0.f;
    const int total = ld * ld;
    const int e = blockIdx.x * 256 + threadIdx.x;           // one element per lane: ceil(ld^2 / 256) workgroups
    if (e < total) {
        // lane (i, j) with i >= j reads the stored (lower) element — consecutive lanes, consecutive addresses — and writes both
        // (i, j) and its mirror (j, i); lanes above the diagonal only clear the pads they own.  (Letting every lane read "its"
        // element made half the lanes walk a column of every slab.)
        const int i = e / ld, j = e - i * ld;
        if (i >= nb || j >= nb) {
            if (G) G[e] = 0.f;
            if (D) D[e] = 0.f;
        } else if (i >= j) {
            const MST_GLOBAL_AS float* src = (const MST_GLOBAL_AS float*)slabs + e;
            float gv = 0.f, dv = 0.f;
            int s = 0;
            for (; s + 8 <= splits; s += 8) {              // eight slab loads in flight, added in split order
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(s + q) * slab_stride];
#pragma unroll
                for (int q = 0; q < 8; ++q) gv += v[q];
            }
            for (; s < splits; ++s) gv += src[(int64_t)s * slab_stride];
            gv *= inv_t;
            if (gs) { const float df = gv - gs[e]; dv = cd * df; sq = i == j ? df * df : 2.f * df * df; }     // G_style is symmetric
            const int et = j * ld + i;
            if (G) { G[e] = gv; G[et] = gv; }
            if (D) { D[e] = dv; D[et] = dv; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0 && loss_part) loss_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void loss_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ loss) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------ C ABI

extern "C" mst_audio_plan* mst_audio_plan_create(int32_t n_fft, int32_t hop, int64_t n_samples, int32_t* status) {
    int32_t dummy; if (!status) status = &dummy;
    if ((n_fft != 1024 && n_fft != 2048) || hop < 1 || hop > n_fft || n_samples <= n_fft / 2 || n_samples / hop > (1 << 22)) { *status = MST_ERR_ARG; return nullptr; }
    mst_audio_plan* p = new mst_audio_plan();
    p->n_fft = n_fft; p->hop = hop; p->n_samples = n_samples;
    p->frames = 1 + (int)(n_samples / hop);
    p->bins = n_fft / 2 + 1;
    p->ld = (p->bins + 7) / 8 * 8;
    const int nt = (p->bins + AG_BM - 1) / AG_BM;
    std::vector<int> tiles;
    for (int i = 0; i < nt; ++i) for (int j = 0; j <= i; ++j) { tiles.push_back(i); tiles.push_back(j); }
    p->ntile = (int)tiles.size() / 2;
    // k-splits of the Gram: the FULL lower-triangle tiles (both tile edges inside the matrix: bins = 513 leaves a fifth tile row of
    // one live bin, whose workgroups are light) times the splits should fill the chip's 256 CUs once, with >= 4 k-tiles per split
    const int nfull_edge = p->bins / AG_BM, nfull = nfull_edge * (nfull_edge + 1) / 2;
    int s = nfull > 0 ? 256 / nfull : 256;
    const int maxs = (p->frames + 4 * AG_KT - 1) / (4 * AG_KT);
    p->splits = s < 1 ? 1 : (s > maxs ? maxs : s);
    {   // a split's share is rounded to whole k-tiles inside the kernel: no split may come out empty (its slab would stay unwritten)
        const int chunk = ((p->frames + p->splits - 1) / p->splits + AG_KT - 1) / AG_KT * AG_KT;
        p->splits = (p->frames + chunk - 1) / chunk;
    }
    std::vector<float> win(n_fft);
    std::vector<float2> tw(n_fft);
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < n_fft; ++i) {
        win[i] = (float)(0.5 - 0.5 * std::cos(two_pi * i / n_fft));
        tw[i].x = (float)std::cos(two_pi * i / n_fft); tw[i].y = (float)(-std::sin(two_pi * i / n_fft));
    }
    bool ok = hipMalloc((void**)&p->d_win, n_fft * sizeof(float)) == hipSuccess && hipMalloc((void**)&p->d_tw, n_fft * sizeof(float2)) == hipSuccess &&
              hipMalloc((void**)&p->d_tiles, tiles.size() * sizeof(int)) == hipSuccess;
    ok = ok && hipMemcpy(p->d_win, win.data(), n_fft * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_tw, tw.data(), n_fft * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_tiles, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { mst_audio_plan_destroy(p); *status = MST_ERR_ALLOC; return nullptr; }
    *status = MST_OK;
    return p;
}

extern "C" void mst_audio_plan_destroy(mst_audio_plan* p) {
    if (!p) return;
    hipFree(p->d_win); hipFree(p->d_tw); hipFree(p->d_tiles);
    delete p;
}

extern "C" int32_t mst_audio_plan_info(const mst_audio_plan* p, int64_t out[6]) {
    if (!p || !out) return MST_ERR_ARG;
    out[0] = p->frames; out[1] = p->bins; out[2] = p->ld;
    // workspace floats of mst_audio_gram / mst_audio_style_iteration: the Gram's k-split slabs, D, the loss partials
    out[3] = (int64_t)p->splits * p->ld * p->ld + (int64_t)p->ld * p->ld + ((int64_t)p->ld * p->ld + 255) / 256 + 64;
    out[4] = p->splits; out[5] = p->ntile;
    return MST_OK;
}

extern "C" int32_t mst_audio_stft(const mst_audio_plan* p, const float* audio, float* spec, float* mag, mst_stream stream) {
    if (!p || !audio || (!spec && !mag)) return MST_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((p->frames + 1) / 2);
    if (p->n_fft == 1024)
        hipLaunchKernelGGL(stft_kernel<1024>, grid, dim3(256), 0, s, audio, p->n_samples, p->hop, p->frames, (const float*)p->d_win,
                           (const float2*)p->d_tw, reinterpret_cast<float2*>(spec), mag, p->bins, p->ld);
    else
        hipLaunchKernelGGL(stft_kernel<2048>, grid, dim3(512), 0, s, audio, p->n_samples, p->hop, p->frames, (const float*)p->d_win,
                           (const float2*)p->d_tw, reinterpret_cast<float2*>(spec), mag, p->bins, p->ld);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

// slabs of G's lower-triangle tiles: feat is (frames x ld) row-major, A(m = i, k = t) = feat[t * ld + i], B(k = t, n = j) = feat[t * ld + j]
static int gram_slabs(const mst_audio_plan* p, const float* feat, float* ws, hipStream_t s) {
    AgGemm g{};
    g.A = feat; g.B = feat; g.C = ws; g.M = p->bins; g.N = p->bins; g.K = p->frames; g.Ma = p->ld; g.Na = p->ld; g.Ka = p->frames;
    g.sA = p->ld; g.sB = p->ld; g.ldc = p->ld; g.ksplit = p->splits; g.slab_stride = (int64_t)p->ld * p->ld;
    g.tiles = p->d_tiles; g.ntile = p->ntile;
    hipLaunchKernelGGL((ag_gemm_kernel<0, 0>), dim3(p->ntile * p->splits), dim3(256), 0, s, g);
    return (int)hipGetLastError();
}

extern "C" int32_t mst_audio_gram(const mst_audio_plan* p, const float* feat, float* gram, float* ws, mst_stream stream) {
    if (!p || !feat || !gram || !ws) return MST_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (gram_slabs(p, feat, ws, s)) return MST_ERR_LAUNCH;
    hipLaunchKernelGGL(gram_finalize_kernel, dim3((p->ld * p->ld + 255) / 256), dim3(256), 0, s, (const float*)ws, p->splits, (int64_t)p->ld * p->ld,
                       p->bins, p->ld, 1.f / (float)p->frames, (const float*)nullptr, 0.f, gram, (float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

extern "C" int32_t mst_audio_style_iteration(const mst_audio_plan* p, float* x, const float* gram_style, float* grad, float* exp_avg,
                                             float* exp_avg_sq, float* state, float* ws, float* loss, double lr, mst_stream stream) {
    if (!p || !x || !gram_style || !grad || !exp_avg || !exp_avg_sq || !state || !ws || !loss) return MST_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t ll = (int64_t)p->ld * p->ld;
    float* D = ws + (int64_t)p->splits * ll;
    float* part = D + ll;
    if (gram_slabs(p, x, ws, s)) return MST_ERR_LAUNCH;
    const int fin_blocks = (int)((ll + 255) / 256);
    hipLaunchKernelGGL(gram_finalize_kernel, dim3(fin_blocks), dim3(256), 0, s, (const float*)ws, p->splits, ll, p->bins, p->ld,
                       1.f / (float)p->frames, gram_style, 4.f / (float)p->frames, (float*)nullptr, D, part);
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)part, fin_blocks, loss);
    // grad = x D:  A(m = t, k = f) = x[t * ld + f] (k unit stride), B(k = f, n = j) = D[f * ld + j]
    AgGemm g{};
    g.A = x; g.B = D; g.C = grad; g.M = p->frames; g.N = p->ld; g.K = p->ld; g.Ma = p->frames; g.Na = p->ld; g.Ka = p->ld;
    g.sA = p->ld; g.sB = p->ld; g.ldc = p->ld; g.ksplit = 1; g.slab_stride = 0; g.tiles = nullptr;
    g.ntile = ((g.M + AG_BM - 1) / AG_BM) * ((g.N + AG_BN - 1) / AG_BN);
    hipLaunchKernelGGL((ag_gemm_kernel<1, 0>), dim3(g.ntile), dim3(256), 0, s, g);
    if (hipGetLastError() != hipSuccess) return MST_ERR_LAUNCH;
    return mst_adam_step(x, grad, exp_avg, exp_avg_sq, (int64_t)p->frames * p->ld, state, lr, .9, .999, 1e-8, 1 << 30, 1.0, 0, stream);
}
