// The note-axis convolution of PitchedChannelsEncoder (style/model.py:46-53,78-84: Conv1d(50 -> 57, k = 14, s = 7, p = 4) over
// the 56 notes of every (position, fraction), output per octave) for BATCHED plans, forward and weight gradient, on the
// 128 x 128-class tile structure of csrc/audio.hip instead of the generic 64 x 64 accessor GEMM (csrc/gemm.hip), which spends
// 13-14 vector instructions of per-element im2col index math per MFMA on these two products (21 of a 64-clip pass's 46 GFLOP).
//
// As a GEMM:  out[(p, oct)][oc] = sum_{f < 10} sum_{r < 70} W'[f * 72 + r][oc] * x[p][f][35 oct - 20 + r]   (0 outside [0, 280)),
// i.e. for a fixed (position, fraction) the eight octaves read eight overlapping 70-float WINDOWS (stride 35) of one contiguous
// 280-float row — no division per element, a lane's 16-byte group never leaves its window.  The reduction index is padded from
// 70 to 72 per fraction (W' holds zeros there; the two pad elements read real neighbours of the window, times zero), so a k-tile
// of 36 is half a fraction and every group of four is aligned inside it.  W' (720 x 64, the Conv1d weight permuted to
// (fraction, tap, feature) x out-channel order, zero-padded) is rebuilt from the parameters once per forward (conv_prep_kernel).
//
//   forward        256 rows (32 positions x 8 octaves) x 64 columns per workgroup, four waves of 64 x 64 (2 x 2 f32 32x32x2 MFMA
//                  blocks: 72 MFMAs per wave and k-tile against 12 loads per lane), bias + leaky in the epilogue
//   weight grad    dW'[oc][k'] = sum over ALL rows of all clips of (dY o leaky'(Y))[row][oc] * window[row][k']: 64 x 256 tiles (four
//                  waves side by side), k-split over the rows into slabs written in PARAMETER layout (the deferred slab reduction of
//                  gemm.hip sums them in order); the bias gradient rides on pad column 70 of fraction 0 (its window value is 1)
// Summation order is fixed (k order inside a split, split order in the reduction): bitwise reproducible.  The forward is, like the
// accessor GEMM's, one exact-f32 fmaf chain per output over k = (fraction, tap, feature) ascending — the pad terms add exact
// zeros — so a clip's activations are bit-identical to the accessor path's (one-clip plans on the 64 x 64 tiling keep that path).
#include "mst_common.h"

#define CV_KP 72                 // padded taps x features per fraction
#define CV_KT 36                 // k-tile
#define CV_K (NF * CV_KP)        // 720
#define CV_ROW (NPN * NPF)       // 280 floats of one (position, fraction)
typedef mst_f32x16 cv_f32x16;
typedef float cv_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const MST_GLOBAL_AS float* cv_gp;

// W'[k' = f * 72 + r][oc] = W[oc][ic = f * 5 + r % 5][tap = r / 5]  (r < 70, oc < OC), else 0
__global__ __launch_bounds__(256) void conv_prep_kernel(const ConvDesc d, Bases b) {
    const float* w = b.p[SP_PAR] + d.w_off;
    float* wp = b.p[SP_TMP] + d.wp_off;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < CV_K * 64; e += gridDim.x * 256) {
        const int kp = e >> 6, oc = e & 63;
        const int f = kp / CV_KP, r = kp - f * CV_KP;
        float v = 0.f;
        if (r < CONV_K * NPF && oc < d.OC) {
            const int tap = r / NPF, feat = r - tap * NPF;
            v = w[(int64_t)oc * (NF * NPF * CONV_K) + (f * NPF + feat) * CONV_K + tap];
        }
        wp[e] = v;
    }
}

// ---------------------------------------------------------------------------------------------- forward
// WR = rows per wave: 64 (2 x 2 blocks per wave, 256-row tiles) or 32 (1 x 2 blocks, 128-row tiles).  The 128-row form has half the
// operand reuse but twice the workgroups at ~2/3 of the registers: its workgroups fill the chip's slots in finer rounds (PMC: at
// 64 clips the 1152 256-row workgroups ran a full round and a half-empty one on 768 slots).
template <int WR>
__global__ __launch_bounds__(256, WR == 64 ? 3 : 4) void conv_fwd_kernel(const ConvDesc d, Bases b) {
    constexpr int BM = 4 * WR, NQ = (BM * 9 + 255) / 256, PA = BM + 1, PB = 64 + 4;
    __shared__ float As[CV_KT][PA];                                    // k-major window tile, transposing stores (odd pitch)
    __shared__ __attribute__((aligned(16))) float Bs[CV_KT][PB];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wv = MST_UNIFORM(tid >> 6);
    const int64_t rows_total = (int64_t)d.clips * d.P * NOCT;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const cv_gp x = (cv_gp)b.p[SP_EXT0];
    const cv_gp wp = (cv_gp)(b.p[SP_TMP] + d.wp_off);
    // item = tid + 256 q (q < NQ) of the BM rows x 9 groups: nine consecutive lanes walk one row's 144 bytes
    unsigned xoff[NQ];            // element offset of the group at fraction 0, first half (window start may be negative: kept as int)
    int e0[NQ], lds_a[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int item = min(tid + 256 * q, BM * 9 - 1), row = item / 9, g = item - row * 9;      // (NQ * 256 > BM * 9 at 128 rows: the surplus lanes repeat the last item)
        const int64_t m = m0 + row;
        const int64_t mm = m < rows_total ? m : rows_total - 1;
        const int64_t p = mm >> 3;
        const int oct = (int)(mm & 7);
        e0[q] = 35 * oct - 20 + 4 * g;                                 // position of the group's first element inside the 280-float row
        xoff[q] = (unsigned)(p * (NF * CV_ROW));                       // (position, fraction 0) row start; plans keep clips x P x 2800 < 2^32
        lds_a[q] = (4 * g) * PA + row;
        if (m >= rows_total) e0[q] = -100000;                          // rows past the end: every element invalid
    }
    cv_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    cv_f4 ra[NQ], rb[3];
    auto issue = [&](const int kt) {
        const int f = kt >> 1, h = kt & 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // Branch-free: ONE 16-byte load per group whatever its position.  The window offsets are 35 oct - 20 + 4 n, so a group
            // is wholly in front of the row (octave 0: conv padding), wholly inside, wholly behind it, or — octave 7 only — starts
            // at element 277 and has exactly its last element outside: that one is loaded one element early and shifted.
            // (the shift / zeroing is applied when the k-tile is committed to LDS, so the load stays in flight under the MFMAs)
            const int e = e0[q] + CV_KT * h;                           // first element of the group inside the row
            const bool part = e == CV_ROW - 3, ok = e >= 0 && e <= CV_ROW - 3;
            const int es = ok ? (part ? e - 1 : e) : 0;
            ra[q] = *reinterpret_cast<const MST_GLOBAL_AS cv_f4*>(x + (xoff[q] + (unsigned)(f * CV_ROW) + (unsigned)es));
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int item = tid + 256 * q;                            // 36 x 16 groups of W'
            cv_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (item < CV_KT * 16) v = *reinterpret_cast<const MST_GLOBAL_AS cv_f4*>(wp + ((kt * CV_KT + (item >> 4)) * 64 + 4 * (item & 15)));
            rb[q] = v;
        }
    };
    issue(0);
    for (int kt = 0; kt < CV_K / CV_KT; ++kt) {
        float* af = &As[0][0];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = e0[q] + CV_KT * (kt & 1);
            const bool part = e == CV_ROW - 3, ok = e >= 0 && e <= CV_ROW - 3;
            const cv_f4 t = ra[q];
            cv_f4 v;
            v[0] = ok ? (part ? t[1] : t[0]) : 0.f; v[1] = ok ? (part ? t[2] : t[1]) : 0.f;
            v[2] = ok ? (part ? t[3] : t[2]) : 0.f; v[3] = (ok && !part) ? t[3] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) af[lds_a[q] + j * PA] = v[j];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int item = tid + 256 * q;
            if (item < CV_KT * 16) {
                float* dst = &Bs[item >> 4][4 * (item & 15)];
                dst[0] = rb[q][0]; dst[1] = rb[q][1]; dst[2] = rb[q][2]; dst[3] = rb[q][3];
            }
        }
        MST_LDS_BARRIER();
        if (kt + 1 < CV_K / CV_KT) issue(kt + 1);                      // flies under this k-tile's 72 MFMAs per wave
        mst_mfma_ktile_2x2<CV_KT, WR == 64, true>(As, Bs, wv * WR, 0, l31, kh, acc);
        MST_LDS_BARRIER();                                              // single LDS buffer: everyone is done reading before the next stores
    }
    // epilogue: x1[clip][p][oc * 8 + oct] = leaky(acc + bias[oc]).  A lane's 32 rows are 8 positions x 4 octaves: one division
    const float* bias = b.p[SP_PAR] + d.b_off;
    float* ws = b.p[SP_WS];
    const unsigned pbase = (unsigned)(m0 >> 3) + (unsigned)wv * (unsigned)(WR / 8);
    unsigned clip = pbase / (unsigned)d.P, pl = pbase - clip * (unsigned)d.P;
    float bn[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) bn[c] = (32 * c + l31) < d.OC ? bias[32 * c + l31] : 0.f;
#pragma unroll
    for (int t = 0; t < WR / 8; ++t) {                                       // position pbase + t: accumulator block a = t >> 2, registers 4 (t & 3) ..
        const int a = t >> 2;
        const int64_t pg = (int64_t)pbase + t;
        if (pg * NOCT < rows_total) {
            float* prow = ws + (int64_t)clip * d.clip_stride + d.x1_off + (int64_t)pl * (d.OC * NOCT);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * (t & 3) + u, oct = u + 4 * kh;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int n = 32 * c + l31;
                    if (n < d.OC) {
                        const float z = acc[a][c][r] + bn[c];
                        prow[n * NOCT + oct] = z > 0.f ? z : z * LEAKY;
                    }
                }
            }
        }
        if (++pl == (unsigned)d.P) { pl = 0; ++clip; }
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient
__global__ __launch_bounds__(256, 2) void conv_dw_kernel(const ConvDesc d, Bases b) {
    constexpr int KT = 32, PA = 64 + 1, PB = 256 + 4;
    __shared__ float As[KT][PA];                                       // (dY o leaky')^T: [row of the k-tile][out channel], transposing stores
    __shared__ __attribute__((aligned(16))) float Bs[KT][PB];          // windows: [row][k' column of the tile]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wv = MST_UNIFORM(tid >> 6);
    const int tn = blockIdx.x % 3, split = blockIdx.x / 3;             // 3 column tiles of 256 cover k' < 720
    const int64_t rows_total = (int64_t)d.clips * d.P * NOCT;
    const int64_t k0 = (int64_t)split * d.rows_per_split, k1 = min(rows_total, k0 + d.rows_per_split);
    const cv_gp x = (cv_gp)b.p[SP_EXT0];
    const cv_gp ws = (cv_gp)b.p[SP_WS], gr = (cv_gp)b.p[SP_GRAD];
    // B: lane -> column group cg = tid & 63 (fixed: its fraction and tap offset never change), rows (tid >> 6) + 4 q
    const int kp = tn * 256 + 4 * (tid & 63);                          // first k' of the lane's group
    const int bf = kp / CV_KP, br = kp - bf * CV_KP;                   // fraction, offset inside the padded 72
    const bool bcol = kp < CV_K;
    // A: items (position slot p4 < 4, out channel oc < OC, half) -> two 16-byte groups of 4 octaves
    cv_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    cv_f4 rg[2], ry[2], rb[8];
    const int na = 4 * d.OC * 2;                                       // A items per k-tile
    // the lane's two A items (position slot, out channel, octave half) never change
    int a_p4[2], a_col[2], a_lds[2];
    bool a_on[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int item = tid + 256 * q;
        a_on[q] = item < na;
        const int it = a_on[q] ? item : 0;
        const int p4 = it / (2 * d.OC), rem = it - p4 * (2 * d.OC), oc = rem >> 1, half = rem & 1;
        a_p4[q] = p4; a_col[q] = oc * NOCT + 4 * half; a_lds[q] = (p4 * 8 + 4 * half) * PA + oc;
    }
    // (clip, position inside the clip) of the k-tile's first position, advanced by four positions per k-tile: no division in the loop
    unsigned t_clip = (unsigned)((k0 >> 3) / d.P), t_pl = (unsigned)((k0 >> 3) - (int64_t)t_clip * d.P);
    auto issue = [&](const int64_t kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            cv_f4 g4 = {0.f, 0.f, 0.f, 0.f}, y4 = {1.f, 1.f, 1.f, 1.f};
            unsigned clip = t_clip, pl = t_pl + (unsigned)a_p4[q];
            while (pl >= (unsigned)d.P) { pl -= (unsigned)d.P; ++clip; }
            if (a_on[q] && ((kt >> 3) + a_p4[q]) * NOCT < k1) {          // (k0 and k1 are multiples of 8: whole positions)
                const int64_t o = (int64_t)clip * d.clip_stride + d.x1_off + (int64_t)pl * (d.OC * NOCT) + a_col[q];
                g4 = *reinterpret_cast<const MST_GLOBAL_AS cv_f4*>(gr + o);
                y4 = *reinterpret_cast<const MST_GLOBAL_AS cv_f4*>(ws + o);
            }
            rg[q] = g4; ry[q] = y4;
        }
        t_pl += 4;
        while (t_pl >= (unsigned)d.P) { t_pl -= (unsigned)d.P; ++t_clip; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = (tid >> 6) + 4 * q;                         // row of the k-tile: position slot row >> 3, octave row & 7
            const int64_t m = kt + row;
            // branch-free like the forward's: one 16-byte load per group; the two pad columns of a fraction (r = 70, 71) may hold
            // anything finite (their products are never stored) except k' = 70, the bias-gradient column, which is 1
            const bool live = bcol && m < k1;
            const int64_t mm = live ? m : k0;
            const unsigned pg = (unsigned)(mm >> 3);
            const int oct = (int)(mm & 7);
            const int e = 35 * oct - 20 + br;
            const bool part = e == CV_ROW - 3, ok = live && e >= 0 && e <= CV_ROW - 3;
            const int es = ok ? (part ? e - 1 : e) : 0;
            rb[q] = *reinterpret_cast<const MST_GLOBAL_AS cv_f4*>(x + (pg * (unsigned)(NF * CV_ROW) + (unsigned)(bf * CV_ROW) + (unsigned)es));
        }
    };
    for (int e = tid; e < KT * (64 - d.OC); e += 256) As[e / (64 - d.OC)][d.OC + e % (64 - d.OC)] = 0.f;      // unused out-channel columns: zero, once
    if (k0 < k1) issue(k0);
    for (int64_t kt = k0; kt < k1; kt += KT) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (a_on[q]) {
                float* af = &As[0][0] + a_lds[q];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j * PA] = rg[q][j] * (ry[q][j] > 0.f ? 1.f : LEAKY);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // the shift / zeroing of the group loaded by issue(kt), applied here so that the load flew under the previous MFMAs
            const int64_t m = kt + (tid >> 6) + 4 * q;
            const bool live = bcol && m < k1;
            const int e = 35 * (int)(m & 7) - 20 + br;
            const bool part = e == CV_ROW - 3, ok = live && e >= 0 && e <= CV_ROW - 3;
            const cv_f4 t = rb[q];
            cv_f4 v;
            v[0] = ok ? (part ? t[1] : t[0]) : 0.f; v[1] = ok ? (part ? t[2] : t[1]) : 0.f;
            v[2] = ok ? (part ? t[3] : t[2]) : 0.f; v[3] = (ok && !part) ? t[3] : 0.f;
            if (live && kp == CONV_K * NPF - 2) v[2] = 1.f;              // k' = 70 sits at element 2 of the group that starts at 68
            float* dst = &Bs[(tid >> 6) + 4 * q][4 * (tid & 63)];
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        MST_LDS_BARRIER();
        if (kt + KT < k1) issue(kt + KT);
        mst_mfma_ktile_2x2<KT>(As, Bs, 0, wv * 64, l31, kh, acc);
        MST_LDS_BARRIER();
    }
    // epilogue: this split's slab, in PARAMETER layout: weight (OC x 50 x 14) then bias (OC)
    float* slab = b.p[SP_TMP] + d.slab_off + (int64_t)split * d.slab_stride;
    const int wsize = d.OC * (NF * NPF * CONV_K);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int kq = tn * 256 + wv * 64 + 32 * c + l31;              // k' of this lane's column
        const int f = kq / CV_KP, r = kq - f * CV_KP;
        int col = -1;
        if (kq < CV_K && r < CONV_K * NPF) { const int tap = r / NPF, feat = r - tap * NPF; col = (f * NPF + feat) * CONV_K + tap; }
        const bool isb = kq == CONV_K * NPF;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int oc = 32 * a + (rr & 3) + 8 * (rr >> 2) + 4 * kh;
                if (oc >= d.OC) continue;
                if (col >= 0) slab[oc * (NF * NPF * CONV_K) + col] = acc[a][c][rr];
                else if (isb) slab[wsize + oc] = acc[a][c][rr];
            }
    }
}

int launch_conv_prep(const ConvDesc& d, Bases b, hipStream_t s) {
    hipLaunchKernelGGL(conv_prep_kernel, dim3((CV_K * 64 + 255) / 256), dim3(256), 0, s, d, b);
    return (int)hipGetLastError();
}
int launch_conv_fwd(const ConvDesc& d, Bases b, hipStream_t s) {
    const int64_t rows = (int64_t)d.clips * d.P * NOCT;
    // 128-row tiles while the 256-row grid would not fill four rounds of the chip's 768 co-resident slots
    if ((rows + 255) / 256 < 4 * 768) hipLaunchKernelGGL(conv_fwd_kernel<32>, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, s, d, b);
    else hipLaunchKernelGGL(conv_fwd_kernel<64>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d, b);
    return (int)hipGetLastError();
}
int launch_conv_dw(const ConvDesc& d, Bases b, hipStream_t s) {
    hipLaunchKernelGGL(conv_dw_kernel, dim3(3 * d.splits), dim3(256), 0, s, d, b);
    return (int)hipGetLastError();
}
