// Note-level fused stages for gfx950.
//
// MelodyEncoder tail (style/model.py:270-296) and PitchedStyleApplier tail (:644-675) share one
// shape: an "octave (+) scale-degree" outer sum  h[o*7+d] = leaky(oct[o] + deg[d])  per note,
// concatenated with a small per-note vector and pushed through a tiny Linear.  The reference
// materialises the (positions x 56 x 50) concat in memory (358 MB at the training cap); here
// every note is one lane, the octave/degree rows of the position sit in LDS, the tiny weights
// are LDS-broadcast, and nothing but the final 8 (or 5) outputs per note touches HBM.
//
// Backward kernels recompute the cheap per-note activations instead of saving them.  The melody
// encoder's reductions over notes run on the matrix cores (three 16x16x4 f32 MFMA chains per sweep),
// the applier's backward puts the hidden feature on the lane so that every reduction over notes is
// a per-lane register accumulation; weight gradients leave one slab row per wave.  No float
// atomics; summation order is fixed.
#include "mst_common.h"

__device__ __forceinline__ float lrelu(float z) { return z > 0.f ? z : z * LEAKY; }
__device__ __forceinline__ float dlrelu(float y) { return y > 0.f ? 1.f : LEAKY; }

// ============================================================================ MelodyEncoder
// Per position p = (channel c, q = (bar, beat)) and note item (fraction f, note n = (octave o, degree dg)):
//   cat = [ leaky(oct[p][o][:] + deg[p][dg][:]) (W) | leaky(channels_linear(x[p][f][n][:5])) (CW) ],  x_c = leaky(linear(cat)) (W)
// and the channels are merged by combine() (style/model.py:296,796-815):
//   n_c = sqrt(1 + sum x_c^2),  melody = sum_c x_c n_c / S,  S = sum_c n_c.
// The per-channel tensor x_c (4.6 MB per clip at the bench shape, written once and re-read four times by the unfused
// path) is never materialised: it is cheap to recompute (15 -> 8 Linear per note), so
//   forward  = me_sumsq (partial sums of x_c^2 per channel)  +  me_notes_fwd (recompute x_c for every channel of a q, write melody)
//   backward = me_bwd_reduce (partial a_c = sum g x_c, b = sum g melody)  +  me_notes_bwd (recompute, dx_c on the fly, gradients).
// Work distribution: ONE WAVE PER POSITION, lane = note (56 of 64 lanes), loop over the 10 fractions; octave / degree rows
// of the position sit in registers (o, dg are fixed per lane), the tiny weights in LDS (hoisted to registers by the compiler).
// Backward reductions over notes run on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32): per sweep the wave stages its
// per-note vectors transposed in a wave-private LDS region ([feature][note], row stride 66 => conflict-free fragment reads) and
//   dW_linear | db_linear = gm^T [cat | 1],   dW_channels | db_channels = gc^T [x | 1],   d_oct | d_deg = onehot(o | dg)^T god
// are three 16-step MFMA chains; accumulators persist in registers across sweeps / positions.  Partial sums meet in a
// fixed order everywhere (per-lane serial, xor-butterfly, MFMA k order): results are bitwise reproducible, no float atomics.
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
#define NT_ROW 66                  // LDS row stride of the transposed staging: bank = 2*(lane&15) + (lane>>4) => conflict-free

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int N>
__device__ __forceinline__ void ld_vec(const float* p, float (&v)[N]) {
    static_assert(N % 4 == 0, "16-byte rows");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { const float4 t = reinterpret_cast<const float4*>(p)[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
}
template <int N>
__device__ __forceinline__ void st_vec(float* p, const float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// the small weights of the melody encoder's note tail: wave-uniform, read through constant-address-space pointers, i.e. by
// scalar loads into SGPRs (they are parameters: nothing writes them while a pass runs).  Staged in LDS and hoisted into
// vector registers by the compiler, as in round 2, they took ~170 VGPRs of every lane and held these kernels at two waves per SIMD.
template <int W, int CW>
struct MeWeights {
    static constexpr int KL = W + CW;
    typedef const MST_CONST_AS float* cptr;
    cptr wc, bc, wl, bl;
    __device__ __forceinline__ MeWeights(const NotesDesc& d, const float* par)
        : wc((cptr)(par + d.wc_off)), bc((cptr)(par + d.bc_off)), wl((cptr)(par + d.wl_off)), bl((cptr)(par + d.bl_off)) {}
};

// One note of one channel: cat (W + CW) and x_c (W).  The octave / degree half of cat — and with it the first W terms of the
// Linear's dot products — does not depend on the fraction f, so a position computes it ONCE (me_pos) and every fraction only
// continues the same fmaf chains with its CW note-feature terms (me_frac): bit-identical to evaluating the chain in one go,
// at 97 instead of 155 multiply-adds per note and fraction.
template <int W, int CW>
__device__ __forceinline__ void me_pos(const MeWeights<W, CW>& w, const float (&octv)[W], const float (&degv)[W], float (&cat_od)[W], float (&pre)[W]) {
    constexpr int KL = W + CW;
#pragma unroll
    for (int j = 0; j < W; ++j) cat_od[j] = lrelu(octv[j] + degv[j]);
#pragma unroll
    for (int j = 0; j < W; ++j) {
        float z = w.bl[j];
#pragma unroll
        for (int i = 0; i < W; ++i) z = fmaf(w.wl[j * KL + i], cat_od[i], z);
        pre[j] = z;
    }
}
template <int W, int CW>
__device__ __forceinline__ void me_frac(const MeWeights<W, CW>& w, const float (&pre)[W], const float (&x5)[NPF], float (&catx)[CW], float (&out)[W]) {
    constexpr int KL = W + CW;
#pragma unroll
    for (int k = 0; k < CW; ++k) {
        float z = w.bc[k];
#pragma unroll
        for (int i = 0; i < NPF; ++i) z = fmaf(w.wc[k * NPF + i], x5[i], z);
        catx[k] = lrelu(z);
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
        float z = pre[j];
#pragma unroll
        for (int k = 0; k < CW; ++k) z = fmaf(w.wl[j * KL + W + k], catx[k], z);
        out[j] = lrelu(z);
    }
}

__device__ __forceinline__ void ld_x5(const float* x, int64_t item, float (&x5)[NPF]) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) x5[i] = x[item * NPF + i];
}

// Reduction pass shared by forward and backward.  BWD = false: partial sums of x_c^2.  BWD = true: partial a_c = sum g x_c
// and (channel-0 waves only) partial b = sum g melody.  Wave g of the grid serves channel g / nwc and the positions
// q = g % nwc, + nwc, ...; lane 0 leaves one partial per wave.
// (Fetching the next sweep's operands one sweep ahead was measured on MI355X and lost: the weights already occupy ~170
// VGPRs, the extra live registers cost the second wave per SIMD, and these kernels then ran 1.6-2x slower.)
template <int W, int CW, bool BWD>
__global__ __launch_bounds__(256) void me_reduce_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    const MeWeights<W, CW> wt(d, b.p[SP_PAR]);
    const int tid = threadIdx.x;
    const int lane = tid & 63, g = blockIdx.x * 4 + (tid >> 6);
    if (g >= d.C * d.nwc) return;                          // wave-uniform
    const int c = g / d.nwc, i0 = g - c * d.nwc;
    const bool valid = lane < NPN;
    const int n = valid ? lane : NPN - 1, o = n / NDEG, dg = n - o * NDEG;
    const float* x = b.p[d.x_space] + d.x_off;
    const float* ws = b.p[SP_WS];
    float acc = 0.f, accb = 0.f;
    for (int q = i0; q < d.Q; q += d.nwc) {
        const int64_t p = (int64_t)c * d.Q + q;
        float octv[W], degv[W], cat_od[W], pre[W];
        ld_vec<W>(ws + d.oct_off + (p * NOCT + o) * W, octv);
        ld_vec<W>(ws + d.deg_off + (p * NDEG + dg) * W, degv);
        me_pos<W, CW>(wt, octv, degv, cat_od, pre);
        for (int f = 0; f < NF; ++f) {
            float x5[NPF], catx[CW], out[W];
            ld_x5(x, (p * NF + f) * NPN + n, x5);
            me_frac<W, CW>(wt, pre, x5, catx, out);
            if constexpr (!BWD) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < W; ++j) a = fmaf(out[j], out[j], a);
                if (valid) acc += a;
            } else {
                const int64_t mi = ((int64_t)q * NF + f) * NPN + n;
                float gv[W];
                ld_vec<W>(b.p[SP_GRAD] + d.g_out_off + mi * W, gv);
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < W; ++j) a = fmaf(gv[j], out[j], a);
                if (valid) acc += a;
                if (c == 0) {
                    float mel[W];
                    ld_vec<W>(ws + d.out_off + mi * W, mel);
                    float bb = 0.f;
#pragma unroll
                    for (int j = 0; j < W; ++j) bb = fmaf(gv[j], mel[j], bb);
                    if (valid) accb += bb;
                }
            }
        }
    }
    acc = wave_sum64(acc);
    if (lane == 0) b.p[SP_TMP][d.part_off + (int64_t)c * d.nwc + i0] = acc;
    if (BWD && c == 0) {
        accb = wave_sum64(accb);
        if (lane == 0) b.p[SP_TMP][d.part_off + (int64_t)d.C * d.nwc + i0] = accb;
    }
}

// melody[q] = sum_c x_c[q] n_c / S: one wave per (q, half of the fractions), all channels recomputed; the first wave
// also leaves n_c, S for the backward
#define ME_FH 2
#ifndef ME_FWD_MINW
#define ME_FWD_MINW 4
#endif
#ifndef ME_BWD_MINW
#define ME_BWD_MINW 3
#endif
template <int W, int CW>
__global__ __launch_bounds__(256, ME_FWD_MINW) void me_notes_fwd_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    const MeWeights<W, CW> wt(d, b.p[SP_PAR]);
    __shared__ float nc_s[4][COMBINE_MAXC + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, gw = blockIdx.x * 4 + wv;
    const int fhn = d.fhn;                                 // waves per q: 2 in batched plans, 10 (one fraction each) for one clip
    const int q = gw / fhn, fh = gw - q * fhn;
    if (q >= d.Q) return;                                   // wave-uniform
    float* tmp = b.p[SP_TMP];
    float S = 0.f;
    for (int c = 0; c < d.C; ++c) {                        // every wave re-sums the partials in the same order
        const float v = wave_sum64(lane < d.nwc ? tmp[d.part_off + (int64_t)c * d.nwc + lane] : 0.f);
        const float nc = sqrtf(1.f + v);
        S += nc;
        if (lane == 0) { nc_s[wv][c] = nc; if (gw == 0) tmp[d.stats_off + c] = nc; }
    }
    if (gw == 0 && lane == 0) tmp[d.stats_off + d.C] = S;
    MST_WAVE_SYNC();
    const bool valid = lane < NPN;
    const int n = valid ? lane : NPN - 1, o = n / NDEG, dg = n - o * NDEG;
    const float* x = b.p[d.x_space] + d.x_off;
    float* ws = b.p[SP_WS];
    const int F0 = NF / fhn;
    const int f_begin = fh * F0, f_end = fh == fhn - 1 ? NF : f_begin + F0;
    // fractions outside, channels inside (one accumulator row live): the octave / degree half is recomputed per (fraction, channel)
    // here — keeping it across the half's five fractions needs five accumulator rows and costs two of the four waves per SIMD
    for (int f = f_begin; f < f_end; ++f) {
        float acc[W];
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] = 0.f;
        for (int c = 0; c < d.C; ++c) {
            const int64_t p = (int64_t)c * d.Q + q;
            float octv[W], degv[W], cat_od[W], pre[W], x5[NPF], catx[CW], out[W];
            ld_vec<W>(ws + d.oct_off + (p * NOCT + o) * W, octv);
            ld_vec<W>(ws + d.deg_off + (p * NDEG + dg) * W, degv);
            ld_x5(x, (p * NF + f) * NPN + n, x5);
            me_pos<W, CW>(wt, octv, degv, cat_od, pre);
            me_frac<W, CW>(wt, pre, x5, catx, out);
            const float nc = nc_s[wv][c];
#pragma unroll
            for (int j = 0; j < W; ++j) acc[j] = fmaf(out[j], nc, acc[j]);
        }
        const float rS = 1.f / S;
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] = acc[j] * rS;
        if (valid) st_vec<W>(ws + d.out_off + (((int64_t)q * NF + f) * NPN + n) * W, acc);
    }
}

// One wave per (position, half of the fractions): the two waves of a position are neighbours in one workgroup and meet
// once, at the end of the position, to add their octave / degree partial sums (fixed order: half 0 + half 1).
template <int W, int CW>
__global__ __launch_bounds__(256, ME_BWD_MINW) void me_notes_bwd_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    constexpr int KL = W + CW;
    constexpr int R_WC = 0, R_BC = CW * NPF, R_WL = R_BC + CW, R_BL = R_WL + W * KL;
    static_assert(KL + 1 <= 16 && W <= 16 && CW <= 16, "one 16x16 MFMA tile per product");
    // transposed staging rows of one wave: gm (W) | cat (KL) | gc (CW) | x (5) | god (W)
    constexpr int T_GM = 0, T_CAT = W, T_GC = W + KL, T_X = T_GC + CW, T_GOD = T_X + NPF, T_ROWS = T_GOD + W;
    const MeWeights<W, CW> wt(d, b.p[SP_PAR]);
    __shared__ float tr[4][T_ROWS][NT_ROW];
    __shared__ float od_s[2][4][64];                       // octave | degree partial sums of the odd wave of each pair
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, pair = wv >> 1, fh = wv & 1;
    const bool valid = lane < NPN;
    const int n = valid ? lane : NPN - 1, o = n / NDEG, dg = n - o * NDEG;
    const float* x = b.p[d.x_space] + d.x_off;
    const float* ws = b.p[SP_WS];
    const float* tmp = b.p[SP_TMP];
    float* gr = b.p[SP_GRAD];
    float (*t)[NT_ROW] = tr[wv];
    const int r = lane & 15, kh = lane >> 4;               // MFMA fragment coordinates: row / column r, k = 4 s + kh
    // one-hot A operand of the octave | degree reduction: item 4 s + kh is note (o', dg'); rows 0..7 octaves, 8..14 degrees
    // (kept as a 16-bit mask: sixteen float registers would cost the second wave per SIMD)
    unsigned ohmask = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int it = 4 * s + kh, io = it / NDEG, idg = it - io * NDEG;
        if (it < NPN && (r < NOCT ? io == r : idg == r - NOCT)) ohmask |= 1u << s;
    }
    const float S = tmp[d.stats_off + d.C];
    const float bsum = wave_sum64(lane < d.nwc ? tmp[d.part_off + (int64_t)d.C * d.nwc + lane] : 0.f);
    nt_f32x4 accW = {0.f, 0.f, 0.f, 0.f}, accC = {0.f, 0.f, 0.f, 0.f};
    const int P = d.C * d.Q;
    constexpr int F0 = NF / 2;
    const int f_begin = fh * F0, f_end = fh ? NF : F0;
    // the loop is workgroup-uniform (both waves of a pair, and both pairs, run the same trip count) because of the
    // barrier at the end of a position; a pair beyond P computes nothing
    const int trips = (P + gridDim.x * 2 - 1) / (gridDim.x * 2);
    for (int tIdx = 0; tIdx < trips; ++tIdx) {
        const int p = (tIdx * gridDim.x + blockIdx.x) * 2 + pair;
        const bool live = p < P;
        nt_f32x4 accO = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const int c = p / d.Q, q = p - c * d.Q;
            const float nc = tmp[d.stats_off + c];
            const float ac = wave_sum64(lane < d.nwc ? tmp[d.part_off + (int64_t)c * d.nwc + lane] : 0.f);
            // two divisions per position instead of two per element: v_div_scale / v_rcp / 4 fma / div_fmas / div_fixup sequences
            // were a seventh of the sweep's instructions
            const float ncS = nc / S, k2n = ((ac - bsum) / S) / nc;
            float octv[W], degv[W], cat_od[W], pre[W], gmsum[W];
            ld_vec<W>(ws + d.oct_off + ((int64_t)p * NOCT + o) * W, octv);
            ld_vec<W>(ws + d.deg_off + ((int64_t)p * NDEG + dg) * W, degv);
            me_pos<W, CW>(wt, octv, degv, cat_od, pre);
#pragma unroll
            for (int j = 0; j < W; ++j) gmsum[j] = 0.f;
            // ---- per fraction: the note-feature half.  dW_linear[:, W:] | db_linear = gm^T [catx | 1],  dW_channels | db = gc^T [x | 1]
            for (int f = f_begin; f < f_end; ++f) {
                float x5[NPF], catx[CW], out[W], gv[W];
                ld_x5(x, ((int64_t)p * NF + f) * NPN + n, x5);
                ld_vec<W>(gr + d.g_out_off + (((int64_t)q * NF + f) * NPN + n) * W, gv);
                me_frac<W, CW>(wt, pre, x5, catx, out);
                float gm[W];
#pragma unroll
                for (int j = 0; j < W; ++j) {              // combine backward, then the linear's leaky
                    const float dx = fmaf(gv[j], ncS, k2n * out[j]);
                    gm[j] = valid ? dx * dlrelu(out[j]) : 0.f;
                    gmsum[j] += gm[j];
                }
                MST_WAVE_SYNC();                           // the previous sweep's fragment reads are done
#pragma unroll
                for (int j = 0; j < W; ++j) t[T_GM + j][lane] = gm[j];
#pragma unroll
                for (int k = 0; k < CW; ++k) {
                    float gsum = 0.f;
#pragma unroll
                    for (int j = 0; j < W; ++j) gsum = fmaf(gm[j], wt.wl[j * KL + W + k], gsum);
                    t[T_CAT + W + k][lane] = catx[k];
                    t[T_GC + k][lane] = gsum * dlrelu(catx[k]);
                }
#pragma unroll
                for (int i = 0; i < NPF; ++i) t[T_X + i][lane] = x5[i];
                MST_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int k = 4 * s + kh;
                    const float a1 = r < W ? t[T_GM + (r < W ? r : 0)][k] : 0.f;
                    // columns 0 .. W-1 (the octave / degree half of cat) are added once per position, below
                    const float b1 = (r >= W && r < KL) ? t[T_CAT + ((r >= W && r < KL) ? r : W)][k] : (r == KL ? 1.f : 0.f);
                    accW = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, accW, 0, 0, 0);
                    const float a2 = r < CW ? t[T_GC + (r < CW ? r : 0)][k] : 0.f;
                    const float b2 = r < NPF ? t[T_X + (r < NPF ? r : 0)][k] : (r == NPF ? 1.f : 0.f);
                    accC = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, accC, 0, 0, 0);
                    if ((s & 3) == 3) MST_SCHED_FENCE();
                }
            }
            // ---- once per position (half): the octave / degree half, from the gradients summed over the fractions:
            //   dW_linear[:, :W] += gmsum^T cat_od,   d_oct | d_deg = onehot(o | dg)^T (gmsum W_l[:, :W] o leaky'(cat_od))
            MST_WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < W; ++j) t[T_GM + j][lane] = gmsum[j];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                float gsum = 0.f;
#pragma unroll
                for (int j = 0; j < W; ++j) gsum = fmaf(gmsum[j], wt.wl[j * KL + i], gsum);
                t[T_CAT + i][lane] = cat_od[i];
                t[T_GOD + i][lane] = gsum * dlrelu(cat_od[i]);
            }
            MST_WAVE_SYNC();
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = 4 * s + kh;
                const float a1 = r < W ? t[T_GM + (r < W ? r : 0)][k] : 0.f;
                const float b1 = r < W ? t[T_CAT + (r < W ? r : 0)][k] : 0.f;
                accW = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, accW, 0, 0, 0);
                const float b3 = r < W ? t[T_GOD + (r < W ? r : 0)][k] : 0.f;
                accO = __builtin_amdgcn_mfma_f32_16x16x4f32((float)((ohmask >> s) & 1u), b3, accO, 0, 0, 0);
                if ((s & 3) == 3) MST_SCHED_FENCE();
            }
        }
        if (fh) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) od_s[pair][rr][lane] = accO[rr];
        }
        __syncthreads();
        // D[row = 4 kh + rr][col = r]: rows 0..7 -> d_oct[o][j = r], rows 8..14 -> d_deg[dg][j = r]; sole writer of these rows
        if (live && !fh && r < W) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = 4 * kh + rr;
                const float v = accO[rr] + od_s[pair][rr][lane];
                if (row < NOCT) gr[d.g_oct_off + ((int64_t)p * NOCT + row) * W + r] = v;
                else if (row < NOCT + NDEG) gr[d.g_deg_off + ((int64_t)p * NDEG + row - NOCT) * W + r] = v;
            }
        }
        __syncthreads();
    }
    // one slab row per workgroup: channels_linear.{weight,bias}, linear.{weight,bias} in parameter order; the four waves'
    // partial sums meet in wave order
    constexpr int NWT = R_BL + W;
    static_assert(NWT <= 256, "slab row wider than the workgroup");
    float* wsum = &tr[0][0][0];                            // the staging rows are free now: 4 x NWT floats
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * kh + rr;
        if (row < W) {
            if (r < KL) wsum[wv * NWT + R_WL + row * KL + r] = accW[rr]; else if (r == KL) wsum[wv * NWT + R_BL + row] = accW[rr];
        }
        if (row < CW) {
            if (r < NPF) wsum[wv * NWT + R_WC + row * NPF + r] = accC[rr]; else if (r == NPF) wsum[wv * NWT + R_BC + row] = accC[rr];
        }
    }
    __syncthreads();
    if (tid < NWT)
        b.p[SP_TMP][d.slab_off + (int64_t)blockIdx.x * d.slab_stride + tid] = (wsum[tid] + wsum[NWT + tid]) + (wsum[2 * NWT + tid] + wsum[3 * NWT + tid]);
}

// ============================================================================ PitchedStyleApplier
#define PSA_HW 30
// (melody_linear — 8 -> 20 + leaky over (positions x 56) rows, style/model.py:606-610,660-662 — is applied HERE, per note, from the
// melody row: the (positions x 56 x 20) tensor it produced in round 2 cost a kernel, a 2.9 MB write and two reads per clip)
template <int ML, int MEL>
__global__ __launch_bounds__(64) void psa_notes_fwd_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    constexpr int KL = PSA_HW + ML;
    // the output Linear's weights TRANSPOSED, [input][output padded to 8]: the five outputs of one input sit side by side, so the
    // packed FMAs take (w[j][i], w[j][i + 1]) straight from one 16-byte broadcast read (row-major, the compiler spent 1.5 v_mov per
    // v_pk_fma_f32 on pairing elements of two weight rows)
    __shared__ __attribute__((aligned(16))) float w_s[KL][8];
    __shared__ float b_s[NPF];
    __shared__ float wm_s[ML * MEL], bm_s[ML];
    __shared__ float lo_s[NOCT * PSA_HW], ld_s[NDEG * PSA_HW];
    const int tid = threadIdx.x;
    const float* par = b.p[SP_PAR];
    for (int i = tid; i < NPF * KL; i += 64) w_s[i % KL][i / KL] = par[d.wl_off + i];
    for (int i = tid; i < ML * MEL; i += 64) wm_s[i] = par[d.wm_off + i];
    if (tid < ML) bm_s[tid] = par[d.wm_off + ML * MEL + tid];
    if (tid < NPF) b_s[tid] = par[d.bl_off + tid];
    float* ws = b.p[SP_WS];
    const int QF = d.Q * NF;
    const int n = tid, o = n / NDEG, dg = n - o * NDEG;
    // octave | degree rows: leaky(rt[qf] + it[c]) (the Linear over the broadcast-concat, decomposed: plan.hip).  The rt row of
    // the qf and the it row of the NEXT channel sit in registers while the current (qf, c) is processed
    constexpr int NLO = NOCT * PSA_HW, NOD = (NOCT + NDEG) * PSA_HW, NPRE = (NOD + 63) / 64;
    float rtv[NPRE], od[NPRE];
    auto fetch_rt = [&](int64_t q_) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int i = tid + 64 * q;
            rtv[q] = i < NLO ? ws[d.rt_oct_off + q_ * NLO + i] : (i < NOD ? ws[d.rt_deg_off + q_ * (NOD - NLO) + (i - NLO)] : 0.f);
        }
    };
    auto fetch_od = [&](int c_) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int i = tid + 64 * q;
            od[q] = i < NLO ? ws[d.it_oct_off + (int64_t)c_ * NLO + i] : (i < NOD ? ws[d.it_deg_off + (int64_t)c_ * (NOD - NLO) + (i - NLO)] : 0.f);
        }
    };
    fetch_od(0);
    for (int qf = blockIdx.x; qf < QF; qf += gridDim.x) {
        __syncthreads();
        fetch_rt(qf);
        float zm[NPF];
#pragma unroll
        for (int i = 0; i < NPF; ++i) zm[i] = 0.f;
        if (n < NPN) {
            float mel[MEL];
            ld_vec<MEL>(ws + d.mel_off + ((int64_t)qf * NPN + n) * MEL, mel);
#pragma unroll
            for (int i = 0; i < NPF; ++i) zm[i] = b_s[i];
#pragma unroll
            for (int k = 0; k < ML; ++k) {
                float a = bm_s[k];
#pragma unroll
                for (int q = 0; q < MEL; ++q) a = fmaf(wm_s[k * MEL + q], mel[q], a);
                const float mlk = lrelu(a);
#pragma unroll
                for (int i = 0; i < NPF; ++i) zm[i] = fmaf(w_s[PSA_HW + k][i], mlk, zm[i]);
            }
        }
        for (int c = 0; c < d.C; ++c) {
            const int64_t row = (int64_t)c * QF + qf;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NPRE; ++q) {
                const int i = tid + 64 * q;
                const float zv = lrelu(rtv[q] + od[q]);
                if (i < NLO) lo_s[i] = zv; else if (i < NOD) ld_s[i - NLO] = zv;
            }
            __syncthreads();
            fetch_od(c + 1 < d.C ? c + 1 : 0);
            if (n < NPN) {
                float z[NPF];
#pragma unroll
                for (int i = 0; i < NPF; ++i) z[i] = zm[i];
#pragma unroll
                for (int j = 0; j < PSA_HW; ++j) {
                    const float h = lrelu(lo_s[o * PSA_HW + j] + ld_s[dg * PSA_HW + j]);
#pragma unroll
                    for (int i = 0; i < NPF; ++i) z[i] = fmaf(w_s[j][i], h, z[i]);
                }
                float* out = ws + d.out_off + (row * NPN + n) * NPF;
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    const float s = MST_FAST_RCP(1.f + MST_FAST_EXP(-z[i]));      // v_exp / v_rcp: ~2 ulp, against ~40 instructions
                    out[i] = i == 0 ? 6.f * s : s;
                }
            }
        }
    }
}

// Backward of the applier's note tail.  Per row (channel c, qf) and note n = (o, dg):
//   h[j] = leaky(lo[o][j] + ld[dg][j]) (30),  z[i] = b[i] + sum_j W[i][j] h[j] + sum_k W[i][30 + k] ml[qf][n][k],  y = act(z) (5).
// ONE WORKGROUP PER qf, ONE WAVE PER CHANNEL PAIR (lanes 0..29: channel 2 w, lanes 32..61: channel 2 w + 1, lane = hidden feature
// j), the workgroup walking its qf's in lockstep.  With the feature on the lane every reduction over notes is a per-lane register
// accumulation: the lane keeps ld[0..6][j] and W[0..4][j] in registers, the pair's 2 x 56 x 5 output gradients dz sit in a
// wave-private LDS region and come back as wave-uniform 16-byte reads, and for every note
//   dh = sum_i dz[i] W[i][j];  dW[i][j] += dz[i] h;  g = dh leaky'(h);  d_lo[o][j] += g;  d_ld[dg][j] += g.
// What the kernel no longer round-trips through HBM (round 2 wrote 18 MB per clip from here and three more kernels re-read it):
//   * dL/dz per row (the (positions x 450) gradients of the octave / degree pre-activations) existed only to be summed: over the
//     channels (= gradient of rt[qf]: summed across the workgroup's waves through LDS, written once per qf) and over qf
//     (= gradient of it[c]: the wave keeps 15 running sums per lane for ITS channels across all its qf's and leaves one partial row
//     per workgroup; a column sum over the workgroups finishes it).  The per-row tensors and their segment reduce are gone.
//   * LOSS = true (the backward of mst_train_iteration): dL/dy is not read either — it is a function of the prediction, the
//     target (the borrowed pitched input) and six scalars of the loss tail (loss_bwd_kernel's formula, same operations), so the
//     elementwise loss backward over the pitched tensor and its 2.9 MB per clip of gradient never exist.
//   * the rt / it rows are fetched once per qf / once per workgroup instead of once per row.
// The melody-linear columns need only the CHANNEL SUM of dz (ml does not depend on the channel): after the waves met,
//   g_ml[n][k] = sum_i dzsum[n][i] W[i][30 + k],  dW[i][30 + k] += dzsum[n][i] ml[n][k],  db[i] += dzsum[n][i]
// with lane = (k, note group), the groups dealt over all waves — and melody_linear itself (ml = leaky(Wm mel + bm), 8 -> 20,
// style/model.py:606-610,660-662) is differentiated right here: the lane recomputes ml[n][k] from the staged melody row, takes
// gpre = g_ml leaky'(ml), accumulates dWm[k][:] += gpre mel[n][:] and dbm[k] += gpre, and after a wave-private exchange of gpre
// the wave's lanes = (note, m) form the melody gradient g_mel[n][m] = sum_k gpre[n][k] Wm[k][m].  Neither ml nor its gradient
// exists in memory any more (round 2: a 2.9 MB tensor per clip written once and read twice, its gradient written and read
// once, and the two row-wise Linear launches around them).  Two workgroup barriers per qf; every sum has a fixed order.
#ifndef PSA_BWD_MINW
#define PSA_BWD_MINW 2
#endif
template <int ML, int MEL, int NPB, bool LOSS>
__global__ __launch_bounds__(64 * NPB, PSA_BWD_MINW) void psa_bwd2_kernel(const NotesDesc* __restrict__ dp, Bases b) {
    const NotesDesc d = dp[blockIdx.y];
    constexpr int KL = PSA_HW + ML;
    constexpr int NLO = NOCT * PSA_HW, NLD = NDEG * PSA_HW, NOD = NOCT + NDEG;
    constexpr int ROWE = NPN * NPF;                       // 280 output elements per row
    constexpr int NSLOT = (2 * ROWE + 63) / 64;            // flat (row A | row B) elements per lane
    constexpr int NG = 64 / ML;                            // note groups per wave in the melody phase
    constexpr int NPMIN = NPB == 1 ? 1 : NPB / 2 + 1;      // fewest waves a workgroup of this bucket has
    constexpr int MAXN = (NPN + NPMIN * NG - 1) / (NPMIN * NG);      // notes per melody-phase lane, at most
    __shared__ __attribute__((aligned(16))) float raw_s[NPB][2][2 * ROWE];      // [wave][y | t or dy][row A | row B], flat; later the wave's dL/dz
    __shared__ __attribute__((aligned(16))) float dz_s[NPB][2][NPN][8];         // [wave][row A | B][note][5 used of 8]
    __shared__ float lo_s[NPB][NOCT][64];                                       // the pair's octave rows, [octave][lane]
    __shared__ __attribute__((aligned(16))) float dzs_s[NPN][8];                // dz summed over the channels of this qf
    __shared__ __attribute__((aligned(16))) float mel_s[2][NPN][MEL];           // the qf's melody rows; two buffers: a wave may stage
                                                                                // the next qf while another still reads this one
    __shared__ float gp_s[NPN][ML + 1];                                         // gpre[n][k]; a wave reads back only the notes it wrote
    __shared__ float coef_s[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NP = blockDim.x >> 6, nthreads = blockDim.x;
    const int half = lane >> 5, jl = lane & 31;
    const bool jvalid = jl < PSA_HW;
    const int j = jvalid ? jl : PSA_HW - 1;
    const int c = 2 * wv + half;
    const bool rvalid = c < d.C;
    const int cc = rvalid ? c : d.C - 1;
    const float* par = b.p[SP_PAR];
    const float* ws = b.p[SP_WS];
    float* gr = b.p[SP_GRAD];
    const float* tgt = LOSS ? b.p[d.x_space] + d.x_off : gr + d.g_out_off;       // target rows (LOSS) or upstream gradient rows
    const int QF = d.Q * NF;
    float (*rto)[64] = reinterpret_cast<float (*)[64]>(&raw_s[wv][0][0]);       // dL/dz of the wave's pair, [octave | degree][lane]
    float wj[NPF], dwj[NPF];                               // W[i][j] and its gradient
#pragma unroll
    for (int i = 0; i < NPF; ++i) { wj[i] = par[d.wl_off + i * KL + j]; dwj[i] = 0.f; }
    float itv[NOD], acc_it[NOD];                           // it[c] rows of this lane's channel; running sums of dL/dz over qf
#pragma unroll
    for (int q = 0; q < NOD; ++q) {
        itv[q] = q < NOCT ? ws[d.it_oct_off + (int64_t)cc * NLO + q * PSA_HW + j] : ws[d.it_deg_off + (int64_t)cc * NLD + (q - NOCT) * PSA_HW + j];
        acc_it[q] = 0.f;
    }
    // melody phase: lane = (k, note group g3); the NP * NG groups of the workgroup deal the 56 notes
    const int k = lane % ML, g3 = lane / ML;
    const bool mact = g3 < NG;
    const int G = wv * NG + g3, NGT = NP * NG;
    float wm[NPF], dwm[NPF], dbm[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) { wm[i] = par[d.wl_off + i * KL + PSA_HW + k]; dwm[i] = 0.f; dbm[i] = 0.f; }
    float wml[MEL], dwml[MEL], dbml = 0.f;                 // melody_linear row k: Wm[k][:], its gradient, d bm[k]
#pragma unroll
    for (int q = 0; q < MEL; ++q) { wml[q] = par[d.wm_off + k * MEL + q]; dwml[q] = 0.f; }
    const float bml = par[d.wm_off + ML * MEL + k];
    if (LOSS) {
        // d total / d (TP FP FN SEvel SEdur BCE) of the pitched tensor from the loss tail's saved Jacobian (loss_bwd_kernel)
        if (tid < 8) {
            const float* saved = ws + d.loss_saved_off;
            const float* gl = ws + d.loss_gl_off;
            float gk[MST_N_LOSSES], sk[MST_N_LOSSES];
#pragma unroll
            for (int q = 0; q < MST_N_LOSSES; ++q) { gk[q] = gl[q]; sk[q] = saved[q * 16 + tid]; }
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < MST_N_LOSSES; ++q) a += gk[q] != 0.f ? gk[q] * sk[q] : 0.f;
            coef_s[tid] = a;
        }
        __syncthreads();
    }
    const float cTP = LOSS ? coef_s[0] : 0.f, cFP = LOSS ? coef_s[1] : 0.f, cFN = LOSS ? coef_s[2] : 0.f;
    const float cSEV = LOSS ? coef_s[3] : 0.f, cSED = LOSS ? coef_s[4] : 0.f, cBCE = LOSS ? coef_s[5] : 0.f;
    // flat element e = lane + 64 q of the pair's (row A | row B): which row, offset inside the row (recomputed where used: nine
    // index registers and their flags cost a wave per SIMD)
    // what a qf needs from global memory — the rt rows, the pair's predictions and targets — fetched behind the PREVIOUS qf's
    // note loop (its registers are free then), so the loads fly under that qf's reductions, barriers and melody phase
    // Every global access below is `uniform base pointer (SGPR pair) [32-bit lane offset]`: with 64-bit per-lane addresses the
    // compiler kept two address registers per outstanding load and spilled ~300 bytes per lane at three waves per SIMD.
    typedef const MST_GLOBAL_AS float* gptr_t;
    const gptr_t y0 = (gptr_t)(ws + d.out_off), t0 = (gptr_t)tgt;
    const gptr_t rto0 = (gptr_t)(ws + d.rt_oct_off), rtd0 = (gptr_t)(ws + d.rt_deg_off), mel0 = (gptr_t)(ws + d.mel_off);
    typedef float mel_f4 __attribute__((ext_vector_type(4)));
    static_assert(MEL % 4 == 0 && NPN * MEL / 4 <= 64 * NPB * 2, "melody rows staged with at most two 16-byte loads per lane");
    struct QfIn { float rt[NOD], y[NSLOT], t[NSLOT]; };
    // flat element e = lane + 64 q of the pair's (row A | row B): q < 4 lies in row A, q > 4 in row B, q = 4 straddles
    const unsigned rowA = (unsigned)((2 * wv < d.C ? 2 * wv : 0) * QF) * ROWE + lane;
    const unsigned rowB = (unsigned)((2 * wv + 1 < d.C ? 2 * wv + 1 : 0) * QF) * ROWE + lane;
    auto fetch = [&](int qf_, QfIn& r) {
        const unsigned oo = (unsigned)qf_ * NLO + j, od = (unsigned)qf_ * NLD + j;
#pragma unroll
        for (int q = 0; q < NOD; ++q) r.rt[q] = q < NOCT ? rto0[oo + q * PSA_HW] : rtd0[od + (q - NOCT) * PSA_HW];
        const unsigned a = rowA + (unsigned)qf_ * ROWE, bb = rowB + (unsigned)qf_ * ROWE;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            const int e0 = 64 * q;                          // compile-time part of the element index
            unsigned pos;
            if (e0 + 63 < ROWE) pos = a + e0;
            else if (e0 >= ROWE) pos = (e0 + 63 < 2 * ROWE || lane + e0 < 2 * ROWE) ? bb + (e0 - ROWE) : bb;
            else pos = lane + e0 < ROWE ? a + e0 : bb + (e0 - ROWE);
            r.y[q] = y0[pos];
            r.t[q] = t0[pos];
        }
    };
    int qf = blockIdx.x;
    bool have = qf < QF;                                   // workgroup-uniform
    QfIn cur;
    if (have) fetch(qf, cur);
    int par_q = 0;
    while (have) {
        const int nqf = qf + gridDim.x;
        const bool nhave = nqf < QF;
        // ---- dz = dL/dy act'(y) of both rows (zeros for an absent row B): rows staged flat, then one lane per (row, note)
#pragma unroll
        for (int q = 0; q < NSLOT; ++q)
            if (lane + 64 * q < 2 * ROWE) { raw_s[wv][0][lane + 64 * q] = cur.y[q]; raw_s[wv][1][lane + 64 * q] = cur.t[q]; }
        MST_WAVE_SYNC();
        float dzv[2][NPF];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = lane + 64 * u;
            const bool act = it < 2 * NPN;
            const int r = it >= NPN ? 1 : 0, n = act ? it - r * NPN : 0;
            const float* yy = &raw_s[wv][0][r * ROWE + n * NPF];
            const float* tt = &raw_s[wv][1][r * ROWE + n * NPF];
            const bool rowok = act && (2 * wv + r) < d.C;
            float y5[NPF], g5[NPF];
#pragma unroll
            for (int i = 0; i < NPF; ++i) { y5[i] = yy[i]; g5[i] = tt[i]; }
            if (LOSS) {                                    // g5 holds the target: same operations as loss_bwd_kernel
                const float p0 = y5[0], pv = y5[1], t0 = g5[0], tv = g5[1];
                const float m = tv > 0.f ? 1.f : 0.f;
                float gv = cTP * (pv < tv ? 1.f : (pv == tv ? 0.5f : 0.f));
                gv += cFP * (pv - tv > 0.f ? 1.f : 0.f) - cFN * (tv - pv > 0.f ? 1.f : 0.f);
                gv -= cSEV * 2.f * (tv - pv) * m;
                g5[0] = cSED * 2.f * (p0 - fminf(t0, 6.f)) * (1.f / 36.f) * m;
                g5[1] = gv;
#pragma unroll
                for (int a = 2; a < NPF; ++a) g5[a] = cBCE * m * (y5[a] - g5[a]) / fmaxf((1.f - y5[a]) * y5[a], 1e-12f);
            }
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const float y = y5[i];
                dzv[u][i] = rowok ? g5[i] * (i == 0 ? y * (1.f - y * (1.f / 6.f)) : y * (1.f - y)) : 0.f;
            }
        }
        // (the staged rows are dead from here on: their storage becomes the wave's dL/dz slots)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = lane + 64 * u;
            if (it < 2 * NPN) {
                const int r = it >= NPN ? 1 : 0, n = it - r * NPN;
#pragma unroll
                for (int i = 0; i < NPF; ++i) dz_s[wv][r][n][i] = dzv[u][i];
            }
        }
        float ldv[NDEG], dld[NDEG];
#pragma unroll
        for (int o = 0; o < NOCT; ++o) lo_s[wv][o][lane] = lrelu(cur.rt[o] + itv[o]);
#pragma unroll
        for (int q = 0; q < NDEG; ++q) { ldv[q] = lrelu(cur.rt[NOCT + q] + itv[NOCT + q]); dld[q] = 0.f; }
        MST_WAVE_SYNC();
        const float* dzr = &dz_s[wv][half][0][0];
        // the octave loop is a REAL loop (a fully unrolled 56-note body made hipcc hoist every independent add / select to
        // the top and spill 500 registers); the degree rows stay in registers, the octave rows come from LDS
#pragma unroll 1
        for (int o = 0; o < NOCT; ++o) {
            const float lo = lo_s[wv][o][lane];
            float dlo = 0.f;
#pragma unroll
            for (int dg = 0; dg < NDEG; ++dg) {
                const float* zr = dzr + (o * NDEG + dg) * 8;
                const float4 z4 = *reinterpret_cast<const float4*>(zr);
                const float z5 = zr[4];
                const float xh = lo + ldv[dg];
                const float slope = xh > 0.f ? 1.f : LEAKY;
                const float h = xh * slope;
                float dh = z4.x * wj[0];
                dh = fmaf(z4.y, wj[1], dh); dh = fmaf(z4.z, wj[2], dh); dh = fmaf(z4.w, wj[3], dh); dh = fmaf(z5, wj[4], dh);
                dwj[0] = fmaf(z4.x, h, dwj[0]); dwj[1] = fmaf(z4.y, h, dwj[1]); dwj[2] = fmaf(z4.z, h, dwj[2]);
                dwj[3] = fmaf(z4.w, h, dwj[3]); dwj[4] = fmaf(z5, h, dwj[4]);
                const float g = dh * slope;
                dlo += g; dld[dg] += g;
            }
            // gradient of the PRE-activation (leaky' from the activation's sign)
            const float dzo = rvalid ? dlo * dlrelu(lo) : 0.f;
            rto[o][lane] = dzo;
        }
#pragma unroll
        for (int o = 0; o < NOCT; ++o) acc_it[o] += rto[o][lane];                // (lane-private slots, static register indices)
        // the qf's melody rows (56 x MEL, contiguous): 16-byte pieces dealt over the workgroup, in LDS before the first barrier
        {
            constexpr int NPIECE = NPN * MEL / 4;
            for (int e = tid; e < NPIECE; e += nthreads) {
                const mel_f4 v = *reinterpret_cast<const MST_GLOBAL_AS mel_f4*>(mel0 + ((unsigned)qf * NPN * MEL + 4 * e));
                float* dst = &mel_s[par_q][0][0] + 4 * e;
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
        }
        if (nhave) fetch(nqf, cur);                        // the next qf's operands: in flight from here to the top of the loop
#pragma unroll
        for (int q = 0; q < NDEG; ++q) {
            const float dzd = rvalid ? dld[q] * dlrelu(ldv[q]) : 0.f;
            rto[NOCT + q][lane] = dzd;
            acc_it[NOCT + q] += dzd;
        }
        __syncthreads();                                   // every wave's dz and dL/dz of this qf are in LDS
        // ---- g_rt[qf] = sum over the channels of dL/dz (waves in order, even channel before odd): sole writer of this row
        for (int w = tid; w < NLO + NLD; w += nthreads) {
            const int q = w / PSA_HW, jj = w - q * PSA_HW;
            float a = 0.f;
            for (int u = 0; u < NP; ++u) {
                const float* ru = &raw_s[u][0][0] + q * 64 + jj;
                a += ru[0] + ru[32];
            }
            if (w < NLO) gr[d.rt_oct_off + (int64_t)qf * NLO + w] = a;
            else gr[d.rt_deg_off + (int64_t)qf * NLD + (w - NLO)] = a;
        }
        for (int e = tid; e < ROWE; e += nthreads) {
            const int n = e / NPF, i = e - n * NPF;
            float a = 0.f;
            for (int u = 0; u < NP; ++u) a += dz_s[u][0][n][i] + dz_s[u][1][n][i];
            dzs_s[n][i] = a;
        }
        __syncthreads();
        // ---- the melody-linear columns from the channel sums, and melody_linear's own backward
        if (mact) {
#pragma unroll
            for (int u = 0; u < MAXN; ++u) {
                const int n = G + u * NGT;
                if (n < NPN) {
                    const float4 s4 = *reinterpret_cast<const float4*>(&dzs_s[n][0]);
                    const float s5 = dzs_s[n][4];
                    float gm = s4.x * wm[0];
                    gm = fmaf(s4.y, wm[1], gm); gm = fmaf(s4.z, wm[2], gm); gm = fmaf(s4.w, wm[3], gm); gm = fmaf(s5, wm[4], gm);
                    float mel[MEL];
                    ld_vec<MEL>(&mel_s[par_q][n][0], mel);
                    float a = bml;
#pragma unroll
                    for (int q = 0; q < MEL; ++q) a = fmaf(wml[q], mel[q], a);      // same chain as the forward kernel
                    const float mv = lrelu(a);
                    dwm[0] = fmaf(s4.x, mv, dwm[0]); dwm[1] = fmaf(s4.y, mv, dwm[1]); dwm[2] = fmaf(s4.z, mv, dwm[2]);
                    dwm[3] = fmaf(s4.w, mv, dwm[3]); dwm[4] = fmaf(s5, mv, dwm[4]);
                    dbm[0] += s4.x; dbm[1] += s4.y; dbm[2] += s4.z; dbm[3] += s4.w; dbm[4] += s5;
                    const float gpre = gm * dlrelu(mv);
#pragma unroll
                    for (int q = 0; q < MEL; ++q) dwml[q] = fmaf(gpre, mel[q], dwml[q]);
                    dbml += gpre;
                    gp_s[n][k] = gpre;
                }
            }
        }
        MST_WAVE_SYNC();
        // g_mel[n][m] = sum_k gpre[n][k] Wm[k][m] for the notes of THIS wave's groups (n = wv * NG + g + u * NGT): lane = (note, m)
        {
            const float* wmf = par + d.wm_off;
            constexpr int NITEM = MAXN * NG * MEL;
            for (int e = lane; e < NITEM; e += 64) {
                const int m = e % MEL, t = e / MEL;        // t = (u, g)
                const int u = t / NG, g = t - u * NG;
                const int n = wv * NG + g + u * NGT;
                if (n < NPN) {
                    float a = 0.f;
#pragma unroll
                    for (int kk = 0; kk < ML; ++kk) a = fmaf(gp_s[n][kk], wmf[kk * MEL + m], a);
                    gr[d.g_mel_off + ((int64_t)qf * NPN + n) * MEL + m] = a;       // sole writer: the applier is melody's only consumer
                }
            }
        }
        qf = nqf; have = nhave; par_q ^= 1;
        // (no barrier here: the next qf's wave-private writes follow its readers by the two barriers above; dzs_s is rewritten
        // only behind the next qf's first barrier)
    }
    // ---- gradient of it[c]: this workgroup's partial sums over its qf's, one row per workgroup (a column sum finishes them)
    if (rvalid && jvalid) {
        float* po = gr + d.itp_oct_off + ((int64_t)blockIdx.x * d.C + c) * NLO + j;
        float* pd = gr + d.itp_deg_off + ((int64_t)blockIdx.x * d.C + c) * NLD + j;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) po[o * PSA_HW] = acc_it[o];
#pragma unroll
        for (int q = 0; q < NDEG; ++q) pd[q * PSA_HW] = acc_it[NOCT + q];
    }
    // ---- one slab row per workgroup, in parameter order: melody_linear.weight (ML x MEL), melody_linear.bias (ML), linear.weight
    // (5 x KL), linear.bias (5); partial sums meet in a fixed order
    constexpr int NML = ML * MEL + ML, NWT = NML + NPF * KL + NPF;
    static_assert(NWT <= 2 * NPN * 8 && (MEL + 1) * 64 <= 4 * ROWE, "slab row wider than the staging it reuses");
    __syncthreads();                                       // all LDS staging is free now
    float* wsum = &dz_s[wv][0][0][0];                      // NWT floats per wave
    float (*red)[64] = reinterpret_cast<float (*)[64]>(&raw_s[wv][0][0]);
#pragma unroll
    for (int i = 0; i < NPF; ++i) red[i][lane] = dwj[i];
    MST_WAVE_SYNC();
    if (lane < PSA_HW) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) wsum[NML + i * KL + lane] = red[i][lane] + red[i][lane + 32];
    }
    MST_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < NPF; ++i) red[i][lane] = dwm[i];
    MST_WAVE_SYNC();
    if (lane < ML) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            float a = red[i][lane];
#pragma unroll
            for (int gq = 1; gq < NG; ++gq) a += red[i][lane + gq * ML];
            wsum[NML + i * KL + PSA_HW + lane] = a;
        }
    }
    MST_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < NPF; ++i) red[i][lane] = dbm[i];
    MST_WAVE_SYNC();
    if (lane < NPF) {
        float a = red[lane][0];
#pragma unroll
        for (int gq = 1; gq < NG; ++gq) a += red[lane][gq * ML];
        wsum[NML + NPF * KL + lane] = a;
    }
    MST_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < MEL; ++q) red[q][lane] = mact ? dwml[q] : 0.f;
    red[MEL][lane] = mact ? dbml : 0.f;
    MST_WAVE_SYNC();
    if (lane < ML) {                                       // lane = k: the note groups' partial sums in group order
#pragma unroll
        for (int q = 0; q <= MEL; ++q) {
            float a = red[q][lane];
#pragma unroll
            for (int gq = 1; gq < NG; ++gq) a += red[q][lane + gq * ML];
            if (q < MEL) wsum[lane * MEL + q] = a; else wsum[ML * MEL + lane] = a;
        }
    }
    __syncthreads();
    for (int w = tid; w < NWT; w += nthreads) {
        float a = 0.f;
        for (int u = 0; u < NP; ++u) a += (&dz_s[u][0][0][0])[w];
        b.p[SP_TMP][d.slab_off + (int64_t)blockIdx.x * d.slab_stride + w] = a;
    }
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
    if (h.ML == 20 && h.W == 8) hipLaunchKernelGGL((KERN<20, 8>), GRID, BLOCK, 0, s, dev, b);         \
    else if (h.ML == 14 && h.W == 4) hipLaunchKernelGGL((KERN<14, 4>), GRID, BLOCK, 0, s, dev, b);    \
    else return MST_ERR_UNSUPPORTED;

#define ME_RED_DISPATCH(BWD, GRID, BLOCK)                                                              \
    if (h.W == 8 && h.CW == 7) hipLaunchKernelGGL((me_reduce_kernel<8, 7, BWD>), GRID, BLOCK, 0, s, dev, b);     \
    else if (h.W == 4 && h.CW == 5) hipLaunchKernelGGL((me_reduce_kernel<4, 5, BWD>), GRID, BLOCK, 0, s, dev, b); \
    else return MST_ERR_UNSUPPORTED;
int launch_me_sumsq(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    ME_RED_DISPATCH(false, dim3((h.C * h.nwc + 3) / 4, count), dim3(256));
    return (int)hipGetLastError();
}
int launch_me_bwd_reduce(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    ME_RED_DISPATCH(true, dim3((h.C * h.nwc + 3) / 4, count), dim3(256));
    return (int)hipGetLastError();
}
int launch_me_notes_fwd(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    ME_DISPATCH(me_notes_fwd_kernel, dim3((h.Q * h.fhn + 3) / 4, count), dim3(256));
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
int psa_bwd_waves(int C) { return (C + 1) / 2; }      // one wave per channel pair
#define PSA_BWD_LAUNCH(ML_, MEL_, NPB_)                                                                                         \
    {                                                                                                                           \
        if (b.flags & MST_BF_LOSS_FUSED) hipLaunchKernelGGL((psa_bwd2_kernel<ML_, MEL_, NPB_, true>), grid, dim3(64 * np), 0, s, dev, b);   \
        else hipLaunchKernelGGL((psa_bwd2_kernel<ML_, MEL_, NPB_, false>), grid, dim3(64 * np), 0, s, dev, b);                    \
    }
#define PSA_BWD_BUCKETS(ML_, MEL_)                                                                  \
    if (np <= 1) PSA_BWD_LAUNCH(ML_, MEL_, 1) else if (np <= 2) PSA_BWD_LAUNCH(ML_, MEL_, 2) else if (np <= 4) PSA_BWD_LAUNCH(ML_, MEL_, 4)  \
    else if (np <= 8) PSA_BWD_LAUNCH(ML_, MEL_, 8) else PSA_BWD_LAUNCH(ML_, MEL_, 12)
int launch_psa_notes_bwd(const NotesDesc* dev, const NotesDesc& h, int count, Bases b, hipStream_t s) {
    const int np = psa_bwd_waves(h.C);
    if (np > 12) return MST_ERR_UNSUPPORTED;         // 24 pitched channels (a MIDI file has 16 channels in all): the LDS of one workgroup
    const dim3 grid(h.nblk, count);
    if (h.ML == 20 && h.W == 8) { PSA_BWD_BUCKETS(20, 8) }
    else if (h.ML == 14 && h.W == 4) { PSA_BWD_BUCKETS(14, 4) }
    else return MST_ERR_UNSUPPORTED;
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
    if (act == ACT_SIGOUT) { const float s = MST_FAST_RCP(1.f + MST_FAST_EXP(-z)); return col == 0 ? 6.f * s : s; }
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
                if (d.first) {                          // first writer of dx in this pass: nothing to read
#pragma unroll
                    for (int k = 0; k < KIN; ++k) dx[k] = 0.f;
                } else rl_load<KIN>(gx0 + (int64_t)row * KIN, dx);
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
