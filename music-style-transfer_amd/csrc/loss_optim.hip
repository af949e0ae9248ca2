// get_total_loss (style/model.py:847-997), Adam+StepLR (train-model.py:89-90,151-154) and
// hard_output (style/model.py:818-832) for gfx950.
//
// Loss = one streaming reduction over predictions/targets (7 partial sums per note tensor,
// HBM-bound, coalesced, per-workgroup partials re-summed in order), a single-lane scalar tail
// tail that evaluates the reference's loss tree with forward-mode dual numbers (16 lanes, one
// per tape input) — so every data-dependent Python branch of the reference (safe_div, safe_sqrt)
// is taken on the device with identical values AND gradients, with no host sync — and one
// elementwise backward pass.  The tail yields the full Jacobian d(leaf)/d(partial sum), so any
// loss leaf can be differentiated.
#include "mst_common.h"

#define LOSS_MAXBLK 256
#define NP_SUMS 7            // TP FP FN SEvel SEdur BCE Nmask
#define N_TAPE_IN 16         // 7 pitched + 6 unpitched + instruments, mode, bpm raw losses
#define SAVED_J 0            // saved[k*16 + j] = d leaf_k / d input_j
#define EPS_DIV 1e-7f

int64_t mst_loss_scratch_floats(void) { return 2 * LOSS_MAXBLK * 8 + 64; }

// ------------------------------------------------------------------ streaming partial sums
template <int NFEAT>
__device__ __forceinline__ void note_terms(const float* p, const float* t, float* acc) {
    const float pv = p[1], tv = t[1];
    const float m = tv > 0.f ? 1.f : 0.f;
    acc[0] += fminf(pv, tv);
    acc[1] += fmaxf(pv - tv, 0.f);
    acc[2] += fmaxf(tv - pv, 0.f);
    const float dv = tv - pv;
    acc[3] += dv * dv * m;
    const float dd = (p[0] - fminf(t[0], 6.f)) / 6.f;
    acc[4] += dd * dd * m;
    if (NFEAT == 5) {
        float bce = 0.f;
#pragma unroll
        for (int a = 2; a < 5; ++a) {   // F.binary_cross_entropy clamps both logs at -100
            const float lp = fmaxf(logf(p[a]), -100.f), lq = fmaxf(logf(1.f - p[a]), -100.f);
            bce -= t[a] * lp + (1.f - t[a]) * lq;
        }
        acc[5] += bce * m;
    }
    acc[6] += m;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// grid (nblk_p + nblk_u): first nblk_p workgroups reduce the pitched tensor, the rest the unpitched
// blockIdx.y = clip of a batched plan (per-clip pointer strides in lb; all 0 for a single clip)
__global__ __launch_bounds__(256) void loss_partials_kernel(const float* pp, const float* pt, int64_t np, int nblk_p,
                                                            const float* up, const float* ut, int64_t nu, int nblk_u,
                                                            float* scratch, LossBatch lb) {
    __shared__ float red[4][NP_SUMS];
    __shared__ float tile_p[256 * 5], tile_t[256 * 5];
    {
        const int64_t k = blockIdx.y;
        pp += k * lb.ws; pt += k * lb.ext0; up += k * lb.ws; ut += k * lb.ext1; scratch += k * lb.tmp;
    }
    const bool pitched = (int)blockIdx.x < nblk_p;
    const int blk = pitched ? blockIdx.x : blockIdx.x - nblk_p;
    const int nb = pitched ? nblk_p : nblk_u;
    const int64_t n = pitched ? np : nu;
    float acc[NP_SUMS];
#pragma unroll
    for (int k = 0; k < NP_SUMS; ++k) acc[k] = 0.f;
    // a tile of 256 positions is staged through LDS with unit-stride loads (a lane's own 5-float record is a stride-5
    // access: every wave load touched 5 x the cache lines it used); lane <-> position and the order of a lane's sums are unchanged
    const int nf = pitched ? 5 : 2;
    const float* P = pitched ? pp : up;
    const float* T = pitched ? pt : ut;
    for (int64_t base = (int64_t)blk * 256; base < n; base += (int64_t)nb * 256) {
        const int rows = (int)(n - base < 256 ? n - base : 256), cnt = rows * nf;
        // global-address-space pointers: through generic ones every load waits for the previous LDS store (may alias)
        const MST_GLOBAL_AS float* Pg = (const MST_GLOBAL_AS float*)(P + base * nf);
        const MST_GLOBAL_AS float* Tg = (const MST_GLOBAL_AS float*)(T + base * nf);
        {
            float pv[5], tv[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {           // clamped, unconditional: all ten loads in flight
                const int j = min((int)threadIdx.x + 256 * u, cnt - 1);
                pv[u] = Pg[j]; tv[u] = Tg[j];
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int j = threadIdx.x + 256 * u;
                if (j < cnt) { tile_p[j] = pv[u]; tile_t[j] = tv[u]; }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            if (pitched) note_terms<5>(tile_p + threadIdx.x * 5, tile_t + threadIdx.x * 5, acc);
            else note_terms<2>(tile_p + threadIdx.x * 2, tile_t + threadIdx.x * 2, acc);
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NP_SUMS; ++k) acc[k] = wave_sum(acc[k]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NP_SUMS; ++k) red[wv][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < NP_SUMS) {
        const int k = threadIdx.x;
        float* dst = scratch + (pitched ? 0 : LOSS_MAXBLK * 8);
        dst[blk * 8 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// ------------------------------------------------------------------ scalar tail, forward-mode AD
// The reference's loss tree (style/model.py:847-997) evaluated with dual numbers: lane j (< 16)
// carries d/d(input_j), so one pass over ~100 scalar ops yields every loss leaf AND the full
// Jacobian d(leaf)/d(partial sum) — all in registers, no tape.  The reference's data-dependent
// Python branches (safe_div, safe_sqrt) become value-dependent selects with identical values
// and gradients.
struct Dual { float v, d; };
__device__ __forceinline__ Dual dl(float v, float d) { Dual r; r.v = v; r.d = d; return r; }
__device__ __forceinline__ Dual d_add(Dual a, Dual b) { return dl(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual d_mul(Dual a, Dual b) { return dl(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual d_cmul(Dual a, float c) { return dl(a.v * c, a.d * c); }
__device__ __forceinline__ Dual d_one_minus(Dual a) { return dl(1.f - a.v, -a.d); }
__device__ __forceinline__ Dual d_sqr(Dual a) { return dl(a.v * a.v, 2.f * a.v * a.d); }
__device__ __forceinline__ Dual d_div(Dual a, Dual b) { const float q = a.v / b.v; return dl(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ Dual d_safe_div(Dual a, Dual b) {          // style/model.py:854-860
    float den = b.v;
    if (fabsf(den) < EPS_DIV) den = den < 0.f ? den - EPS_DIV : den + EPS_DIV;
    const float q = a.v / den;
    return dl(q, (a.d - q * b.d) / den);
}
__device__ __forceinline__ Dual d_safe_sqrt(Dual a) {                  // style/utils/pytorch.py:68-71
    if (a.v == 0.f) return dl(0.f, 0.f);
    const float r = sqrtf(a.v);
    return dl(r, a.d * 0.5f / r);
}
__device__ __forceinline__ Dual d_tanh(Dual a) { const float t = tanhf(a.v); return dl(t, a.d * (1.f - t * t)); }
// quadratic mean with constant weights 1/k (get_mean, style/utils/pytorch.py:74-94)
__device__ __forceinline__ Dual d_qmean2(Dual a, Dual b) { return d_safe_sqrt(d_add(d_cmul(d_sqr(a), 0.5f), d_cmul(d_sqr(b), 0.5f))); }
__device__ __forceinline__ Dual d_qmean3(Dual a, Dual b, Dual c) {
    const float w = (float)(1.0 / 3.0);
    return d_safe_sqrt(d_add(d_add(d_cmul(d_sqr(a), w), d_cmul(d_sqr(b), w)), d_cmul(d_sqr(c), w)));
}

struct ChannelLeaves { Dual total, notes, vel, dur, acc; };

// channels losses of one note tensor from its partial sums (style/model.py:863-932)
__device__ __forceinline__ ChannelLeaves channel_tree(const Dual* in, bool pitched, bool normalize) {
    const Dual TP = in[0], FP = in[1], FN = in[2], SEV = in[3], SED = in[4];
    const Dual NM = pitched ? in[6] : in[5];
    ChannelLeaves o;
    const Dual prec = d_safe_div(TP, d_add(TP, FP));
    const Dual rec = d_safe_div(TP, d_add(TP, FN));
    const Dual f = d_cmul(d_safe_div(d_mul(prec, rec), d_add(prec, rec)), 2.f);
    o.notes = d_one_minus(f);
    o.vel = d_div(SEV, NM);
    o.dur = d_div(SED, NM);
    // first learn the right notes, then the right velocities: weights [notes, 1 - notes] are live
    const Dual nv = d_safe_sqrt(d_add(d_mul(o.notes, d_sqr(o.notes)), d_mul(d_one_minus(o.notes), d_sqr(o.vel))));
    if (pitched) {
        o.acc = d_div(in[5], d_cmul(NM, 3.f));
        if (normalize) o.acc = d_tanh(o.acc);
        o.total = d_qmean3(o.dur, o.acc, nv);
    } else {
        o.acc = dl(0.f, 0.f);
        o.total = d_qmean2(o.dur, nv);
    }
    return o;
}

__global__ __launch_bounds__(64) void loss_tail_kernel(const float* scratch, int nblk_p, int nblk_u, int has_u,
                                                       const float* il, const float* it, int ni,
                                                       const float* mlg, const float* mt,
                                                       const float* bp, const float* bt, int normalize,
                                                       float* losses, float* saved, LossBatch lb,
                                                       float* gl_onehot, float* losses_out) {
    __shared__ float sums[N_TAPE_IN];
    const int tid = threadIdx.x;
    {
        const int64_t k = blockIdx.y;
        scratch += k * lb.tmp; il += k * lb.ws; it += k * lb.ws; mlg += k * lb.ws; mt += k * lb.ws; bp += k * lb.ws;
        bt += k * lb.ws; losses += k * lb.ws; saved += k * lb.ws;
        // train-iteration extras (null for the stand-alone entry point): the upstream gradient of loss.backward() — one-hot
        // on the total — and a dense copy of the leaves for the caller, written here instead of by two more launches
        if (gl_onehot) gl_onehot += k * lb.ws;
        if (losses_out) losses_out += k * MST_N_LOSSES;
    }
    // partial rows hold 7 sums {TP FP FN SEvel SEdur BCE Nmask}; tape inputs 0..6 are the pitched tensor's,
    // 7..12 the unpitched tensor's {TP FP FN SEvel SEdur Nmask}.  Lanes stride over the per-workgroup partials,
    // then a fixed-order wave reduction (the tail is one 64-lane wave).
    // A lane takes whole 8-float partial rows (two 16-byte loads per row, both tensors' loops in flight together) instead of
    // fourteen dependent strided passes; per lane and per sum the order over q, and the wave reduction, are the same.
    {
        float sp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, su[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        typedef float row4 __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte load, 4-byte alignment suffices
        const row4* rp = reinterpret_cast<const row4*>(scratch);
        const row4* ru = reinterpret_cast<const row4*>(scratch + LOSS_MAXBLK * 8);
        const int nbu = has_u ? nblk_u : 0;
        for (int q = tid; q < nblk_p; q += 64) {
            const row4 x = rp[2 * q], y = rp[2 * q + 1];
            sp[0] += x.x; sp[1] += x.y; sp[2] += x.z; sp[3] += x.w; sp[4] += y.x; sp[5] += y.y; sp[6] += y.z;
        }
        for (int q = tid; q < nbu; q += 64) {
            const row4 x = ru[2 * q], y = ru[2 * q + 1];
            su[0] += x.x; su[1] += x.y; su[2] += x.z; su[3] += x.w; su[4] += y.x; su[5] += y.y; su[6] += y.z;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) { sp[k] = wave_sum(sp[k]); su[k] = wave_sum(su[k]); }
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) sums[k] = sp[k];
#pragma unroll
            for (int k = 0; k < 5; ++k) sums[7 + k] = su[k];
            sums[12] = su[6];
        }
    }
    // instruments: BCE-with-logits, mean over ni (style/model.py:903)
    float v = 0.f;
    for (int j = tid; j < ni; j += 64) {
        const float x = il[j];
        v += fmaxf(x, 0.f) - x * it[j] + log1pf(expf(-fabsf(x)));
    }
    v = wave_sum(v);
    if (tid == 0) {
        sums[13] = v / (float)ni;
        // mode: cross entropy against argmax of the one-hot target (style/model.py:904), 2 classes
        const int tgt = mt[1] > mt[0] ? 1 : 0;
        const float mx = fmaxf(mlg[0], mlg[1]);
        const float lse = mx + logf(expf(mlg[0] - mx) + expf(mlg[1] - mx));
        sums[14] = lse - mlg[tgt];
        const float db = (bp[0] - bt[0]) / 150.f;
        sums[15] = db * db;
    }
    __syncthreads();
    if (tid >= N_TAPE_IN) return;
    Dual in[N_TAPE_IN];
#pragma unroll
    for (int j = 0; j < N_TAPE_IN; ++j) in[j] = dl(sums[j], j == tid ? 1.f : 0.f);
    Dual leaf[MST_N_LOSSES];
    bool present[MST_N_LOSSES];
#pragma unroll
    for (int k = 0; k < MST_N_LOSSES; ++k) { leaf[k] = dl(0.f, 0.f); present[k] = true; }
    const ChannelLeaves p = channel_tree(in, true, normalize != 0);
    leaf[MST_L_P_TOTAL] = p.total; leaf[MST_L_P_NOTES] = p.notes; leaf[MST_L_P_VELOCITY] = p.vel;
    leaf[MST_L_P_DURATION] = p.dur; leaf[MST_L_P_ACCIDENTALS] = p.acc;
    if (has_u) {
        const ChannelLeaves u = channel_tree(in + 7, false, normalize != 0);
        leaf[MST_L_U_TOTAL] = u.total; leaf[MST_L_U_NOTES] = u.notes; leaf[MST_L_U_VELOCITY] = u.vel;
        leaf[MST_L_U_DURATION] = u.dur;
        leaf[MST_L_CH_TOTAL] = d_qmean2(p.total, u.total);
    } else {
        present[MST_L_U_TOTAL] = present[MST_L_U_NOTES] = present[MST_L_U_VELOCITY] = present[MST_L_U_DURATION] = false;
        leaf[MST_L_CH_TOTAL] = p.total;
    }
    Dual li = in[13], lm = in[14];
    if (normalize) { li = d_tanh(li); lm = d_tanh(lm); }
    leaf[MST_L_SI_INSTRUMENTS] = li;
    leaf[MST_L_SI_MODE] = lm;
    leaf[MST_L_SI_BPM] = in[15];
    leaf[MST_L_SI_TOTAL] = d_qmean3(li, lm, in[15]);
    leaf[MST_L_TOTAL] = d_qmean2(leaf[MST_L_CH_TOTAL], leaf[MST_L_SI_TOTAL]);
#pragma unroll
    for (int k = 0; k < MST_N_LOSSES; ++k) {
        if (tid == 0) {
            const float v = present[k] ? leaf[k].v : __builtin_nanf("");
            losses[k] = v;
            if (losses_out) losses_out[k] = v;
            if (gl_onehot) gl_onehot[k] = k == MST_L_TOTAL ? 1.f : 0.f;
        }
        saved[SAVED_J + k * N_TAPE_IN + tid] = present[k] ? leaf[k].d : 0.f;
    }
}

// ------------------------------------------------------------------ elementwise backward
// grid (nblk_p + nblk_u + 1): note-tensor gradients, last workgroup = song-info head gradients
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* pp, const float* pt, int64_t np, int nblk_p,
                                                       const float* up, const float* ut, int64_t nu, int nblk_u,
                                                       const float* il, const float* it, int ni,
                                                       const float* mlg, const float* mt, const float* bp, const float* bt,
                                                       const float* saved, const float* gl,
                                                       float* gp, float* gu, float* gi, float* gm, float* gb, LossBatch lb, float info_scale) {
    __shared__ float coef[N_TAPE_IN];
    __shared__ float tile_p[256 * 5], tile_t[256 * 5];
    {
        const int64_t k = blockIdx.y;
        pp += k * lb.ws; pt += k * lb.ext0; up += k * lb.ws; ut += k * lb.ext1;
        il += k * lb.ws; it += k * lb.ws; mlg += k * lb.ws; mt += k * lb.ws; bp += k * lb.ws; bt += k * lb.ws;
        saved += k * lb.ws; gl += k * lb.ws;
        gp += k * lb.grad; gu += k * lb.grad; gi += k * lb.grad; gm += k * lb.grad; gb += k * lb.grad;
    }
    if (threadIdx.x < N_TAPE_IN) {
        // all 30 loads first (a conditional load per leaf was a chain of 15 dependent round trips at the head of every workgroup)
        float gk[MST_N_LOSSES], sk[MST_N_LOSSES];
#pragma unroll
        for (int k = 0; k < MST_N_LOSSES; ++k) { gk[k] = gl[k]; sk[k] = saved[SAVED_J + k * N_TAPE_IN + threadIdx.x]; }
        float c = 0.f;
#pragma unroll
        for (int k = 0; k < MST_N_LOSSES; ++k) c += gk[k] != 0.f ? gk[k] * sk[k] : 0.f;
        coef[threadIdx.x] = c;
    }
    __syncthreads();
    const int bx = blockIdx.x;
    if (bx < nblk_p + nblk_u) {
        const bool pitched = bx < nblk_p;
        const int blk = pitched ? bx : bx - nblk_p;
        const int nb = pitched ? nblk_p : nblk_u;
        const int64_t n = pitched ? np : nu;
        const int nf = pitched ? 5 : 2;
        const float* P = pitched ? pp : up;
        const float* T = pitched ? pt : ut;
        float* G = pitched ? gp : gu;
        const float* c = coef + (pitched ? 0 : 7);
        const float cTP = c[0], cFP = c[1], cFN = c[2], cSEV = c[3], cSED = c[4];
        const float cBCE = pitched ? c[5] : 0.f;
        // tiles of 256 positions through LDS: unit-stride loads and stores (see loss_partials_kernel)
        for (int64_t base = (int64_t)blk * 256; base < n; base += (int64_t)nb * 256) {
            const int rows = (int)(n - base < 256 ? n - base : 256), cnt = rows * nf;
            const MST_GLOBAL_AS float* Pg = (const MST_GLOBAL_AS float*)(P + base * nf);
            const MST_GLOBAL_AS float* Tg = (const MST_GLOBAL_AS float*)(T + base * nf);
            MST_GLOBAL_AS float* Gg = (MST_GLOBAL_AS float*)(G + base * nf);
            {
                float pv[5], tv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {       // clamped, unconditional: all ten loads in flight
                    const int j = min((int)threadIdx.x + 256 * u, cnt - 1);
                    pv[u] = Pg[j]; tv[u] = Tg[j];
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int j = threadIdx.x + 256 * u;
                    if (j < cnt) { tile_p[j] = pv[u]; tile_t[j] = tv[u]; }
                }
            }
            __syncthreads();
            if ((int)threadIdx.x < rows) {
                float* p = tile_p + threadIdx.x * nf;             // the gradient record replaces the prediction's
                const float* t = tile_t + threadIdx.x * nf;
                const float pv = p[1], tv = t[1], p0 = p[0];
                const float m = tv > 0.f ? 1.f : 0.f;
                // torch.min splits the gradient on ties; relu'(0) = 0
                float gv = cTP * (pv < tv ? 1.f : (pv == tv ? 0.5f : 0.f));
                gv += cFP * (pv - tv > 0.f ? 1.f : 0.f) - cFN * (tv - pv > 0.f ? 1.f : 0.f);
                gv -= cSEV * 2.f * (tv - pv) * m;
                p[1] = gv;
                p[0] = cSED * 2.f * (p0 - fminf(t[0], 6.f)) * (1.f / 36.f) * m;
                if (pitched) {
#pragma unroll
                    for (int a = 2; a < 5; ++a)   // aten binary_cross_entropy_backward
                        p[a] = cBCE * m * (p[a] - t[a]) / fmaxf((1.f - p[a]) * p[a], 1e-12f);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int j = threadIdx.x + 256 * u;
                if (j < cnt) Gg[j] = tile_p[j];
            }
            __syncthreads();
        }
    } else {
        const float ci = coef[13] * info_scale, cm = coef[14] * info_scale, cb = coef[15] * info_scale;
        // d tanh already folded into coef (the tape's inputs are the raw losses)
        for (int j = threadIdx.x; j < ni; j += 256) gi[j] = ci * (1.f / (1.f + expf(-il[j])) - it[j]) / (float)ni;
        if (threadIdx.x == 0) {
            const int tgt = mt[1] > mt[0] ? 1 : 0;
            const float mx = fmaxf(mlg[0], mlg[1]);
            const float e0 = expf(mlg[0] - mx), e1 = expf(mlg[1] - mx);
            gm[0] = cm * (e0 / (e0 + e1) - (tgt == 0 ? 1.f : 0.f));
            gm[1] = cm * (e1 / (e0 + e1) - (tgt == 1 ? 1.f : 0.f));
            gb[0] = cb * 2.f * (bp[0] - bt[0]) / (150.f * 150.f);
        }
    }
}

static int blocks_for(int64_t n) {
    int64_t b = (n + 1023) / 1024;
    if (b < 1) b = 1;
    if (b > LOSS_MAXBLK) b = LOSS_MAXBLK;
    return (int)b;
}

int loss_fwd_batched(const float* pp, const float* pt, int64_t np, const float* up, const float* ut, int64_t nu, const float* il,
                     const float* it, int ni, const float* mlg, const float* mt, const float* bp, const float* bt, int normalize,
                     float* losses, float* saved, float* scratch, LossBatch lb, hipStream_t s, float* gl_onehot, float* losses_out) {
    if (!pp || !pt || !il || !it || !mlg || !mt || !bp || !bt || !losses || !saved || !scratch || np <= 0 || lb.clips < 1)
        return MST_ERR_ARG;
    const int has_u = (up && ut && nu > 0) ? 1 : 0;
    const int nbp = blocks_for(np), nbu = has_u ? blocks_for(nu) : 0;
    hipLaunchKernelGGL(loss_partials_kernel, dim3(nbp + nbu, lb.clips), dim3(256), 0, s, pp, pt, np, nbp, up, ut,
                       has_u ? nu : (int64_t)0, nbu, scratch, lb);
    hipLaunchKernelGGL(loss_tail_kernel, dim3(1, lb.clips), dim3(64), 0, s, (const float*)scratch, nbp, nbu, has_u, il, it,
                       ni, mlg, mt, bp, bt, normalize, losses, saved, lb, gl_onehot, losses_out);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

int loss_bwd_batched(const float* pp, const float* pt, int64_t np, const float* up, const float* ut, int64_t nu, const float* il,
                     const float* it, int ni, const float* mlg, const float* mt, const float* bp, const float* bt,
                     const float* saved, const float* gl, float* gp, float* gu, float* gi, float* gm, float* gb, LossBatch lb,
                     hipStream_t s, float info_scale) {
    if (!pp || !pt || !saved || !gl || !gi || !gm || !gb || np <= 0 || lb.clips < 1) return MST_ERR_ARG;
    const int has_u = (up && ut && gu && nu > 0) ? 1 : 0;
    // gp == nullptr: the pitched tensor's gradient is not wanted here (the applier's backward kernel computes it on the fly)
    const int nbp = gp ? blocks_for(np) : 0, nbu = has_u ? blocks_for(nu) : 0;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(nbp + nbu + 1, lb.clips), dim3(256), 0, s, pp, pt, np, nbp, up, ut,
                       has_u ? nu : (int64_t)0, nbu, il, it, ni, mlg, mt, bp, bt, saved, gl, gp, gu, gi, gm, gb, lb, info_scale);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

static const LossBatch ONE_CLIP = {1, 0, 0, 0, 0, 0};

int loss_blocks(int64_t n) { return blocks_for(n); }

int loss_fwd_partials(const float* pp, const float* pt, int64_t np, const float* up, const float* ut, int64_t nu, float* scratch,
                      hipStream_t s) {
    const int has_u = (up && ut && nu > 0) ? 1 : 0;
    const int nbp = blocks_for(np), nbu = has_u ? blocks_for(nu) : 0;
    hipLaunchKernelGGL(loss_partials_kernel, dim3(nbp + nbu, 1), dim3(256), 0, s, pp, pt, np, nbp, up, ut,
                       has_u ? nu : (int64_t)0, nbu, scratch, ONE_CLIP);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

int loss_fwd_tail(int64_t np, int64_t nu, int has_u, const float* il, const float* it, int ni, const float* mlg, const float* mt,
                  const float* bp, const float* bt, int normalize, float* losses, float* saved, float* scratch, hipStream_t s,
                  float* gl_onehot, float* losses_out) {
    const int nbp = blocks_for(np), nbu = has_u ? blocks_for(nu) : 0;
    hipLaunchKernelGGL(loss_tail_kernel, dim3(1, 1), dim3(64), 0, s, (const float*)scratch, nbp, nbu, has_u, il, it,
                       ni, mlg, mt, bp, bt, normalize, losses, saved, ONE_CLIP, gl_onehot, losses_out);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

extern "C" int32_t mst_total_loss_fwd(const float* pp, const float* pt, int64_t np, const float* up, const float* ut,
                                      int64_t nu, const float* il, const float* it, int32_t ni, const float* mlg,
                                      const float* mt, const float* bp, const float* bt, int32_t normalize,
                                      float* losses, float* saved, float* scratch, mst_stream stream) {
    return loss_fwd_batched(pp, pt, np, up, ut, nu, il, it, (int)ni, mlg, mt, bp, bt, (int)normalize, losses, saved, scratch,
                            ONE_CLIP, (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int32_t mst_total_loss_bwd(const float* pp, const float* pt, int64_t np, const float* up, const float* ut,
                                      int64_t nu, const float* il, const float* it, int32_t ni, const float* mlg,
                                      const float* mt, const float* bp, const float* bt, const float* saved,
                                      const float* gl, float* gp, float* gu, float* gi, float* gm, float* gb,
                                      mst_stream stream) {
    return loss_bwd_batched(pp, pt, np, up, ut, nu, il, it, (int)ni, mlg, mt, bp, bt, saved, gl, gp, gu, gi, gm, gb, ONE_CLIP,
                            (hipStream_t)stream);
}

// ------------------------------------------------------------------ Adam + StepLR
// state[0] = optimizer steps taken so far (t); the prepare kernel turns it into the step's
// scalars in double precision exactly as torch's Python-side arithmetic does, then bumps t.
__global__ void adam_prepare_kernel(float* state, double lr0, double b1, double b2, int step_size, double gamma) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int t = (int)state[0] + 1;
    const double lr = lr0 * pow(gamma, (double)((t - 1) / step_size));
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    state[0] = (float)t;
    state[1] = (float)(lr / bc1);
    state[2] = (float)sqrt(bc2);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ g2,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const float* __restrict__ state, float b1, float b2, float eps, int zero_grad) {
    const float step = state[1], bc2s = state[2];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // g2: gradients of a second, concurrently run accumulation iteration (same sum as accumulating in place)
        const float gi = g2 ? g[i] + g2[i] : g[i];
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);        // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * (mi / (sqrtf(vi) / bc2s + eps));
        if (zero_grad) { g[i] = 0.f; if (g2) g2[i] = 0.f; }
    }
}

static int32_t adam_launch(float* params, float* grads, float* grads2, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                           double lr0, double beta1, double beta2, double eps, int32_t step_size, double gamma,
                           int32_t zero_grad, mst_stream stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !state || n <= 0 || step_size <= 0) return MST_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, s, state, lr0, beta1, beta2, (int)step_size, gamma);
    int64_t nb = (n + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, s, params, grads, grads2, exp_avg, exp_avg_sq, n,
                       (const float*)state, (float)beta1, (float)beta2, (float)eps, (int)zero_grad);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}

extern "C" int32_t mst_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                 double lr0, double beta1, double beta2, double eps, int32_t step_size, double gamma,
                                 int32_t zero_grad, mst_stream stream) {
    return adam_launch(params, grads, nullptr, exp_avg, exp_avg_sq, n, state, lr0, beta1, beta2, eps, step_size, gamma, zero_grad, stream);
}

extern "C" int32_t mst_adam_step2(float* params, float* grads, float* grads2, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  float* state, double lr0, double beta1, double beta2, double eps, int32_t step_size,
                                  double gamma, int32_t zero_grad, mst_stream stream) {
    if (!grads2) return MST_ERR_ARG;
    return adam_launch(params, grads, grads2, exp_avg, exp_avg_sq, n, state, lr0, beta1, beta2, eps, step_size, gamma, zero_grad, stream);
}

// ------------------------------------------------------------------ hard_output
__global__ __launch_bounds__(256) void hard_output_kernel(float* x, float* out, int64_t n, int nf) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float* xi = x + i * nf;
        float* o = out + i * nf;
        o[0] = xi[0];
        const float vel = xi[1] > .01f ? xi[1] : 0.f;   // velocity *= (velocity > .01), in place on the input too
        xi[1] = vel;
        o[1] = vel;
        if (nf > 2) {
            const float mx = fmaxf(xi[2], fmaxf(xi[3], xi[4]));
            for (int a = 2; a < 5; ++a) o[a] = (xi[a] == mx && xi[a] > .1f) ? 1.f : 0.f;
        }
    }
}

extern "C" int32_t mst_hard_output(float* x, float* out, int64_t n_pos, int32_t nfeat, mst_stream stream) {
    if (!x || !out || n_pos <= 0 || (nfeat != 5 && nfeat != 2)) return MST_ERR_ARG;
    int64_t nb = (n_pos + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(hard_output_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, n_pos, (int)nfeat);
    return hipGetLastError() == hipSuccess ? MST_OK : MST_ERR_LAUNCH;
}
