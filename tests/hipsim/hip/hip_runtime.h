// hipsim — TEST TOOLING ONLY.  A single-threaded CPU interpreter for the subset of HIP
// that music-style-transfer_amd/csrc uses, so the very same .hip sources can be executed
// (and address-sanitized) in the GPU-less build container before they are launched on an
// MI355X.  It is never linked into the product library: libmst_amd.so is built by hipcc for
// gfx950 only and has no CPU path.  Selected by putting tests/hipsim ahead of ROCm on the
// include path (tests/hipsim/build.sh).
//
// Model: blocks run one after another; the threads of a block are ucontext fibers that are
// resumed round-robin and park at __syncthreads() / wave collectives (shuffles, MFMA) until
// every live participant has arrived.  Wave = 64 lanes, as on gfx950.
// Kernels whose workgroups wait for each other (the multi-workgroup LSTM's tagged-granule exchange) are launched with
// launch_coop(): every block of the grid is alive at once, one State per block, the scheduler sweeps block after block and
// a spinning lane yields (s_sleep).  Such kernels declare their LDS through MST_COOP_LDS* (one copy per block) because
// `__shared__` is a function-level static here.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPSIM_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void**, const void*, size_t);
extern "C" void __sanitizer_finish_switch_fiber(void*, const void**, size_t*);
#endif
#endif

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIPSIM 1
#define MST_GLOBAL_AS
#define MST_CONST_AS
#define MST_UNIFORM(x) (x)
#define MST_LDS_BARRIER() __syncthreads()
#define MST_FAST_EXP(x) expf(x)
#define MST_FAST_RCP(x) (1.f / (x))
#define MST_PIN(x) ((void)0)
#define MST_WAVE_SYNC() hipsim::wave_barrier()
#define MST_SCHED_FENCE() ((void)0)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }

typedef struct ihipStream_t* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
static inline const char* hipGetErrorString(hipError_t e) { return e ? "hipsim error" : "no error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

namespace hipsim {

constexpr int WAVE = 64;
constexpr size_t STACK = 96 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    dim3 tid;
    bool done = true;
};

struct State {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    std::function<void()> body;
    int cur = -1, nthreads = 0, alive = 0;
    // block barrier
    int bar_arrived = 0; unsigned bar_gen = 0;
    // wave collectives
    int wave_alive[16]; int wave_arrived[16]; unsigned wave_gen[16];
    float slot_f[16][WAVE][4];
    long progress = 0;
    dim3 bidx;                  // launch_coop: the block this State runs
};
inline State*& coop_cur() { static State* p = nullptr; return p; }      // launch_coop: the State of the block being swept
inline State& st() { static State s; return coop_cur() ? *coop_cur() : s; }

}  // namespace hipsim

inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipsim {

inline void switch_to_sched() {
    State& s = st();
    Fiber& f = s.fibers[s.cur];
#ifdef HIPSIM_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(f.done ? nullptr : &fake, nullptr, 0);
#endif
    swapcontext(&f.ctx, &s.sched);
#ifdef HIPSIM_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

inline void fiber_entry() {
#ifdef HIPSIM_ASAN
    __sanitizer_finish_switch_fiber(nullptr, nullptr, nullptr);
#endif
    State& s = st();
    s.body();
    Fiber& f = s.fibers[s.cur];
    f.done = true;
    s.alive--;
    int lin = f.tid.x + blockDim.x * (f.tid.y + blockDim.y * f.tid.z);
    s.wave_alive[lin / WAVE]--;
    s.progress++;
    // a thread that exits releases barriers the others are parked on (s_barrier counts live waves)
    if (s.bar_arrived && s.bar_arrived >= s.alive) { s.bar_arrived = 0; s.bar_gen++; }
    int w = lin / WAVE;
    if (s.wave_arrived[w] && s.wave_arrived[w] >= s.wave_alive[w]) { s.wave_arrived[w] = 0; s.wave_gen[w]++; }
    switch_to_sched();
    abort();
}

inline void yield() {
    switch_to_sched();
    threadIdx = st().fibers[st().cur].tid;
    if (coop_cur()) blockIdx = st().bidx;
}

inline void block_barrier() {
    State& s = st();
    unsigned g = s.bar_gen;
    s.progress++;
    if (++s.bar_arrived >= s.alive) { s.bar_arrived = 0; s.bar_gen++; return; }
    while (s.bar_gen == g) yield();
}

inline int lin_tid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }

inline void wave_barrier() {
    State& s = st();
    int w = lin_tid() / WAVE;
    unsigned g = s.wave_gen[w];
    s.progress++;
    if (++s.wave_arrived[w] >= s.wave_alive[w]) { s.wave_arrived[w] = 0; s.wave_gen[w]++; return; }
    while (s.wave_gen[w] == g) yield();
}

inline void init_block(State& s, const std::function<void()>& body) {
    int n = blockDim.x * blockDim.y * blockDim.z;
    if (n > 1024 || n <= 0) { fprintf(stderr, "hipsim: bad block size %d\n", n); abort(); }
    if ((int)s.fibers.size() < n) s.fibers.resize(1024);
    s.body = body;
    s.nthreads = s.alive = n;
    s.bar_arrived = 0;
    for (int w = 0; w < 16; ++w) {
        s.wave_alive[w] = std::max(0, std::min(WAVE, n - w * WAVE));
        s.wave_arrived[w] = 0;
    }
    for (int i = 0; i < n; ++i) {
        Fiber& f = s.fibers[i];
        if (!f.stack) f.stack = (char*)malloc(STACK);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        f.done = false;
        f.tid = dim3(i % blockDim.x, (i / blockDim.x) % blockDim.y, i / (blockDim.x * blockDim.y));
    }
}

// one round-robin pass over the live fibers of a block
inline void sweep_block(State& s) {
    for (int i = 0; i < s.nthreads; ++i) {
        Fiber& f = s.fibers[i];
        if (f.done) continue;
        s.cur = i;
        threadIdx = f.tid;
#ifdef HIPSIM_ASAN
        void* fake = nullptr;
        __sanitizer_start_switch_fiber(&fake, f.stack, STACK);
#endif
        swapcontext(&s.sched, &f.ctx);
#ifdef HIPSIM_ASAN
        __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
    }
}

inline void run_block(const std::function<void()>& body) {
    State& s = st();
    init_block(s, body);
    while (s.alive > 0) {
        long before = s.progress;
        sweep_block(s);
        if (s.progress == before) {
            fprintf(stderr, "hipsim: deadlock (divergent barrier / collective) in block (%u,%u,%u)\n",
                    blockIdx.x, blockIdx.y, blockIdx.z);
            abort();
        }
    }
}

template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
    gridDim = grid;
    blockDim = block;
    std::function<void()> body = [=]() { kernel(args...); };
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                blockIdx = dim3(x, y, z);
                run_block(body);
            }
}

// Co-resident launch: all blocks alive at once.  A lane that polls another block's data yields through spin_yield(); the
// launch is declared dead only when no fiber of any block has made progress for very many sweeps (a polling loop with a
// bound of its own ends long before that).
constexpr int COOP_MAXB = 48;
inline int coop_block() { return (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); }
inline void spin_yield() { yield(); }

template <class K, class... A>
inline void launch_coop(K kernel, dim3 grid, dim3 block, hipStream_t, A... args) {
    gridDim = grid;
    blockDim = block;
    const int nb = (int)(grid.x * grid.y * grid.z);
    if (nb > COOP_MAXB) { fprintf(stderr, "hipsim: co-resident launch of %d blocks (max %d)\n", nb, COOP_MAXB); abort(); }
    std::function<void()> body = [=]() { kernel(args...); };
    std::vector<State*> blocks;
    for (int i = 0; i < nb; ++i) {
        State* s = new State();
        s->bidx = dim3(i % grid.x, (i / grid.x) % grid.y, i / (grid.x * grid.y));
        blockIdx = s->bidx;
        init_block(*s, body);
        blocks.push_back(s);
    }
    long idle = 0;
    for (;;) {
        int alive = 0;
        long before = 0, after = 0;
        for (State* s : blocks) before += s->progress;
        for (State* s : blocks) {
            if (s->alive <= 0) continue;
            coop_cur() = s;
            blockIdx = s->bidx;
            sweep_block(*s);
            alive += s->alive > 0;
        }
        coop_cur() = nullptr;
        if (!alive) break;
        for (State* s : blocks) after += s->progress;
        idle = after == before ? idle + 1 : 0;
        if (idle > (1 << 16)) { fprintf(stderr, "hipsim: co-resident launch made no progress\n"); abort(); }
    }
    for (State* s : blocks) {
        for (Fiber& f : s->fibers) free(f.stack);
        delete s;
    }
}

template <class T>
inline T wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    State& s = st();
    int lin = lin_tid(), w = lin / WAVE, lane = lin % WAVE;
    memcpy(&s.slot_f[w][lane][0], &v, 4);
    wave_barrier();
    T r;
    memcpy(&r, &s.slot_f[w][src_lane & (WAVE - 1)][0], 4);
    wave_barrier();
    return r;
}

}  // namespace hipsim

#define hipLaunchKernelGGL(k, g, b, sh, stream, ...) hipsim::launch((k), (g), (b), (sh), (stream), __VA_ARGS__)

static inline void __syncthreads() { hipsim::block_barrier(); }
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return hipsim::wave_exchange(v, (hipsim::lin_tid() % 64) ^ mask); }
template <class T> static inline T __shfl_down(T v, unsigned d, int = 64) {
    int lane = hipsim::lin_tid() % 64;
    return hipsim::wave_exchange(v, lane + (int)d < 64 ? lane + (int)d : lane);
}
template <class T> static inline T __shfl(T v, int src, int = 64) { return hipsim::wave_exchange(v, src); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float __fdividef(float a, float b) { return a / b; }
// agent-scope atomics, bit casts, sleep and the wall clock of the multi-workgroup LSTM's granule exchange: fibers never
// pre-empt each other, so plain accesses are atomic; s_sleep is where a polling lane lets the other blocks run; the clock
// is a call counter (the kernels' time bound becomes a bound on polls)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_fetch_or(p, v, order, scope) hipsim_fetch_or((p), (v))
static inline int hipsim_fetch_or(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline void __builtin_amdgcn_s_sleep(int) { hipsim::spin_yield(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline long long wall_clock64() { static long long t = 0; hipsim::st().progress++; return ++t; }
static inline void __threadfence() {}

// f32 MFMA (gfx950 v_mfma_f32_16x16x4_f32): bit-for-bit a k-ordered fmaf chain per the CDNA4
// guide.  Lane l supplies A[row=l&15][k=l>>4] and B[k=l>>4][col=l&15]; D[row=(l>>4)*4+j][col=l&15].
typedef float hipsim_f32x4 __attribute__((ext_vector_type(4)));
static inline hipsim_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipsim_f32x4 c, int, int, int) {
    using namespace hipsim;
    State& s = st();
    int lin = lin_tid(), w = lin / WAVE, lane = lin % WAVE;
    s.slot_f[w][lane][0] = a;
    s.slot_f[w][lane][1] = b;
    wave_barrier();
    int col = lane & 15;
    for (int j = 0; j < 4; ++j) {
        int row = (lane >> 4) * 4 + j;
        float acc = c[j];
        for (int k = 0; k < 4; ++k) acc = fmaf(s.slot_f[w][row + 16 * k][0], s.slot_f[w][col + 16 * k][1], acc);
        c[j] = acc;
    }
    wave_barrier();
    return c;
}

// f32 MFMA 32x32x2: lane l supplies A[row=l&31][k=l>>5] and B[k=l>>5][col=l&31];
// D register r of lane l is C[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]; k-ordered fmaf chain.
typedef float hipsim_f32x16 __attribute__((ext_vector_type(16)));
static inline hipsim_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipsim_f32x16 c, int, int, int) {
    using namespace hipsim;
    State& s = st();
    int lin = lin_tid(), w = lin / WAVE, lane = lin % WAVE;
    s.slot_f[w][lane][0] = a;
    s.slot_f[w][lane][1] = b;
    wave_barrier();
    int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(s.slot_f[w][row + 32 * k][0], s.slot_f[w][col + 32 * k][1], acc);
        c[r] = acc;
    }
    wave_barrier();
    return c;
}

// ---- host-side runtime bits used by plan.hip
using std::max;
using std::min;
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
// events (timing is meaningless on the interpreter; the API just has to exist)
typedef struct ihipEvent_t* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
// multi-stream plumbing used by the plan's dataflow lanes (everything is sequential on the interpreter)
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
