// Probe (developer tool, not shipped): which shape of fork/join stream capture does hipGraph on this ROCm accept?
//   ./capture_probe <variant>    0: trivial kernels, same side stream + same events re-forked 10 times
//                                1: the same with a fresh event pair per fork
//                                2: mst_train_iteration with branches = 1 (whole-model plan, bench dims)
//                                3: mst_train_iteration with branches = 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/mst_amd.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main(int argc, char** argv) {
    const int v = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t main_s, side; CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    if (v < 2) {
        float *a, *b; CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
        std::vector<hipEvent_t> ev(40);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 10; ++i) {
            hipEvent_t f = v ? ev[2 * i] : ev[0], j = v ? ev[2 * i + 1] : ev[1];
            CK(hipEventRecord(f, main_s)); CK(hipStreamWaitEvent(side, f, 0));
            touch<<<4, 256, 0, side>>>(b, 1024); CK(hipEventRecord(j, side));
            touch<<<4, 256, 0, main_s>>>(a, 1024); CK(hipStreamWaitEvent(main_s, j, 0));
            touch<<<4, 256, 0, main_s>>>(a, 1024);
        }
        CK(hipStreamEndCapture(main_s, &g)); printf("captured\n"); fflush(stdout);
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); printf("instantiated\n"); fflush(stdout);
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, main_s));
        CK(hipStreamSynchronize(main_s));
        float ha, hb; CK(hipMemcpy(&ha, a, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost));
        printf("variant %d ok: a=%g (60) b=%g (30)\n", v, ha, hb);
        return 0;
    }
    if (v >= 10) {
        // cost of graph edges: 40 small dependent kernels as (10) one chain, (11) two chains of 20 with one fork and one join,
        // (12) 20 levels of two kernels, fork + join at every level, (13) two chains with a cross edge every 4 kernels
        float *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
        std::vector<hipEvent_t> ev(100);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const int n = 1 << 16;
        CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeGlobal));
        if (v == 10) { for (int i = 0; i < 40; ++i) touch<<<n / 256, 256, 0, main_s>>>(i & 1 ? a : b, n); }
        else if (v == 11) {
            CK(hipEventRecord(ev[0], main_s)); CK(hipStreamWaitEvent(side, ev[0], 0));
            for (int i = 0; i < 20; ++i) { touch<<<n / 256, 256, 0, main_s>>>(a, n); touch<<<n / 256, 256, 0, side>>>(b, n); }
            CK(hipEventRecord(ev[1], side)); CK(hipStreamWaitEvent(main_s, ev[1], 0));
        } else if (v == 12) {
            for (int i = 0; i < 20; ++i) {
                CK(hipEventRecord(ev[2 * i], main_s)); CK(hipStreamWaitEvent(side, ev[2 * i], 0));
                touch<<<n / 256, 256, 0, main_s>>>(a, n); touch<<<n / 256, 256, 0, side>>>(b, n);
                CK(hipEventRecord(ev[2 * i + 1], side)); CK(hipStreamWaitEvent(main_s, ev[2 * i + 1], 0));
            }
        } else {
            CK(hipEventRecord(ev[0], main_s)); CK(hipStreamWaitEvent(side, ev[0], 0));
            for (int i = 0; i < 20; ++i) {
                touch<<<n / 256, 256, 0, main_s>>>(a, n); touch<<<n / 256, 256, 0, side>>>(b, n);
                if (i % 4 == 3) { CK(hipEventRecord(ev[2 + i], side)); CK(hipStreamWaitEvent(main_s, ev[2 + i], 0)); }
            }
            CK(hipEventRecord(ev[1], side)); CK(hipStreamWaitEvent(main_s, ev[1], 0));
        }
        CK(hipStreamEndCapture(main_s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, main_s));
        CK(hipEventRecord(e0, main_s));
        for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, main_s));
        CK(hipEventRecord(e1, main_s)); CK(hipStreamSynchronize(main_s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d: %.1f us per replay (40 kernels)\n", v, ms * 1000 / 200);
        return 0;
    }
    mst_dims d{4, 16, 4, 64, 128, 8, 256, 8, 32, 51, 41, 1, 1};
    mst_plan_options o{}; o.branches = v == 2 ? 1 : 0;
    int32_t st = 0;
    mst_plan* p = mst_plan_create_ex(&d, &o, &st);
    if (!p) { printf("plan create failed %d\n", st); return 3; }
    const int64_t np = mst_param_floats(&d), nw = mst_plan_workspace_floats(p);
    float *par, *gp, *ws, *xp, *xu, *ls;
    const int64_t npit = (int64_t)d.C * d.R * d.T * 10 * 56 * 5, nun = (int64_t)d.R * d.T * 10 * 47 * 2;
    CK(hipMalloc(&par, np * 4)); CK(hipMalloc(&gp, np * 4)); CK(hipMalloc(&ws, nw * 4)); CK(hipMalloc(&xp, npit * 4)); CK(hipMalloc(&xu, nun * 4)); CK(hipMalloc(&ls, 64 * 4));
    std::vector<float> h(np); srand(1); for (auto& x : h) x = (rand() / (float)RAND_MAX - .5f) * .2f;
    CK(hipMemcpy(par, h.data(), np * 4, hipMemcpyHostToDevice)); CK(hipMemset(gp, 0, np * 4)); CK(hipMemset(ws, 0, nw * 4)); CK(hipMemset(xp, 0, npit * 4)); CK(hipMemset(xu, 0, nun * 4));
    for (int i = 0; i < 2; ++i) { st = mst_train_iteration(p, par, gp, ws, xp, xu, ls, main_s); if (st) { printf("eager iteration failed %d\n", st); return 4; } }
    CK(hipStreamSynchronize(main_s)); printf("eager ok\n"); fflush(stdout);
    CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeGlobal));
    st = mst_train_iteration(p, par, gp, ws, xp, xu, ls, main_s);
    printf("captured call returned %d\n", st); fflush(stdout);
    CK(hipStreamEndCapture(main_s, &g)); printf("captured\n"); fflush(stdout);
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("%zu nodes\n", nn); fflush(stdout);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); printf("instantiated\n"); fflush(stdout);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, main_s));
    CK(hipEventRecord(e0, main_s));
    for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, main_s));
    CK(hipEventRecord(e1, main_s)); CK(hipStreamSynchronize(main_s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    float hl[16]; CK(hipMemcpy(hl, ls, 64, hipMemcpyDeviceToHost));
    printf("variant %d ok: %.1f us per replayed iteration, total loss %g\n", v, ms * 1000 / 200, hl[0]);
    mst_plan_destroy(p);
    return 0;
}
