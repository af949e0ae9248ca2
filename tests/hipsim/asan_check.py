"""AddressSanitizer pass over the product's kernels on the hipsim interpreter (numpy only, no torch):
    ASAN=1 tests/hipsim/build.sh
    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) \
      ASAN_OPTIONS=detect_leaks=0 python tests/hipsim/asan_check.py
Runs one full train iteration (+Adam) on exactly-sized heap buffers so any out-of-bounds access of
workspace / parameters / note tensors / LDS arrays aborts with a report."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, 'libmst_sim_asan.so'))


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('C', 'R', 'T', 'beat', 'bar', 'nrf', 'style', 'melody', 'rhythm',
                                         'instr', 'n_instruments', 'has_unpitched', 'clips')]


class Opts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('gemm_tile', 'no_merge', 'gemm_run', 'tile_r0', 'tile_rows', 'lstm_flavour', 'dense_flavour', 'branches')]


lib.mst_plan_create_ex.restype = C.c_void_p
lib.mst_plan_workspace_floats.restype = C.c_int64
lib.mst_param_floats.restype = C.c_int64
P = C.c_void_p


def run(C_, R, T, widths, unp, clips=1, gemm=None, dense=0):
    d = Dims(C_, R, T, *widths, 51, 41, int(unp), clips)
    st = C.c_int32()
    o = Opts(gemm_tile={None: 0, 'mfma': 64, 'valu': 32}[gemm], dense_flavour=dense)
    plan = C.c_void_p(lib.mst_plan_create_ex(C.byref(d), C.byref(o), C.byref(st)))
    assert plan.value, st.value
    n = lib.mst_param_floats(C.byref(d))
    rng = np.random.default_rng(0)
    params = (rng.standard_normal(n) * 0.1).astype(np.float32)
    g = np.zeros(n, np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    ws = np.zeros(lib.mst_plan_workspace_floats(plan), np.float32)
    pitched = (rng.random((clips * C_ * R * T * 10 * 56, 5)) * (rng.random((clips * C_ * R * T * 10 * 56, 1)) < 0.05)).astype(np.float32)
    unpitched = (rng.random((clips * R * T * 10 * 47, 2)) * (rng.random((clips * R * T * 10 * 47, 1)) < 0.05)).astype(np.float32)
    losses = np.zeros(15 * clips, np.float32); state = np.zeros(4, np.float32)
    a = lambda x: x.ctypes.data_as(P)
    for it in range(2):
        e = lib.mst_train_iteration(plan, a(params), a(g), a(ws), a(pitched), a(unpitched) if unp else None, a(losses), None)
        assert e == 0, e
    lib.mst_adam_step.argtypes = [P, P, P, P, C.c_int64, P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_int32, P]
    assert lib.mst_adam_step(a(params), a(g), a(m), a(v), n, a(state), .01, .9, .999, 1e-8, 200, .9, 1, None) == 0
    lib.mst_plan_destroy(plan)
    print('ok', (C_, R, T), widths, unp, 'clips', clips, gemm or 'default gemm', 'dense', dense, 'total loss', float(losses[0]), 'finite grads', bool(np.isfinite(params).all()))


# (every MFMA of the emulated GEMMs is two fiber round trips per lane: shapes are kept tiny so the pass takes minutes)
run(2, 2, 1, (8, 6, 3, 12, 4, 6), True)
run(2, 1, 2, (8, 6, 3, 12, 4, 6), False)
run(1, 1, 1, (64, 128, 8, 256, 8, 32), True)
run(1, 3, 1, (64, 128, 8, 256, 8, 32), False)                      # three bars: the 12-workgroup LSTM exchanges h_t / dz_t (co-resident launch)
run(1, 2, 1, (8, 6, 3, 12, 4, 6), True, clips=3)                    # batched plan, 32x32 GEMM tiling
run(2, 1, 1, (8, 6, 3, 12, 4, 6), True, clips=2, gemm='mfma')       # batched plan on the 64x64 GEMM tiling
run(1, 1, 2, (64, 128, 8, 256, 8, 32), False, gemm='mfma')          # one clip, full widths, 64x64 tiling
run(2, 1, 1, (8, 6, 3, 12, 4, 6), True, clips=3, gemm='mfma', dense=2)   # lin.hip's kernels on every eligible Linear, ragged row totals
run(1, 1, 1, (64, 128, 8, 256, 8, 32), True, gemm='mfma', dense=2)       # ... at full widths (K = 514, 280, 82; N = 16 .. 376)
print('asan pass clean')
