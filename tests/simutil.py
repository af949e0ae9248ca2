"""Helpers for the CPU-interpreter (hipsim) tests: builds tests/hipsim/libmst_sim.so from the
product's .hip sources and binds it through the product's own ctypes layer."""
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, 'tests', 'hipsim')
CSRC = os.path.join(ROOT, 'music-style-transfer_amd', 'csrc')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
_cache = {}


def sim_native(asan=False):
    from style import _native
    name = 'libmst_sim_asan.so' if asan else 'libmst_sim.so'
    path = os.path.join(SIM_DIR, name)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(SIM_DIR, 'hip', 'hip_runtime.h'),
                                                                 os.path.join(ROOT, 'include', 'mst_amd.h')]
    stale = not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if stale:
        env = dict(os.environ, ASAN='1' if asan else '0')
        subprocess.run([os.path.join(SIM_DIR, 'build.sh')], check=True, env=env, capture_output=True)
    if path not in _cache:
        _cache[path] = _native.Native(path)
    return _cache[path]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def make_dims(widths, C, R, T, unpitched, instr=51, n_instruments=41, clips=1):
    from style._native import Dims
    return Dims(C=C, R=R, T=T, beat=widths['beat'], bar=widths['bar'], nrf=widths['nrf'], style=widths['style'],
                melody=widths['melody'], rhythm=widths['rhythm'], instr=instr, n_instruments=n_instruments,
                has_unpitched=int(unpitched), clips=clips)


def flat_from_named(native, dims, named):
    """Pack {state_dict name: array} into the flat parameter buffer of the C ABI."""
    table = native.param_table(dims)
    flat = torch.zeros(native.param_floats(dims), dtype=torch.float32)
    assert set(n for n, _, _ in table) == set(named), set(n for n, _, _ in table) ^ set(named)
    for name, off, shape in table:
        a = torch.as_tensor(np.asarray(named[name]), dtype=torch.float32)
        assert tuple(a.shape) == shape, (name, tuple(a.shape), shape)
        flat[off:off + a.numel()] = a.reshape(-1)
    return flat, table
