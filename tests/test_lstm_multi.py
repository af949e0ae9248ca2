"""The multi-workgroup H = 192 LSTM (csrc/lstm.hip: 12 workgroups per sequence exchanging h_t / dz_t through tagged granules)
on the CPU interpreter, whose co-resident launches keep every block of such a grid alive at once: the exchange protocol
itself, its equality with the single-workgroup flavour, and the loud failure path (device status word, mst_plan_status)."""
import numpy as np
import pytest
import torch

import parity_cases as pc
from tools.synth import synth_clip
from simutil import sim_native
from style import _native as nat


def lstm_steps(plan, backward=False):
    n = plan.lib.mst_plan_step_count(plan.handle, 7, int(backward))
    info = np.zeros((n, 8), np.int32)
    assert plan.lib.mst_plan_step_info(plan.handle, 7, int(backward), info.ctypes.data) == n
    return [tuple(r) for r in info.tolist() if r[2] == 192]          # {B, S, H, multi, count, kind, level, chain}


def run(plan, flat, clip, K=1):
    params = flat.clone()
    g = torch.zeros_like(params)
    losses = torch.zeros(K, nat.N_LOSSES)
    for k in range(K):
        plan.set_inputs(mode=clip['mode'], bpm=clip['bpm'], instr=clip['instruments_features'], used=clip['used_instruments'],
                        bpm_target=float(clip['bpm_int']), clip=k)
    xp = torch.cat([clip['pitched']] * K).contiguous()
    xu = torch.cat([clip['unpitched']] * K).contiguous()
    pc.poison(plan)
    plan.train_iteration(params, g, xp, xu, losses)
    return g, losses


def test_multi_and_single_workgroup_flavours_are_bit_identical():
    native = sim_native()
    C, R, T = 1, 4, 1                                   # four bars: three exchanged steps per direction
    dims = pc.make_dims(pc.FULL, C, R, T, True)
    flat, _, _ = pc.random_params(native, dims)
    clip = synth_clip(3, C, R, T, True, density=0.05)
    multi = nat.Plan(native, dims, 'cpu')
    single = nat.Plan(native, dims, 'cpu', lstm_flavour=1)
    assert [s[3] for s in lstm_steps(multi)] == [1] and [s[3] for s in lstm_steps(multi, True)] == [1]
    assert [s[3] for s in lstm_steps(single)] == [0] and [s[3] for s in lstm_steps(single, True)] == [0]
    g1, l1 = run(multi, flat, clip)
    g0, l0 = run(single, flat, clip)
    assert multi.status() == 0 and single.status() == 0
    assert torch.isfinite(l1[0, 0]) and torch.equal(l1.nan_to_num(-1.), l0.nan_to_num(-1.))
    assert torch.equal(multi.view('style'), single.view('style'))
    assert torch.equal(g1, g0)


def test_batched_plan_keeps_the_multi_flavour_within_the_resident_slots():
    native = sim_native()
    # the interpreter models 16 CUs x (4 - 1) workgroup slots = 48 co-resident workgroups = 4 clips of 12
    dims = pc.make_dims(pc.FULL, 1, 2, 1, True, clips=4)
    assert [s[3] for s in lstm_steps(nat.Plan(native, dims, 'cpu'))] == [1]
    dims = pc.make_dims(pc.FULL, 1, 2, 1, True, clips=5)
    assert [s[3] for s in lstm_steps(nat.Plan(native, dims, 'cpu'))] == [0]


def test_two_clips_on_the_multi_flavour_match_one_clip_plans():
    native = sim_native()
    C, R, T = 1, 3, 1
    dims1, dims2 = pc.make_dims(pc.FULL, C, R, T, True), pc.make_dims(pc.FULL, C, R, T, True, clips=2)
    flat, _, _ = pc.random_params(native, dims1)
    clip = synth_clip(4, C, R, T, True, density=0.05)
    two = nat.Plan(native, dims2, 'cpu', gemm_tile=32)
    one = nat.Plan(native, dims1, 'cpu', gemm_tile=32)
    assert [s[3] for s in lstm_steps(two)] == [1]
    g2, l2 = run(two, flat, clip, K=2)
    g1, l1 = run(one, flat, clip)
    assert torch.equal(l2[0].nan_to_num(-1.), l1[0].nan_to_num(-1.)) and torch.equal(l2[1].nan_to_num(-1.), l1[0].nan_to_num(-1.))
    assert pc.rel(g2.numpy(), 2 * g1.numpy()) < 2e-6


def test_exchange_fault_sets_the_status_word_and_poisons_the_results():
    native = sim_native()
    C, R, T = 1, 3, 1
    dims = pc.make_dims(pc.FULL, C, R, T, True)
    flat, _, _ = pc.random_params(native, dims)
    clip = synth_clip(3, C, R, T, True, density=0.05)
    plan = nat.Plan(native, dims, 'cpu', lstm_flavour=2)          # workgroup 0 publishes its first step under a wrong epoch
    assert [s[3] for s in lstm_steps(plan)] == [2]
    g, losses = run(plan, flat, clip)
    assert torch.isnan(losses[0, 0])                              # nothing computed after the failure looks healthy
    assert plan.status(clear=False) == nat.DEV_LSTM_TIMEOUT       # sticky ...
    with pytest.raises(nat.MstError, match='MST_DEV_LSTM_TIMEOUT'):
        plan.check_status()                                       # ... until read with clear (check_status clears)
    assert plan.status() == 0
    # the same workspace and parameters on healthy kernels afterwards: finite again
    ok = nat.Plan(native, dims, 'cpu')
    g, losses = run(ok, flat, clip)
    assert ok.status() == 0 and torch.isfinite(losses[0, 0]) and torch.isfinite(g).all()


def test_bad_flavour_is_refused():
    native = sim_native()
    with pytest.raises(nat.MstError):
        nat.Plan(native, pc.make_dims(pc.SMALL, 1, 1, 1, False), 'cpu', lstm_flavour=3)
