"""The product's HIP kernels, executed by the hipsim CPU interpreter, against the golden
fixtures produced by the reference (forward, 15 loss leaves, every parameter gradient, Adam).
CPU-only stand-in for the -m gpu parity tests: same sources, same C ABI, no GPU."""
import os

import numpy as np
import pytest
import torch

from oracle.synth import synth_clip
from simutil import GOLDEN, flat_from_named, make_dims, rel, sim_native
from style import _native as nat

SMALL = dict(beat=8, bar=6, nrf=3, style=12, melody=4, rhythm=6)
TOL = 1e-4      # rel-L2, the north_star tolerance


def setup(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    C, R, T = (int(v) for v in z['crt'])
    unp = bool(z['unpitched'])
    native = sim_native()
    dims = make_dims(SMALL, C, R, T, unp)
    params, table = flat_from_named(native, dims, {k[3:]: z[k] for k in z.files if k.startswith('p0/')})
    plan = nat.Plan(native, dims, 'cpu')
    clip = synth_clip(0, C, R, T, unp, density=float(z['density']))
    plan.set_inputs(mode=clip['mode'], bpm=clip['bpm'], instr=clip['instruments_features'],
                    used=clip['used_instruments'], bpm_target=float(clip['bpm_int']))
    return z, native, dims, params, table, plan, clip


@pytest.mark.parametrize('name', ['small_unpitched', 'small_pitched_only'])
def test_forward_loss_backward(name):
    z, native, dims, params, table, plan, clip = setup(name)
    unp = clip['unpitched'] is not None
    gparams = torch.zeros_like(params)
    losses = torch.zeros(nat.N_LOSSES)
    plan.train_iteration(params, gparams, clip['pitched'].contiguous(), clip['unpitched'], losses)
    checks = [('pitched_beats', 'mid/pitched_channels_encoder/0'), ('pitched_bars', 'mid/pitched_channels_encoder/1'),
              ('pitched_rhythm', 'mid/pitched_rhythm_encoder/0'), ('style', 'mid/style_encoder/0'),
              ('melody', 'mid/melody_encoder/0'), ('instruments_pred', 'out/instruments'), ('mode_pred', 'out/mode'),
              ('bpm_pred', 'out/bpm'), ('pitched_pred', 'out/pitched')]
    if unp:
        checks += [('unpitched_beats', 'mid/unpitched_channels_encoder/0'), ('unpitched_bars', 'mid/unpitched_channels_encoder/1'),
                   ('unpitched_rhythm', 'mid/unpitched_rhythm_encoder/0'), ('unpitched_pred', 'out/unpitched')]
    for slot, key in checks:
        assert rel(plan.view(slot).numpy(), z[key]) < TOL, (slot, rel(plan.view(slot).numpy(), z[key]))
    for i, k in enumerate(nat.LOSS_KEYS):
        if 'loss0/' + k in z.files:
            assert abs(float(losses[i]) - float(z['loss0/' + k])) < 2e-5, (k, float(losses[i]), float(z['loss0/' + k]))
        else:
            assert np.isnan(float(losses[i])), k
    bad = []
    for pname, off, shape in table:
        ref = z['g0/' + pname].reshape(-1)
        got = gparams[off:off + ref.size].numpy()
        if np.linalg.norm(ref) < 1e-12:
            if np.abs(got).max() > 1e-6:
                bad.append((pname, 'nonzero', float(np.abs(got).max())))
        elif rel(got, ref) > 5e-4:
            bad.append((pname, rel(got, ref)))
    assert not bad, bad


def test_accumulate_and_adam():
    z, native, dims, params, table, plan, clip = setup('small_unpitched')
    gparams = torch.zeros_like(params)
    plan.train_iteration(params, gparams, clip['pitched'].contiguous(), clip['unpitched'])
    clip1 = synth_clip(1, dims.C, dims.R, dims.T, True, density=float(z['density']))
    plan.set_inputs(mode=clip1['mode'], bpm=clip1['bpm'], instr=clip1['instruments_features'],
                    used=clip1['used_instruments'], bpm_target=float(clip1['bpm_int']))
    losses = torch.zeros(nat.N_LOSSES)
    plan.train_iteration(params, gparams, clip1['pitched'].contiguous(), clip1['unpitched'], losses)
    assert abs(float(losses[0]) - float(z['loss1/total'])) < 2e-5
    m, v, state = torch.zeros_like(params), torch.zeros_like(params), torch.zeros(4)
    nat.check(native.lib.mst_adam_step(nat.ptr(params), nat.ptr(gparams), nat.ptr(m), nat.ptr(v), params.numel(),
                                       nat.ptr(state), .01, .9, .999, 1e-8, 200, .9, 1, None), 'adam')
    assert float(state[0]) == 1.0 and float(gparams.abs().max()) == 0.0
    for pname, off, shape in table:
        ref = z['p1/' + pname].reshape(-1)
        assert np.abs(params[off:off + ref.size].numpy() - ref).max() < 3e-4, pname
