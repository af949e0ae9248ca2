"""Pins oracle/style_oracle.py against fixtures produced by the reference itself
(tests/golden/make_golden.py). CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import style_oracle as so
from tools.synth import synth_clip

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
TOL = 2e-5      # rel-L2; oracle and reference run the same torch ops, so this is tight


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def load_small(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    flat = {k[3:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith('p0/')}
    return z, flat


@pytest.mark.parametrize('name', ['small_unpitched', 'small_pitched_only'])
def test_small_forward_loss_grads_adam(name):
    z, flat = load_small(name)
    C, R, T = (int(v) for v in z['crt'])
    unp = bool(z['unpitched'])
    dens = float(z['density'])
    mids = {}
    clip0 = synth_clip(0, C, R, T, unp, density=dens)
    (info, xp, xu), losses = so.iteration(flat, clip0, mids=mids)
    assert rel(mids['pitched_beats'].detach(), z['mid/pitched_channels_encoder/0']) < TOL
    assert rel(mids['pitched_bars'].detach(), z['mid/pitched_channels_encoder/1']) < TOL
    assert rel(mids['pitched_rhythm'].detach(), z['mid/pitched_rhythm_encoder/0']) < TOL
    assert rel(mids['style'].detach(), z['mid/style_encoder/0']) < TOL
    assert rel(mids['melody'].detach(), z['mid/melody_encoder/0']) < TOL
    if unp:
        assert rel(mids['unpitched_beats'].detach(), z['mid/unpitched_channels_encoder/0']) < TOL
        assert rel(mids['unpitched_bars'].detach(), z['mid/unpitched_channels_encoder/1']) < TOL
        assert rel(mids['unpitched_rhythm'].detach(), z['mid/unpitched_rhythm_encoder/0']) < TOL
        assert rel(xu.detach(), z['out/unpitched']) < TOL
    assert rel(info[0].detach(), z['out/instruments']) < TOL
    assert rel(info[1].detach(), z['out/mode']) < TOL
    assert rel(info[2].detach(), z['out/bpm']) < TOL
    assert rel(xp.detach(), z['out/pitched']) < TOL
    for k, v in losses.items():
        assert abs(v - float(z['loss0/' + k])) < 1e-5 * max(1, abs(v)), k
    assert set('loss0/' + k for k in losses) == set(k for k in z.files if k.startswith('loss0/'))
    worst = 0
    for n, p in flat.items():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        ref = z['g0/' + n]
        if np.linalg.norm(ref) == 0:
            assert float(g.abs().max()) < 1e-7, n
            continue
        worst = max(worst, rel(g, ref))
        assert rel(g, ref) < 2e-4, n
    # second clip accumulates, then one Adam step (train-model.py:151-154)
    opt = so.Adam(flat.values())
    clip1 = synth_clip(1, C, R, T, unp, density=dens)
    _, losses1 = so.iteration(flat, clip1)
    for k, v in losses1.items():
        assert abs(v - float(z['loss1/' + k])) < 1e-5 * max(1, abs(v)), k
    opt.step()
    for n, p in flat.items():
        assert np.abs(p.detach().numpy() - z['p1/' + n]).max() < 2e-4, n


def test_fast_lstm_matches_explicit():
    z, flat = load_small('small_unpitched')
    C, R, T = (int(v) for v in z['crt'])
    clip = synth_clip(0, C, R, T, True, density=float(z['density']))
    with torch.no_grad():
        a = so.forward(flat, clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'])
        b = so.forward(flat, clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'], fast=True)
    assert rel(a[1], b[1]) < 1e-5 and rel(a[2], b[2]) < 1e-5


def test_hard_output_against_reference_fixture():
    """style/model.py:818-832 on a hand-made tensor (threshold velocities, accidental ties, maxima around .1) and on the
    style-swap predictions, including the in-place mutation of the input's velocities — outputs of the reference itself."""
    z = np.load(os.path.join(GOLDEN, 'inference_small.npz'))
    for pre in ('hard/x', 'hard/u'):
        x = torch.from_numpy(z[pre + '_in']).clone()
        y = so.hard_output(x)
        assert np.array_equal(y.numpy(), z[pre + '_out']) and np.array_equal(x.numpy(), z[pre + '_after']), pre
    for name in ('pitched', 'unpitched'):
        x = torch.from_numpy(z['swap/' + name]).clone()
        y = so.hard_output(x)
        assert np.array_equal(y.numpy(), z['swap/hard_' + name]) and np.array_equal(x.numpy(), z[f'swap/{name}_after']), name


def test_style_swap_inference_against_reference_fixture():
    """style/style_transfer.py:41-54: style of song B (pitched only) applied to melody + rhythm of song A."""
    z = np.load(os.path.join(GOLDEN, 'inference_small.npz'))
    _, flat = load_small('small_unpitched')          # the same seed-7 parameters (asserted by make_golden.py)
    C, R, T = (int(v) for v in z['crt'])
    a = synth_clip(0, C, R, T, True, density=float(z['density']))
    b = synth_clip(1, C, R, T, True, density=float(z['density']))
    with torch.no_grad():
        style_a, melody_a, rhythm_a = so.extract_style(flat, a['mode'], a['bpm'], a['pitched'], a['instruments_features'], a['unpitched'])
        style_b, _, _ = so.extract_style(flat, b['mode'], b['bpm'], b['pitched'], b['instruments_features'], None)
        P = so.Params(flat)
        ip, mp, bp = so.song_info(P.sub('song_info_model'), style_b, rhythm_a)
        psa = P.sub('pitched_style_applier')
        xp = so.pitched_style_applier(psa, style_b, melody_a, rhythm_a, b['instruments_features'][:, :1])
        xu = so.unpitched_style_applier(P.sub('unpitched_style_applier'), style_b, rhythm_a)
        xp_all = so.pitched_style_applier(psa, style_b, melody_a, rhythm_a, b['instruments_features'])
    for got, key in ((style_a, 'style_a'), (style_b, 'style_b'), (melody_a, 'melody_a'), (rhythm_a, 'rhythm_a'), (ip, 'instruments'),
                     (mp, 'mode'), (bp, 'bpm'), (xp, 'pitched'), (xu, 'unpitched'), (xp_all, 'pitched_all_channels')):
        assert got.shape == z['swap/' + key].shape, key
        assert rel(got, z['swap/' + key]) < TOL, key


def test_two_clip_gradient_sum_against_reference_fixture():
    z = np.load(os.path.join(GOLDEN, 'inference_small.npz'))
    _, flat = load_small('small_unpitched')
    C, R, T = (int(v) for v in z['crt'])
    for k in (0, 1):
        so.iteration(flat, synth_clip(k, C, R, T, True, density=float(z['density'])))
    for n, p in flat.items():
        ref = z['g01/' + n]
        if np.linalg.norm(ref) > 0:
            assert rel(p.grad, ref) < 2e-4, n


def test_total_loss_normalize_false_and_true_against_reference_fixture():
    z = np.load(os.path.join(GOLDEN, 'loss_normalize.npz'))
    C, R, T = (int(v) for v in z['crt'])
    clip = synth_clip(int(z['clip_id']), C, R, T, True, density=float(z['density']))
    for normalize in (0, 1):
        for unp in (1, 0):
            tag = f'n{normalize}u{unp}'
            leaves = {k: torch.tensor(z['in/' + k]).requires_grad_(True) for k in ('pitched_pred', 'unpitched_pred', 'instruments', 'mode', 'bpm')}
            out = so.total_loss(leaves['instruments'], clip['used_instruments'], leaves['bpm'], clip['bpm_int'], leaves['mode'], clip['mode'],
                                leaves['pitched_pred'], clip['pitched'], leaves['unpitched_pred'] if unp else None,
                                clip['unpitched'] if unp else None, normalize=bool(normalize))
            keys = [k[len(tag) + 6:] for k in z.files if k.startswith(tag + '/loss/')]
            assert sorted(keys) == sorted(out), tag
            for k in keys:
                assert abs(float(out[k]) - float(z[f'{tag}/loss/{k}'])) < 1e-5, (tag, k)
            out['total'].backward()
            for name, t in leaves.items():
                if f'{tag}/grad/{name}' in z.files:
                    assert rel(t.grad, z[f'{tag}/grad/{name}']) < 1e-5, (tag, name)
                else:
                    assert t.grad is None, (tag, name)


def test_hard_output():
    x = torch.rand(1, 2, 2, 2, 10, 56, 5)
    y = so.hard_output(x.clone())
    assert ((y[..., 1] == 0) | (y[..., 1] > .01)).all()
    assert set(np.unique(y[..., 2:].numpy())) <= {0.0, 1.0}
    xu = torch.rand(1, 1, 2, 2, 10, 47, 2)
    assert so.hard_output(xu).shape == xu.shape
