"""-m gpu: the reference's Python surface (style.model) on an MI355X against fixtures produced by the
reference itself: forward/loss/backward through autograd, torch.optim.Adam and the fused Adam, the
three stage methods, the seed-108 full-width model, and a 6-iteration / 3-optimizer-step trajectory."""
import os

import numpy as np
import pytest
import torch

from oracle import style_oracle as so
from tools.synth import synth_clip
from simutil import GOLDEN, rel
from test_host_surface import FULL, SMALL, build_model

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def to_dev(clip):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in clip.items()}


def reference_call(model, clip):
    """train-model.py:113-123, verbatim argument order (bpm before mode)."""
    import style.model as m
    (ip, mp, bp), xp, xu = model(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'])
    losses = m.get_total_loss(ip, clip['used_instruments'], bp, clip['bpm_int'], mp, clip['mode'], xp, clip['pitched'],
                              xu, clip['unpitched'], normalize=True)
    return (ip, mp, bp, xp, xu), losses


def flat(d, prefix=''):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if isinstance(v, dict):
            out.update(flat(v, prefix + k + '_'))
        else:
            out[prefix + k] = float(v)
    return out


def load_small(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    model = build_model(SMALL)
    model.load_state_dict({k: torch.from_numpy(z['p0/' + k]) for k in model.state_dict()})
    return z, model.to(DEV)


@pytest.mark.parametrize('name,fused', [('small_unpitched', False), ('small_unpitched', True), ('small_pitched_only', False)])
def test_train_loop_body_matches_reference(name, fused):
    from style.optim import FusedAdam
    z, model = load_small(name)
    C, R, T = (int(v) for v in z['crt'])
    unp = bool(z['unpitched'])
    opt = FusedAdam(model) if fused else torch.optim.Adam(model.parameters(), lr=.01)
    clip = to_dev(synth_clip(0, C, R, T, unp, density=float(z['density'])))
    (ip, mp, bp, xp, xu), losses = reference_call(model, clip)
    assert rel(xp.detach().cpu(), z['out/pitched']) < 1e-4 and rel(ip.detach().cpu(), z['out/instruments']) < 1e-4
    assert rel(mp.detach().cpu(), z['out/mode']) < 1e-4 and rel(bp.detach().cpu(), z['out/bpm']) < 1e-4
    if unp:
        assert rel(xu.detach().cpu(), z['out/unpitched']) < 1e-4
    else:
        assert xu is None and losses['channels_loss']['unpitched'] is None
    assert losses['total'].shape == (1,)
    fl = flat(losses)
    assert set('loss0/' + k for k in fl) == set(k for k in z.files if k.startswith('loss0/'))
    for k, v in fl.items():
        assert abs(v - float(z['loss0/' + k])) < 2e-5, k
    losses['total'].backward()
    for n, p in model.named_parameters():
        ref = z['g0/' + n]
        got = p.grad.cpu().numpy()
        if np.linalg.norm(ref) < 1e-12:
            assert np.abs(got).max() < 1e-6, n
        else:
            assert rel(got, ref) < 5e-4, n
    clip1 = to_dev(synth_clip(1, C, R, T, unp, density=float(z['density'])))
    _, losses1 = reference_call(model, clip1)
    assert abs(float(losses1['total']) - float(z['loss1/total'])) < 2e-5
    losses1['total'].backward()                 # accumulates (sum) into the same p.grad
    opt.step()
    for n, p in model.named_parameters():
        assert np.abs(p.detach().cpu().numpy() - z['p1/' + n]).max() < 3e-4, n


def test_stage_methods_compose_to_forward_and_backprop():
    z, model = load_small('small_unpitched')
    C, R, T = (int(v) for v in z['crt'])
    clip = to_dev(synth_clip(0, C, R, T, True, density=float(z['density'])))
    style, melody, rhythm = model.extract_style(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'],
                                                clip['unpitched'])
    assert rel(style.detach().cpu(), z['mid/style_encoder/0']) < 1e-4
    assert rel(melody.detach().cpu(), z['mid/melody_encoder/0']) < 1e-4
    ip, mp, bp = model.predict_song_info(style, rhythm)
    xp, xu = model.apply_style(style, melody, rhythm, clip['instruments_features'], unpitched=True)
    assert rel(xp.detach().cpu(), z['out/pitched']) < 1e-4 and rel(xu.detach().cpu(), z['out/unpitched']) < 1e-4
    import style.model as m
    losses = m.get_total_loss(ip, clip['used_instruments'], bp, clip['bpm_int'], mp, clip['mode'], xp, clip['pitched'],
                              xu, clip['unpitched'], normalize=True)
    losses['total'].backward()
    for n, p in model.named_parameters():
        ref = z['g0/' + n]
        if np.linalg.norm(ref) > 1e-12:
            assert rel(p.grad.cpu().numpy(), ref) < 5e-4, n
    with torch.no_grad():      # inference use (style_transfer.py:67-74,101-131)
        s2, m2, r2 = model.extract_style(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], None)
        xp2, xu2 = model.apply_style(s2, m2, r2, clip['instruments_features'][:, :1], unpitched=False)
    assert xu2 is None and xp2.shape == (1, 1, R, T, 10, 56, 5) and not xp2.requires_grad


def test_differentiating_another_loss_leaf():
    z, model = load_small('small_unpitched')
    C, R, T = (int(v) for v in z['crt'])
    clipc = synth_clip(0, C, R, T, True, density=float(z['density']))
    clip = to_dev(clipc)
    _, losses = reference_call(model, clip)
    losses['channels_loss']['pitched']['velocity_loss'].backward()
    named = {k[3:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith('p0/')}
    info, xp, xu = so.forward(named, clipc['mode'], clipc['bpm'], clipc['pitched'], clipc['instruments_features'], clipc['unpitched'])
    ref = so.total_loss(info[0], clipc['used_instruments'], info[2], clipc['bpm_int'], info[1], clipc['mode'], xp,
                        clipc['pitched'], xu, clipc['unpitched'])
    ref['channels_loss_pitched_velocity_loss'].backward()
    n = 'pitched_style_applier.linear.weight'
    assert rel(dict(model.named_parameters())[n].grad.cpu().numpy(), named[n].grad.numpy()) < 5e-4


def test_full_width_seed108_model():
    z = np.load(os.path.join(GOLDEN, 'full_seed108.npz'))
    C, R, T = (int(v) for v in z['crt'])
    model = build_model(FULL, seed=108).to(DEV)
    clip = to_dev(synth_clip(3, C, R, T, True))
    (ip, mp, bp, xp, xu), losses = reference_call(model, clip)
    assert rel(ip.detach().cpu(), z['out/instruments']) < 1e-4 and rel(bp.detach().cpu(), z['out/bpm']) < 1e-4
    assert rel(xp.detach().cpu()[0, 1, 1, 2], z['slice/pitched']) < 1e-4
    assert rel(xu.detach().cpu()[0, 0, 1, 2], z['slice/unpitched']) < 1e-4
    for k, v in flat(losses).items():
        assert abs(v - float(z['loss0/' + k])) < 2e-5, k
    losses['total'].backward()
    for n, p in model.named_parameters():
        g = p.grad.double().reshape(-1).cpu()
        ref = z['gf/' + n]                          # [sum, abs-sum, sq-sum, first, last] from the reference
        if ref[1] < 1e-9:
            continue
        assert abs(float(g.abs().sum()) - ref[1]) < 5e-4 * ref[1] + 1e-9, n
        assert abs(float((g * g).sum()) - ref[2]) < 1e-3 * ref[2] + 1e-12, n


def test_trajectory_six_iterations_three_adam_steps():
    """train-model.py loop on the bench clip shape with seed-108 weights: loss leaves of every iteration
    against the reference's own run (tests/golden/trajectory_seed108.npz)."""
    from style.optim import FusedAdam
    z = np.load(os.path.join(GOLDEN, 'trajectory_seed108.npz'))
    C, R, T = (int(v) for v in z['crt'])
    keys = [str(k) for k in z['loss_keys']]
    model = build_model(FULL, seed=108).to(DEV)
    opt = FusedAdam(model)
    for it in range(6):
        clip = to_dev(synth_clip(it, C, R, T, True))
        _, losses = reference_call(model, clip)
        losses['total'].backward()
        fl = flat(losses)
        for j, k in enumerate(keys):
            assert abs(fl[k] - z['losses'][it, j]) < 3e-4, (it, k, fl[k], z['losses'][it, j])
        if (it + 1) % 2 == 0:
            opt.step()
    for n, p in model.named_parameters():
        x = p.detach().double().reshape(-1).cpu()
        ref = z['pf/' + n]
        assert abs(float(x.abs().sum()) - ref[1]) < 2e-3 * ref[1] + 1e-6, n


def test_two_pending_forwards_of_one_shape_keep_their_own_activations():
    """Two grad-enabled forwards with the same (C, R, T) before either backward — the summed loss of two clips.
    Each forward must own its workspace (a shared one lets the second overwrite the first's saved activations);
    expected: the reference's g(clip 0) + g(clip 1) (tests/golden/inference_small.npz)."""
    z, model = load_small('small_unpitched')
    zi = np.load(os.path.join(GOLDEN, 'inference_small.npz'))
    C, R, T = (int(v) for v in z['crt'])
    clips = [to_dev(synth_clip(k, C, R, T, True, density=float(z['density']))) for k in (0, 1)]
    _, l0 = reference_call(model, clips[0])
    _, l1 = reference_call(model, clips[1])
    assert abs(float(l0['total']) - float(z['loss0/total'])) < 2e-5 and abs(float(l1['total']) - float(z['loss1/total'])) < 2e-5
    (l0['total'] + l1['total']).backward()
    for n, p in model.named_parameters():
        ref = zi['g01/' + n]
        if np.linalg.norm(ref) > 1e-12:
            assert rel(p.grad.cpu().numpy(), ref) < 5e-4, n
    # the workspaces went back to the pool; a stage cannot be back-propagated twice
    style, melody, rhythm = model.extract_style(clips[0]['mode'], clips[0]['bpm'], clips[0]['pitched'],
                                                clips[0]['instruments_features'], clips[0]['unpitched'])
    style.sum().backward(retain_graph=True)
    from style import _native
    with pytest.raises(_native.MstError):
        style.sum().backward()


def test_fused_train_iteration_eager_then_graph_replay():
    """StyleTransferModel.train_iteration: the loop body as one C-ABI call.  First use of a clip shape runs eagerly, the second
    captures a hipGraph, later ones replay it: same losses, gradients accumulate exactly like three backward() calls."""
    z, model = load_small('small_unpitched')
    C, R, T = (int(v) for v in z['crt'])
    clip = to_dev(synth_clip(0, C, R, T, True, density=float(z['density'])))
    packed = [model.train_iteration(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'],
                                    clip['used_instruments'], clip['bpm_int']).clone() for _ in range(3)]
    torch.cuda.synchronize()
    assert abs(float(packed[0][0]) - float(z['loss0/total'])) < 2e-5
    assert torch.equal(packed[0].nan_to_num(-1.), packed[1].nan_to_num(-1.)) and torch.equal(packed[1].nan_to_num(-1.), packed[2].nan_to_num(-1.))
    for n, p in model.named_parameters():
        ref = 3.0 * z['g0/' + n]
        if np.linalg.norm(ref) > 1e-12:
            assert rel(p.grad.cpu().numpy(), ref) < 5e-4, n
    # a different clip of the same shape through the replayed graph
    model.zero_grad()
    clip1 = to_dev(synth_clip(1, C, R, T, True, density=float(z['density'])))
    l1 = model.train_iteration(clip1['mode'], clip1['bpm'], clip1['pitched'], clip1['instruments_features'], clip1['unpitched'],
                               clip1['used_instruments'], clip1['bpm_int'])
    assert abs(float(l1[0]) - float(z['loss1/total'])) < 2e-5


def test_sub_modules_called_on_their_own():
    """The reference's sub-modules are callable (style/model.py:77-99,128-141,557-562,624-675,703-724); here the channel
    encoders, SongInfoModel and the two appliers run through the owning model's plan, against the reference's fixture."""
    import io
    z, model = load_small('small_unpitched')
    C, R, T = (int(v) for v in z['crt'])
    clip = to_dev(synth_clip(0, C, R, T, True, density=float(z['density'])))
    beats, bars = model.pitched_channels_encoder(clip['pitched'], clip['instruments_features'])
    assert rel(beats.cpu(), z['mid/pitched_channels_encoder/0']) < 1e-4 and rel(bars.cpu(), z['mid/pitched_channels_encoder/1']) < 1e-4
    ub, ubars = model.unpitched_channels_encoder(clip['unpitched'])
    assert rel(ub.cpu(), z['mid/unpitched_channels_encoder/0']) < 1e-4 and rel(ubars.cpu(), z['mid/unpitched_channels_encoder/1']) < 1e-4
    with torch.no_grad():
        style, melody, rhythm = model.extract_style(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'])
        ip, mp, bp = model.song_info_model(style, rhythm)
        xp = model.pitched_style_applier(style, melody, rhythm, clip['instruments_features'])
        xu = model.unpitched_style_applier(style, rhythm)
    assert rel(ip.cpu(), z['out/instruments']) < 1e-4 and rel(xp.cpu(), z['out/pitched']) < 1e-4 and rel(xu.cpu(), z['out/unpitched']) < 1e-4
    import style.model as m
    with pytest.raises(NotImplementedError):
        model.melody_encoder(beats, bars, clip['pitched'], clip['instruments_features'])
    with pytest.raises(NotImplementedError):                     # not part of a model: only a parameter container
        m.SongInfoModel(3, 12, 6, 41)(style, rhythm)
    buf = io.BytesIO()
    torch.save(model, buf)                                        # whole-module snapshot keeps the sub-modules callable
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    assert rel(again.song_info_model(style, rhythm)[0].detach().cpu(), z['out/instruments']) < 1e-4


def test_style_swap_inference_matches_reference_fixture():
    """style/style_transfer.py:41-54,101-131 through the product surface: style of song B (pitched only,
    unpitched_channels=None) on melody + rhythm of song A, then hard_output — against outputs of the reference."""
    import style.model as m
    z, model = load_small('small_unpitched')
    zi = np.load(os.path.join(GOLDEN, 'inference_small.npz'))
    C, R, T = (int(v) for v in z['crt'])
    a, b = (to_dev(synth_clip(k, C, R, T, True, density=float(z['density']))) for k in (0, 1))
    with torch.no_grad():
        style_a, melody_a, rhythm_a = model.extract_style(a['mode'], a['bpm'], a['pitched'], a['instruments_features'], a['unpitched'])
        style_b, _, _ = model.extract_style(b['mode'], b['bpm'], b['pitched'], b['instruments_features'], None)
        ip, mp, bp = model.predict_song_info(style_b, rhythm_a)
        xp, xu = model.apply_style(style_b, melody_a, rhythm_a, b['instruments_features'][:, :1], True)
        xp_all, none = model.apply_style(style_b, melody_a, rhythm_a, b['instruments_features'], False)
    assert none is None
    for got, key in ((style_a, 'style_a'), (style_b, 'style_b'), (melody_a, 'melody_a'), (rhythm_a, 'rhythm_a'), (ip, 'instruments'),
                     (mp, 'mode'), (bp, 'bpm'), (xp, 'pitched'), (xu, 'unpitched'), (xp_all, 'pitched_all_channels')):
        assert tuple(got.shape) == zi['swap/' + key].shape, key
        assert rel(got.cpu(), zi['swap/' + key]) < 1e-4, key
    # hard_output on the REFERENCE's predictions (bit-exact decisions need bit-identical inputs)
    for name in ('pitched', 'unpitched'):
        x = torch.from_numpy(zi['swap/' + name]).to(DEV)
        y = m.hard_output(x)
        assert np.array_equal(y.cpu().numpy(), zi['swap/hard_' + name]), name
        assert np.array_equal(x.cpu().numpy(), zi[f'swap/{name}_after']), name      # velocities zeroed in place


def test_hard_output_matches_reference_fixture():
    """style/model.py:818-832 on the hand-made tensor of inference_small.npz: velocities around .01, tied accidental
    maxima (both stay 1), maxima at or below .1 (all zeros) — bit-exact, including the in-place mutation."""
    import style.model as m
    zi = np.load(os.path.join(GOLDEN, 'inference_small.npz'))
    for pre in ('hard/x', 'hard/u'):
        x = torch.from_numpy(zi[pre + '_in']).to(DEV)
        y = m.hard_output(x)
        assert np.array_equal(y.cpu().numpy(), zi[pre + '_out']), pre
        assert np.array_equal(x.cpu().numpy(), zi[pre + '_after']), pre


def test_unsupported_widths_fail_in_the_constructor():
    from style import _native
    with pytest.raises(_native.MstError):
        build_model(dict(FULL, melody=6))


def test_hard_output_matches_oracle_and_mutates_input():
    import style.model as m
    x = torch.rand(1, 2, 2, 3, 10, 56, 5)
    x[..., 1] *= (torch.rand(x.shape[:-1]) < .5) * 0.03 + (torch.rand(x.shape[:-1]) < .3)
    ref = so.hard_output(x.clone())
    xd = x.to(DEV)
    out = m.hard_output(xd)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(xd[..., 1].cpu(), ref[..., 1])       # velocities zeroed in place, like the reference
    xu = torch.rand(1, 1, 2, 3, 10, 47, 2)
    assert torch.equal(m.hard_output(xu.to(DEV)).cpu(), so.hard_output(xu.clone()))


def test_replaced_parameter_objects_are_picked_up():
    """load_state_dict(assign=True) and `module.weight = nn.Parameter(...)` install NEW Parameter objects while the old ones
    keep aliasing the flat buffer the kernels read: the next forward must run on the new weights."""
    z, model = load_small('small_unpitched')
    C, R, T = (int(v) for v in z['crt'])
    clip = to_dev(synth_clip(0, C, R, T, True, density=float(z['density'])))
    with torch.no_grad():
        (ip, _, _), xp, _ = model(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'])
    assert rel(xp.cpu(), z['out/pitched']) < 1e-4
    # (a) every parameter replaced through load_state_dict(assign=True): the post-Adam parameters of the fixture
    sd = {k: torch.from_numpy(z['p1/' + k]).to(DEV) for k in model.state_dict()}
    model.load_state_dict(sd, assign=True)
    other = build_model(SMALL)
    other.load_state_dict({k: torch.from_numpy(z['p1/' + k]) for k in other.state_dict()})
    other = other.to(DEV)
    with torch.no_grad():
        (ip1, _, _), xp1, _ = model(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'])
        (ip2, _, _), xp2, _ = other(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'])
    assert torch.equal(xp1, xp2) and torch.equal(ip1, ip2) and not torch.equal(xp1, xp)
    # (b) ONE parameter in the middle of the tree replaced by attribute assignment
    lin = model.song_info_model.instruments_linear
    lin.bias = torch.nn.Parameter(lin.bias.detach() + 3.)
    with torch.no_grad():
        (ip3, _, _), xp3, _ = model(clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'])
    assert torch.allclose(ip3, ip1 + 3., atol=1e-5) and torch.equal(xp3, xp1)
    assert all(p.data_ptr() == model._flat.data_ptr() + 4 * off for p, off in zip(model.parameters(), model._offsets))
