"""Parity cases shared by the CPU-interpreter tests and the -m gpu tests: both drive the SAME
C ABI (include/mst_amd.h); only the library build (hipsim vs hipcc/gfx950) and the device differ."""
import os

import numpy as np
import torch

from oracle import style_oracle as so
from tools.synth import synth_clip
from simutil import GOLDEN, flat_from_named, make_dims, rel
from style import _native as nat

SMALL = dict(beat=8, bar=6, nrf=3, style=12, melody=4, rhythm=6)
FULL = dict(beat=64, bar=128, nrf=8, style=256, melody=8, rhythm=32)
TOL = 1e-4          # rel-L2 tolerance of north_star ("within 1e-4 rel-L2 of the CPU reference")


def poison(plan):
    """NaN-fill the gradient and scratch arenas of the plan's workspace: a backward pass must write every gradient /
    scratch element before it reads or accumulates into it (mst_zero_grads clears only what the first-writer analysis
    could not prove written), so nothing of the poison may reach a result."""
    goff = plan.clips * plan.clip_stride
    plan.ws[goff:].fill_(float('nan'))


def set_clip(plan, clip):
    plan.set_inputs(mode=clip['mode'], bpm=clip['bpm'], instr=clip['instruments_features'],
                    used=clip['used_instruments'], bpm_target=float(clip['bpm_int']))


def dev_clip(clip, device):
    return clip['pitched'].contiguous().to(device), (None if clip['unpitched'] is None else clip['unpitched'].contiguous().to(device))


def golden_small(native, device, name):
    """Fixture produced by the reference: forward mids, 15 loss leaves, every gradient, 2-clip Adam step."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    C, R, T = (int(v) for v in z['crt'])
    unp = bool(z['unpitched'])
    dims = make_dims(SMALL, C, R, T, unp)
    params, table = flat_from_named(native, dims, {k[3:]: z[k] for k in z.files if k.startswith('p0/')})
    params = params.to(device)
    plan = nat.Plan(native, dims, device)
    clip = synth_clip(0, C, R, T, unp, density=float(z['density']))
    set_clip(plan, clip)
    gparams = torch.zeros_like(params)
    losses = torch.zeros(nat.N_LOSSES, device=device)
    xp, xu = dev_clip(clip, device)
    poison(plan)
    plan.train_iteration(params, gparams, xp, xu, losses)
    checks = [('pitched_beats', 'mid/pitched_channels_encoder/0'), ('pitched_bars', 'mid/pitched_channels_encoder/1'),
              ('pitched_rhythm', 'mid/pitched_rhythm_encoder/0'), ('style', 'mid/style_encoder/0'),
              ('melody', 'mid/melody_encoder/0'), ('instruments_pred', 'out/instruments'), ('mode_pred', 'out/mode'),
              ('bpm_pred', 'out/bpm'), ('pitched_pred', 'out/pitched')]
    if unp:
        checks += [('unpitched_beats', 'mid/unpitched_channels_encoder/0'), ('unpitched_bars', 'mid/unpitched_channels_encoder/1'),
                   ('unpitched_rhythm', 'mid/unpitched_rhythm_encoder/0'), ('unpitched_pred', 'out/unpitched')]
    for slot, key in checks:
        e = rel(plan.view(slot).cpu().numpy(), z[key])
        assert e < TOL, (slot, e)
    lc = losses.cpu()
    for i, k in enumerate(nat.LOSS_KEYS):
        if 'loss0/' + k in z.files:
            assert abs(float(lc[i]) - float(z['loss0/' + k])) < 2e-5, (k, float(lc[i]), float(z['loss0/' + k]))
        else:
            assert np.isnan(float(lc[i])), k
    gc = gparams.cpu()
    bad = []
    for pname, off, shape in table:
        ref = z['g0/' + pname].reshape(-1)
        got = gc[off:off + ref.size].numpy()
        if np.linalg.norm(ref) < 1e-12:
            if np.abs(got).max() > 1e-6:
                bad.append((pname, 'nonzero', float(np.abs(got).max())))
        elif rel(got, ref) > 5e-4:
            bad.append((pname, rel(got, ref)))
    assert not bad, bad
    # second clip accumulates (sum), then Adam + StepLR + zero_grad (train-model.py:151-154)
    clip1 = synth_clip(1, C, R, T, unp, density=float(z['density']))
    set_clip(plan, clip1)
    xp, xu = dev_clip(clip1, device)
    plan.train_iteration(params, gparams, xp, xu, losses)
    assert abs(float(losses.cpu()[0]) - float(z['loss1/total'])) < 2e-5
    m, v, state = torch.zeros_like(params), torch.zeros_like(params), torch.zeros(4, device=device)
    nat.check(native.lib.mst_adam_step(nat.ptr(params), nat.ptr(gparams), nat.ptr(m), nat.ptr(v), params.numel(),
                                       nat.ptr(state), .01, .9, .999, 1e-8, 200, .9, 1, nat.current_stream(device)), 'adam')
    assert float(state.cpu()[0]) == 1.0 and float(gparams.abs().max()) == 0.0
    pc = params.cpu()
    for pname, off, shape in table:
        ref = z['p1/' + pname].reshape(-1)
        assert np.abs(pc[off:off + ref.size].numpy() - ref).max() < 3e-4, pname


def random_params(native, dims, seed=0):
    table = native.param_table(dims)
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(native.param_floats(dims))
    named = {}
    for name, off, shape in table:
        n = int(np.prod(shape))
        fan = shape[1] * (shape[2] if len(shape) > 2 else 1) if len(shape) > 1 else shape[0]
        flat[off:off + n] = (torch.rand(n, generator=g) * 2 - 1) / fan ** 0.5
        named[name] = flat[off:off + n].view(*shape).clone().requires_grad_(True)
    return flat, named, table


def oracle_case(native, device, widths, C, R, T, unp, seed=0, clip_id=5, density=0.02, check_bitwise=False, **plan_opts):
    """Same seeded inputs through the oracle (torch CPU autograd) and through the C ABI."""
    dims = make_dims(widths, C, R, T, unp)
    flat, named, table = random_params(native, dims, seed)
    clip = synth_clip(clip_id, C, R, T, unp, density=density)
    mids = {}
    (info, xp_ref, xu_ref), ref_losses = so.iteration(named, clip, mids=mids)
    plan = nat.Plan(native, dims, device, **plan_opts)
    set_clip(plan, clip)
    params = flat.to(device)
    gparams = torch.zeros_like(params)
    losses = torch.zeros(nat.N_LOSSES, device=device)
    xp, xu = dev_clip(clip, device)
    poison(plan)
    plan.train_iteration(params, gparams, xp, xu, losses)
    out = dict(style=mids['style'], melody=mids['melody'], rhythm=mids['rhythm'], pitched_pred=xp_ref,
               instruments_pred=info[0], mode_pred=info[1], bpm_pred=info[2])
    if unp:
        out['unpitched_pred'] = xu_ref
    for k, v in out.items():
        e = rel(plan.view(k).cpu().numpy(), v.detach().numpy())
        assert e < TOL, (k, e)
    lc = losses.cpu()
    for i, k in enumerate(nat.LOSS_KEYS):
        if k in ref_losses:
            assert abs(float(lc[i]) - ref_losses[k]) < 5e-5, (k, float(lc[i]), ref_losses[k])
    gref = torch.cat([(named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])).reshape(-1)
                      for n, _, _ in table])
    gc = gparams.cpu()
    e = rel(gc.numpy(), gref.numpy())
    assert e < TOL, ('all gradients', e)
    # per tensor: the north_star bar (1e-4 rel-L2) for every tensor that carries more than 1 % of the gradient norm; tensors
    # below that are dominated by fp32 cancellation in long sums (their absolute error is far under the bar of the whole
    # vector) and get a loose guard against structural mistakes only
    worst, worst_big = 0.0, 0.0
    for pname, off, shape in table:
        n = int(np.prod(shape))
        r = gref[off:off + n]
        if float(r.norm()) > 1e-7 * float(gref.norm()):
            err = rel(gc[off:off + n].numpy(), r.numpy())
            worst = max(worst, err)
            if float(r.norm()) > 1e-2 * float(gref.norm()):
                worst_big = max(worst_big, err)
                assert err < TOL, ('per-tensor gradient (> 1 % of the norm)', pname, err)
    assert worst < 2e-3, ('worst per-tensor gradient', worst)
    if check_bitwise:   # fixed summation order everywhere: a second run must be bit-identical
        g2 = torch.zeros_like(params)
        poison(plan)
        plan.train_iteration(params, g2, xp, xu, losses)
        assert torch.equal(g2, gparams)
    return e, worst


def batch_case(native, device, widths, C, R, T, unp, K, seed=0, density=0.03, check_oracle=True, gemm_tile=None, gemm_run=None, dense_flavour=None, branches=None):
    """K different clips in ONE plan (mst_dims.clips = K) against (a) the oracle run clip by clip with
    gradients accumulating like train-model.py:126 and (b) the product's own one-clip plan run K times:
    per-clip activations and losses are bit-identical; the summed gradient is equal to rounding (the order in
    which per-clip partial sums meet differs)."""
    dims1 = make_dims(widths, C, R, T, unp)
    dimsK = make_dims(widths, C, R, T, unp, clips=K)
    # a plan picks its GEMM tiling from the clip count (K >= 6: 64x64 tiles, else 32x32 split-K) unless `gemm_tile` forces
    # one; the bitwise comparison needs the one-clip plan on the tiling the batched plan chose
    planK = nat.Plan(native, dimsK, device, gemm_tile=gemm_tile, gemm_run=gemm_run, dense_flavour=dense_flavour, branches=branches)
    plan1 = nat.Plan(native, dims1, device, gemm_tile=planK.gemm_tile, gemm_run=gemm_run, dense_flavour=dense_flavour, branches=branches)
    flat, named, table = random_params(native, dims1, seed)
    clips = [synth_clip(10 + k, C, R, T, unp, density=density) for k in range(K)]
    params = flat.to(device)
    assert planK.clips == K
    for k, clip in enumerate(clips):
        planK.set_inputs(mode=clip['mode'], bpm=clip['bpm'], instr=clip['instruments_features'],
                         used=clip['used_instruments'], bpm_target=float(clip['bpm_int']), clip=k)
    xp = torch.cat([c['pitched'] for c in clips]).contiguous().to(device)
    xu = torch.cat([c['unpitched'] for c in clips]).contiguous().to(device) if unp else None
    gK = torch.zeros_like(params)
    lossesK = torch.zeros(K, nat.N_LOSSES, device=device)
    poison(planK)
    planK.train_iteration(params, gK, xp, xu, lossesK)
    # (b) one-clip plan, K sequential iterations
    g1 = torch.zeros_like(params)
    losses1 = torch.zeros(nat.N_LOSSES, device=device)
    names = ['style', 'melody', 'rhythm', 'pitched_pred', 'instruments_pred', 'mode_pred', 'bpm_pred'] + (['unpitched_pred'] if unp else [])
    for k, clip in enumerate(clips):
        set_clip(plan1, clip)
        a, b = dev_clip(clip, device)
        poison(plan1)
        plan1.train_iteration(params, g1, a, b, losses1)
        for name in names:
            assert torch.equal(planK.view(name, clip=k), plan1.view(name)), (name, k)
        assert torch.equal(lossesK[k].nan_to_num(-1.), losses1.nan_to_num(-1.)), k
    # same partial sums, other association: the deferred slab reduction adds (clip, split) rows in four contiguous
    # quarters, and batched plans cut their weight-gradient reductions into fewer, longer k-splits / slabs
    assert rel(gK.cpu().numpy(), g1.cpu().numpy()) < 2e-6
    if not check_oracle:
        return
    # (a) oracle, clip by clip, grads accumulate
    for k, clip in enumerate(clips):
        (info, xp_ref, xu_ref), ref_losses = so.iteration(named, clip)
        assert rel(planK.view('pitched_pred', clip=k).cpu().numpy(), xp_ref.detach().numpy()) < TOL, k
        assert rel(planK.view('instruments_pred', clip=k).cpu().numpy(), info[0].detach().numpy()) < TOL, k
        lc = lossesK[k].cpu()
        for i, key in enumerate(nat.LOSS_KEYS):
            if key in ref_losses:
                assert abs(float(lc[i]) - ref_losses[key]) < 5e-5, (k, key)
    gref = torch.cat([(named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])).reshape(-1)
                      for n, _, _ in table])
    e = rel(gK.cpu().numpy(), gref.numpy())
    assert e < TOL, ('summed gradients', e)


def loss_normalize_case(native, device):
    """get_total_loss on its own through mst_total_loss_fwd / _bwd against the reference-generated fixture
    tests/golden/loss_normalize.npz: normalize = False (the reference's default, style/model.py:937) and True, with and
    without the unpitched pair — every loss leaf and the gradient of `total` with respect to every prediction."""
    z = np.load(os.path.join(GOLDEN, 'loss_normalize.npz'))
    C, R, T = (int(v) for v in z['crt'])
    clip = synth_clip(int(z['clip_id']), C, R, T, True, density=float(z['density']))
    lib = native.lib
    f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).contiguous().to(device)
    pp, up, il, ml, bp = (f(z['in/' + k]) for k in ('pitched_pred', 'unpitched_pred', 'instruments', 'mode', 'bpm'))
    pt, ut, it, mt = f(clip['pitched']), f(clip['unpitched']), f(clip['used_instruments']), f(clip['mode'])
    bt = f([float(clip['bpm_int'])])
    P = nat.ptr
    stream = nat.current_stream(device)
    for normalize in (0, 1):
        for unp in (1, 0):
            tag = f'n{normalize}u{unp}'
            losses = torch.full((nat.N_LOSSES,), -7., device=device)
            saved = torch.zeros(nat.LOSS_SAVED, device=device)
            scratch = torch.full((lib.mst_loss_scratch_floats(),), float('nan'), device=device)
            n_p, n_u = pp.numel() // 5, (up.numel() // 2 if unp else 0)
            upx, utx = (up, ut) if unp else (None, None)
            nat.check(lib.mst_total_loss_fwd(P(pp), P(pt), n_p, P(upx), P(utx), n_u, P(il), P(it), il.numel(), P(ml), P(mt), P(bp),
                                             P(bt), normalize, P(losses), P(saved), P(scratch), stream), 'mst_total_loss_fwd')
            lc = losses.cpu()
            for i, k in enumerate(nat.LOSS_KEYS):
                key = f'{tag}/loss/{k}'
                if key in z.files:
                    assert abs(float(lc[i]) - float(z[key])) < 2e-5 * max(1., abs(float(z[key]))), (tag, k, float(lc[i]), float(z[key]))
                else:
                    assert np.isnan(float(lc[i])), (tag, k)
            gl = torch.zeros(nat.N_LOSSES, device=device)
            gl[0] = 1.
            nanf = lambda t: torch.full_like(t, float('nan'))
            g_pp, g_il, g_ml, g_bp = nanf(pp), nanf(il), nanf(ml), nanf(bp)
            g_up = nanf(up) if unp else None
            nat.check(lib.mst_total_loss_bwd(P(pp), P(pt), n_p, P(upx), P(utx), n_u, P(il), P(it), il.numel(), P(ml), P(mt), P(bp),
                                             P(bt), P(saved), P(gl), P(g_pp), P(g_up), P(g_il), P(g_ml), P(g_bp), stream),
                      'mst_total_loss_bwd')
            got = dict(pitched_pred=g_pp, instruments=g_il, mode=g_ml, bpm=g_bp)
            if unp:
                got['unpitched_pred'] = g_up
            for name, g in got.items():
                e = rel(g.cpu().numpy(), z[f'{tag}/grad/{name}'])
                assert e < TOL, (tag, name, e)
