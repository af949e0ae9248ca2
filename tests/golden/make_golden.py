"""Generate golden fixtures by RUNNING the reference (build container only).

    cd /root/repo && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/make_golden.py

Imports the reference's `style.model` unmodified from /root/reference, drives
it exactly like train-model.py:52-90,113-126,151-154 does (same construction
order, seed, Adam/StepLR settings, positional get_total_loss call including the
bpm-before-mode quirk) on the synthetic clips of tools/synth.py, and stores
inputs' seeds + expected outputs as .npz under tests/golden/.  The fixtures are
data only; nothing of the reference's source travels.
"""
import os
import sys

import numpy as np
import torch

import style.model as ref                    # the REFERENCE (PYTHONPATH=/root/reference)
from tools.synth import synth_clip, INSTRUMENT_SIZE, N_INSTRUMENTS

assert os.path.realpath(ref.__file__).startswith('/root/reference/'), ref.__file__
HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)

FULL = dict(beat=64, bar=128, nrf=8, style=256, melody=8, rhythm=32)
SMALL = dict(beat=8, bar=6, nrf=3, style=12, melody=4, rhythm=6)


def build(w, seed=108):
    torch.manual_seed(seed)                  # train-model.py:52
    pce = ref.PitchedChannelsEncoder(w['beat'], w['bar'], INSTRUMENT_SIZE)
    uce = ref.UnpitchedChannelsEncoder(w['beat'], w['bar'])
    pre = ref.PitchedRhythmEncoder(w['rhythm'], w['beat'], w['bar'], INSTRUMENT_SIZE)
    ure = ref.UnpitchedRhythmEncoder(w['rhythm'], w['beat'], w['bar'])
    se = ref.StyleEncoder(w['style'], w['bar'], INSTRUMENT_SIZE)
    me = ref.MelodyEncoder(w['melody'], w['beat'], w['bar'], INSTRUMENT_SIZE)
    sim = ref.SongInfoModel(w['nrf'], w['style'], w['rhythm'], N_INSTRUMENTS)
    psa = ref.PitchedStyleApplier(w['style'], w['melody'], w['rhythm'], INSTRUMENT_SIZE)
    usa = ref.UnpitchedStyleApplier(w['style'], w['rhythm'])
    return ref.StyleTransferModel(pce, uce, se, me, pre, ure, sim, psa, usa)


def flat_losses(d, prefix=''):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if isinstance(v, dict):
            out.update(flat_losses(v, prefix + k + '_'))
        else:
            out[prefix + k] = float(v)
    return out


def iteration(model, clip, capture=None):
    """One train-model.py loop body up to loss.backward() (train-model.py:113-126)."""
    hooks = []
    if capture is not None:
        def mk(name):
            def hook(_m, _i, o):
                o = o if isinstance(o, tuple) else (o,)
                for j, t in enumerate(o):
                    capture[f'mid/{name}/{j}'] = t.detach().numpy().copy()
            return hook
        for name, mod in model.named_children():
            hooks.append(mod.register_forward_hook(mk(name)))
    (ip, mp, bp), xp, xu = model(clip['mode'], clip['bpm'], clip['pitched'],
                                 clip['instruments_features'], clip['unpitched'])
    for h in hooks:
        h.remove()
    losses = ref.get_total_loss(
        ip, clip['used_instruments'],
        bp, clip['bpm_int'],
        mp, clip['mode'],
        xp, clip['pitched'],
        xu, clip['unpitched'],
        normalize=True,
    )
    losses['total'].backward()
    if capture is not None:
        capture['out/instruments'] = ip.detach().numpy().copy()
        capture['out/mode'] = mp.detach().numpy().copy()
        capture['out/bpm'] = bp.detach().numpy().copy()
        capture['out/pitched'] = xp.detach().numpy().copy()
        if xu is not None:
            capture['out/unpitched'] = xu.detach().numpy().copy()
    return flat_losses(losses)


def fingerprint(t):
    x = t.detach().double().reshape(-1)
    return np.array([x.sum(), x.abs().sum(), (x * x).sum(), x[0], x[-1]], dtype=np.float64)


def small_case(name, unpitched):
    C, R, T = 2, 3, 2
    model = build(SMALL, seed=7)
    opt = torch.optim.Adam(model.parameters(), lr=.01)           # train-model.py:89
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=200, gamma=.9)
    out = dict(widths=np.array([SMALL[k] for k in ('beat', 'bar', 'nrf', 'style', 'melody', 'rhythm')]),
               crt=np.array([C, R, T]), unpitched=np.array(int(unpitched)), density=np.array(0.05))
    for n, p in model.named_parameters():
        out['p0/' + n] = p.detach().numpy().copy()
    opt.zero_grad()
    l0 = iteration(model, synth_clip(0, C, R, T, unpitched, density=0.05), capture=out)
    for n, p in model.named_parameters():
        out['g0/' + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    l1 = iteration(model, synth_clip(1, C, R, T, unpitched, density=0.05))
    opt.step(); opt.zero_grad(); sched.step()                    # train-model.py:151-154
    for n, p in model.named_parameters():
        out['p1/' + n] = p.detach().numpy().copy()
    for k, v in l0.items():
        out['loss0/' + k] = np.array(v)
    for k, v in l1.items():
        out['loss1/' + k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'total0', l0['total'], 'total1', l1['total'], len(out), 'arrays')


def full_case():
    """Seed-108 full-width model (train-model.py:52-85) on a 2x2x4 clip: fingerprints only."""
    C, R, T = 2, 2, 4
    model = build(FULL)
    out = dict(crt=np.array([C, R, T]))
    out['n_params'] = np.array(sum(p.numel() for p in model.parameters()))
    for n, p in model.named_parameters():
        out['pf/' + n] = fingerprint(p)
    cap = {}
    l0 = iteration(model, synth_clip(3, C, R, T, True), capture=cap)
    for k, v in cap.items():
        out['f/' + k] = fingerprint(torch.from_numpy(v))
    out['out/instruments'] = cap['out/instruments']
    out['out/mode'] = cap['out/mode']
    out['out/bpm'] = cap['out/bpm']
    out['mid/style'] = cap['mid/style_encoder/0']
    out['slice/pitched'] = cap['out/pitched'][0, 1, 1, 2]        # (10,56,5)
    out['slice/unpitched'] = cap['out/unpitched'][0, 0, 1, 2]
    for n, p in model.named_parameters():
        out['gf/' + n] = fingerprint(p.grad)
    for k, v in l0.items():
        out['loss0/' + k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, 'full_seed108.npz'), **out)
    print('full', l0['total'], int(out['n_params']))


def trajectory_case():
    """6 iterations / 3 Adam steps on the bench clip shape (C=4,R=16,T=4), clips k=0..5."""
    C, R, T = 4, 16, 4
    model = build(FULL)
    opt = torch.optim.Adam(model.parameters(), lr=.01)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=200, gamma=.9)
    opt.zero_grad()
    out = dict(crt=np.array([C, R, T]))
    keys = None
    rows = []
    for it in range(6):
        l = iteration(model, synth_clip(it, C, R, T, True))
        keys = keys or sorted(l)
        rows.append([l[k] for k in keys])
        if (it + 1) % 2 == 0:
            opt.step(); opt.zero_grad(); sched.step()
    out['loss_keys'] = np.array(keys)
    out['losses'] = np.array(rows, dtype=np.float64)
    for n, p in model.named_parameters():
        out['pf/' + n] = fingerprint(p)
    np.savez_compressed(os.path.join(HERE, 'trajectory_seed108.npz'), **out)
    print('traj totals', out['losses'][:, keys.index('total')])


def inference_case():
    """Forward-only behaviour of the reference (style/style_transfer.py:41-54,101-131; style/model.py:751-782,818-832):
    extract_style on a composition A (with percussion) and on a style song B (pitched only, unpitched_channels=None),
    predict_song_info / apply_style with B's style on A's melody and rhythm, hard_output of the predictions (with its
    in-place velocity mutation), hard_output on a hand-made tensor with ties / threshold values — plus the SUM of the
    two clips' gradients before the optimizer step (train-model.py:126 accumulates), which pins two pending forwards."""
    C, R, T = 2, 3, 2
    model = build(SMALL, seed=7)
    out = dict(widths=np.array([SMALL[k] for k in ('beat', 'bar', 'nrf', 'style', 'melody', 'rhythm')]),
               crt=np.array([C, R, T]), density=np.array(0.05))
    z0 = np.load(os.path.join(HERE, 'small_unpitched.npz'))      # same seed-7 parameters: not stored twice
    for n, p in model.named_parameters():
        assert np.array_equal(z0['p0/' + n], p.detach().numpy()), n
    a = synth_clip(0, C, R, T, True, density=0.05)
    b = synth_clip(1, C, R, T, True, density=0.05)
    model.zero_grad()
    iteration(model, a)
    iteration(model, b)
    for n, p in model.named_parameters():
        out['g01/' + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    with torch.no_grad():
        style_a, melody_a, rhythm_a = model.extract_style(a['mode'], a['bpm'], a['pitched'], a['instruments_features'], a['unpitched'])
        style_b, _, _ = model.extract_style(b['mode'], b['bpm'], b['pitched'], b['instruments_features'], None)
        ip, mp, bp = model.predict_song_info(style_b, rhythm_a)
        instr = b['instruments_features'][:, :1]            # one predicted instrument (style_transfer.py:124-127)
        xp, xu = model.apply_style(style_b, melody_a, rhythm_a, instr, True)
        xp1, xu1 = model.apply_style(style_b, melody_a, rhythm_a, b['instruments_features'], False)
        assert xu1 is None
        out.update({'swap/style_a': style_a.numpy().copy(), 'swap/style_b': style_b.numpy().copy(),
                    'swap/melody_a': melody_a.numpy().copy(), 'swap/rhythm_a': rhythm_a.numpy().copy(),
                    'swap/instruments': ip.numpy().copy(), 'swap/mode': mp.numpy().copy(), 'swap/bpm': bp.numpy().copy(),
                    'swap/pitched': xp.numpy().copy(), 'swap/unpitched': xu.numpy().copy(),
                    'swap/pitched_all_channels': xp1.numpy().copy()})
        hp = ref.hard_output(xp)                             # mutates xp's velocities in place (style/model.py:822)
        hu = ref.hard_output(xu)
        out.update({'swap/hard_pitched': hp.numpy().copy(), 'swap/hard_unpitched': hu.numpy().copy(),
                    'swap/pitched_after': xp.numpy().copy(), 'swap/unpitched_after': xu.numpy().copy()})
        # hand-made: velocities around the .01 threshold, accidental ties, maxima around the .1 threshold
        g = torch.Generator().manual_seed(11)
        x = torch.rand(1, 2, 1, 2, 10, 56, 5, generator=g)
        x[..., 1] = torch.tensor([0., .005, .01, .0100001, .02, .5, 1., .009999])[torch.randint(0, 8, x.shape[:-1], generator=g)]
        tie = torch.rand(x.shape[:-1], generator=g) < .3
        x[..., 3] = torch.where(tie, x[..., 2], x[..., 3])                              # two equal maxima stay two ones
        low = torch.rand(x.shape[:-1], generator=g) < .2
        x[..., 2:] = torch.where(low.unsqueeze(-1), x[..., 2:] * .1, x[..., 2:])         # max <= .1 -> all zeros
        x[0, 0, 0, 0, 0, 0, 2:] = torch.tensor([.1, .1, .05])
        x[0, 0, 0, 0, 0, 1, 2:] = torch.tensor([.1000001, .05, .1000001])
        u = torch.rand(1, 1, 1, 2, 10, 47, 2, generator=g)
        u[..., 1] = torch.tensor([0., .005, .01, .0100001, .02, .5, 1., .009999])[torch.randint(0, 8, u.shape[:-1], generator=g)]
        out['hard/x_in'] = x.numpy().copy(); out['hard/u_in'] = u.numpy().copy()
        out['hard/x_out'] = ref.hard_output(x).numpy().copy(); out['hard/u_out'] = ref.hard_output(u).numpy().copy()
        out['hard/x_after'] = x.numpy().copy(); out['hard/u_after'] = u.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'inference_small.npz'), **out)
    print('inference: bpm', float(bp), 'hard ones', float(hp[..., 2:].sum()), len(out), 'arrays')


def loss_case():
    """get_total_loss on its own (style/model.py:935-997) with normalize=False — the reference's DEFAULT, which its training
    script overrides — and normalize=True, on free-standing predictions (not model outputs): every loss leaf and the gradient
    of `total` with respect to each prediction, with and without the unpitched pair.  Called with the positional contract of
    train-model.py:115-122 (bpm before mode)."""
    C, R, T = 2, 2, 2
    clip = synth_clip(4, C, R, T, True, density=0.08)
    g = torch.Generator().manual_seed(21)
    pp = torch.rand(clip['pitched'].shape, generator=g) * .98 + .01          # accidentals go through BCE: keep inside (0, 1)
    pp[..., 0] = pp[..., 0] * 6                                              # durations live in (0, 6)
    up = torch.rand(clip['unpitched'].shape, generator=g) * .98 + .01
    up[..., 0] = up[..., 0] * 6
    il = torch.randn(1, N_INSTRUMENTS, generator=g)
    ml = torch.randn(1, 2, generator=g)
    bp = torch.tensor([97.5])
    out = dict(crt=np.array([C, R, T]), density=np.array(0.08), clip_id=np.array(4))
    out.update({'in/pitched_pred': pp.numpy().copy(), 'in/unpitched_pred': up.numpy().copy(), 'in/instruments': il.numpy().copy(),
                'in/mode': ml.numpy().copy(), 'in/bpm': bp.numpy().copy()})
    for normalize in (False, True):
        for unp in (True, False):
            leaves = [t.clone().requires_grad_(True) for t in (pp, up, il, ml, bp)]
            a, b, c, d, e = leaves
            losses = ref.get_total_loss(c, clip['used_instruments'], e, clip['bpm_int'], d, clip['mode'], a, clip['pitched'],
                                        b if unp else None, clip['unpitched'] if unp else None, normalize=normalize)
            losses['total'].backward()
            tag = f'n{int(normalize)}u{int(unp)}'
            for k, v in flat_losses(losses).items():
                out[f'{tag}/loss/{k}'] = np.array(v)
            for name, t in zip(('pitched_pred', 'unpitched_pred', 'instruments', 'mode', 'bpm'), leaves):
                if t.grad is not None:
                    out[f'{tag}/grad/{name}'] = t.grad.numpy().copy()
            print(tag, 'total', float(losses['total']))
    np.savez_compressed(os.path.join(HERE, 'loss_normalize.npz'), **out)


if __name__ == '__main__':
    if sys.argv[1:] == ['inference']:          # added later: leaves the earlier fixtures untouched
        inference_case()
        sys.exit(0)
    if sys.argv[1:] == ['loss']:
        loss_case()
        sys.exit(0)
    small_case('small_unpitched', True)
    small_case('small_pitched_only', False)
    full_case()
    trajectory_case()
    inference_case()
    loss_case()
