"""Audio-extension cases shared by the CPU-interpreter test and the -m gpu test (same C ABI, other build / device).  The oracle is
build-defined (torch.stft / matmul / autograd / Adam on the CPU): EXTENSION, PARITY UNPINNED — the reference has no audio path."""
import numpy as np
import torch

from oracle import audio_oracle as ao
from simutil import rel


def synth_audio(n, seed=0, sr=44100):
    """Deterministic test signal: a few partials with vibrato + noise, in [-1, 1]."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / sr
    x = torch.zeros(n, dtype=torch.float64)
    for f0, a in ((220., .4), (440., .25), (1375., .15), (5234., .1)):
        x += a * torch.sin(2 * np.pi * f0 * t + 3. * torch.sin(2 * np.pi * 5. * t))
    x += .05 * torch.randn(n, generator=g, dtype=torch.float64)
    return (x / x.abs().max()).float()


def stft_case(plan_cls, native, device, n, n_fft, hop):
    from style import _native as nat
    audio = synth_audio(n)
    plan = plan_cls(n, n_fft, hop, device=device, native=native)
    ref = ao.stft(audio, n_fft, hop)
    assert (plan.frames, plan.bins) == tuple(ref.shape) and plan.ld % 8 == 0 and plan.ld >= plan.bins
    spec, mag = plan.stft(audio.to(device))
    e = rel(torch.view_as_real(spec.cpu()).numpy(), torch.view_as_real(ref).numpy())
    assert e < 1e-4, ('stft rel-L2', e)
    assert rel(mag.cpu()[:, :plan.bins].numpy(), ref.abs().numpy()) < 1e-4
    assert float(mag[:, plan.bins:].abs().max()) == 0.0              # pad columns are part of the matrix: zeros
    # magnitude-only and spectrum-only calls write the same values
    _, mag2 = plan.stft(audio.to(device), want_spec=False)
    spec2, _ = plan.stft(audio.to(device), want_mag=False)
    assert torch.equal(mag2, mag) and torch.equal(torch.view_as_real(spec2), torch.view_as_real(spec))
    with_bad = torch.zeros(n + 1)
    try:
        plan.stft(with_bad.to(device))
        raise AssertionError('length check missing')
    except nat.MstError:
        pass
    return plan, audio, mag


def gram_and_iteration_case(plan, mag, device, iters=3, lr=1e-2):
    feat = mag.cpu()[:, :plan.bins].contiguous()
    g = plan.gram(mag)
    gref = ao.gram(feat)
    assert rel(g.cpu()[:plan.bins, :plan.bins].numpy(), gref.numpy()) < 1e-4
    assert torch.equal(g, g.t()) and float(g[plan.bins:].abs().max()) == 0.0 and float(g[:, plan.bins:].abs().max()) == 0.0
    # style iteration: x starts as a perturbed copy of the content magnitudes, the style Gram comes from a time-reversed, scaled clip
    style = torch.flip(feat, [0]) * 1.3
    gs_ref = ao.gram(style)
    gs = torch.zeros(plan.ld, plan.ld)
    gs[:plan.bins, :plan.bins] = gs_ref
    x0 = feat * .9 + .01
    x = torch.zeros(plan.frames, plan.ld)
    x[:, :plan.bins] = x0
    x = x.to(device)
    opt = plan.optimizer_state()
    xr, losses_ref = ao.style_iterations(x0, gs_ref, iters, lr)
    losses = []
    for _ in range(iters):
        losses.append(float(plan.style_iteration(x, gs.to(device), opt, lr=lr).cpu()[0]))
    for a, b in zip(losses, losses_ref):
        assert abs(a - b) <= 2e-4 * abs(b), (losses, losses_ref)
    assert losses[-1] < losses[0]                                     # it optimises
    assert float(x[:, plan.bins:].abs().max()) == 0.0
    # Adam's first steps move every element by ~lr * sign(g): compare the update as a whole (like tests/test_train_driver.py)
    upd, upd_ref = x.cpu()[:, :plan.bins] - x0, xr - x0
    assert float((upd - upd_ref).norm() / upd_ref.norm()) < 2e-2
    return losses
