"""AUDIO EXTENSION (no reference counterpart, oracle build-defined, parity unpinned): STFT, feature Gram and the
style-transfer optimisation iteration.  CPU: the same .hip sources on the hipsim interpreter at small sizes; -m gpu: the gfx950
build at BASELINE.json's sizes (30 s @ 44.1 kHz, STFT 1024/256; 2048/512 at 48 kHz) through size-independent properties and the
oracle."""
import pytest
import torch

import audio_cases as ac
from simutil import sim_native


def _plan_cls():
    from style.audio import AudioPlan
    return AudioPlan


@pytest.mark.parametrize('n,n_fft,hop', [(2300, 1024, 256), (4500, 2048, 512)])
def test_stft_gram_iteration_on_the_interpreter(n, n_fft, hop):
    plan, audio, mag = ac.stft_case(_plan_cls(), sim_native(), 'cpu', n, n_fft, hop)
    if n_fft == 1024:
        ac.gram_and_iteration_case(plan, mag, 'cpu', iters=2)


def test_bad_arguments_are_refused():
    from style import _native as nat
    with pytest.raises(nat.MstError):
        _plan_cls()(10000, 512, 128, device='cpu', native=sim_native())       # only 1024 and 2048 are instantiated
    with pytest.raises(nat.MstError):
        _plan_cls()(300, 1024, 256, device='cpu', native=sim_native())         # shorter than the reflection padding


@pytest.mark.gpu
@pytest.mark.parametrize('seconds,sr,n_fft,hop', [(30, 44100, 1024, 256), (6, 48000, 2048, 512)])
def test_stft_at_baseline_sizes(seconds, sr, n_fft, hop):
    n = seconds * sr
    plan, audio, mag = ac.stft_case(_plan_cls(), None, 'cuda:0', n, n_fft, hop)
    if n_fft == 1024:
        assert (plan.frames, plan.bins) == (5168, 513)                         # BASELINE.json: 30 s @ 44.1 kHz, STFT 1024 / 256
    # Parseval per frame (size-independent): sum_k w_k |X[k]|^2 = N sum_n (w[n] x[n])^2, w_k = 1 for DC / Nyquist, 2 otherwise
    spec, _ = plan.stft(audio.cuda())
    p = spec.abs().double() ** 2
    lhs = p[:, 0] + p[:, -1] + 2 * p[:, 1:-1].sum(1)
    pad = torch.nn.functional.pad(audio.double()[None, None], (n_fft // 2, n_fft // 2), mode='reflect')[0, 0]
    frames = pad.unfold(0, n_fft, hop)[:plan.frames] * torch.hann_window(n_fft, dtype=torch.float64)
    rhs = n_fft * (frames ** 2).sum(1)
    assert float(((lhs.cpu() - rhs).abs() / rhs).max()) < 1e-4


@pytest.mark.gpu
def test_gram_and_style_iteration_at_baseline_size():
    n = 30 * 44100
    plan, audio, mag = ac.stft_case(_plan_cls(), None, 'cuda:0', n, 1024, 256)
    losses = ac.gram_and_iteration_case(plan, mag, 'cuda:0', iters=3)
    print('audio style iteration losses', losses)
