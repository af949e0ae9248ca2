"""Deterministic synthetic piano-roll clips (TEST / BENCH INFRASTRUCTURE; SURVEY §8(d) input recipe).

Lives outside oracle/ because bench.py's timed leg needs inputs too, and only its cpu_baseline leg may touch the oracle;
the tests import it from here as well.

No reference counterpart: the reference trains on the Lakh MIDI dataset, which
cannot be shipped.  Shapes and feature order follow the reference's piano-roll
layout (style/midi_conversion.py:501-511 feature order
[duration, velocity, flat, natural, sharp]; style/data.py:130-156 tensor
shapes; style/data.py:122-127 instrument one-hot ++ group one-hot;
style/data.py:159-169 used-instrument target).

This file imports nothing from the reference and nothing from the product.
"""
import torch

N_FRACTIONS = 10
N_PITCHED_NOTES = 56
N_UNPITCHED_NOTES = 47
N_INCLUDED = 40          # len(included_instruments), style/midi.py:23-64
N_GROUPS = 11            # distinct program groups among those 40
INSTRUMENT_SIZE = N_INCLUDED + N_GROUPS   # 51, style/data.py:29-30
N_INSTRUMENTS = N_INCLUDED + 1            # + percussion, style/data.py:21


def _roll(g, shape, n_acc, density):
    mask = (torch.rand(shape, generator=g) < density).float()
    duration = torch.rand(shape, generator=g) * 4.0 * mask
    velocity = (0.1 + 0.9 * torch.rand(shape, generator=g)) * mask
    feats = [duration, velocity]
    if n_acc:
        which = torch.randint(0, n_acc, shape, generator=g)
        for a in range(n_acc):
            feats.append((which == a).float() * mask)
    return torch.stack(feats, -1).contiguous()


def synth_clip(k, C, R, T, unpitched=True, density=0.02, bpm=120):
    """Clip number `k`: dict of CPU float32 tensors shaped like prepare_input's output."""
    g = torch.Generator().manual_seed(int(k))
    pitched = _roll(g, (1, C, R, T, N_FRACTIONS, N_PITCHED_NOTES), 3, density)
    unp = None
    if unpitched:
        unp = _roll(g, (1, 1, R, T, N_FRACTIONS, N_UNPITCHED_NOTES), 0, density)
    feats = torch.zeros(1, C, INSTRUMENT_SIZE)
    for c in range(C):
        feats[0, c, c % N_INCLUDED] = 1.0
        feats[0, c, N_INCLUDED + (c % N_GROUPS)] = 1.0
    mode = torch.tensor([[1.0, 0.0]])
    bpm_t = torch.tensor([float(bpm)])
    used = (feats[:, :, :N_INCLUDED].sum(1) > 0).float()
    used = torch.cat([used, torch.tensor([[1.0 if unpitched else 0.0]])], 1)
    return dict(mode=mode, bpm=bpm_t, bpm_int=int(bpm), pitched=pitched,
                instruments_features=feats, unpitched=unp, used_instruments=used)
