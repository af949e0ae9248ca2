"""The training loop of train-model.py:52-160 as a function, without its per-iteration host syncs.

Same semantics: seed-108 construction order, one song per iteration cut to `800 // C` bars, skipped
empty songs (a skipped iteration also skips the iter_size check), silent percussion dropped, loss with
normalize=True and the positional bpm-before-mode call, gradients accumulated (summed) over
`iter_size` iterations, Adam(lr=.01) + StepLR(200, .9) stepped once per optimizer step, a CSV row and a
progress update per iteration, a whole-module snapshot every `save_interval` iterations.

What is different: the reference reads ~20 scalars back per iteration (`sum() == 0`, `isnan`, `float()`
of every loss leaf).  Here emptiness is decided on the host arrays before upload, the 15 loss leaves of
an iteration stay on the device as one packed tensor, and every `flush_every` iterations ONE copy brings
them back for the CSV rows, the progress bar and the NaN assertion (which therefore fires up to
`flush_every - 1` iterations late).  The optimizer is the fused HIP Adam (style/optim.py).
"""
import csv
import math
import os

import numpy as np
import torch

from style import _native
from style.data import (iter_inputs, instrument_size, n_instruments, included_instruments, prepare_input,
                        get_used_instruments)
from style.model import (device, get_total_loss, PitchedChannelsEncoder, UnpitchedChannelsEncoder, PitchedRhythmEncoder,
                         UnpitchedRhythmEncoder, StyleEncoder, MelodyEncoder, SongInfoModel, PitchedStyleApplier,
                         UnpitchedStyleApplier, StyleTransferModel)
from style.optim import FusedAdam
from style.utils.misc import ProgressBar, assert_dir
from style.utils.parallel import iter_parallel

CSV_FIELDS = ['iteration'] + _native.LOSS_KEYS


def build_model(beat_size=64, bar_size=128, n_rhythm_features=8, style_size=256, melody_size=8, rhythm_size=32, seed=108):
    """train-model.py:52-85: same seed, same construction order => same initial weights."""
    torch.manual_seed(seed)
    pce = PitchedChannelsEncoder(beat_size, bar_size, instrument_size).to(device)
    uce = UnpitchedChannelsEncoder(beat_size, bar_size).to(device)
    pre = PitchedRhythmEncoder(rhythm_size, beat_size, bar_size, instrument_size).to(device)
    ure = UnpitchedRhythmEncoder(rhythm_size, beat_size, bar_size).to(device)
    se = StyleEncoder(style_size, bar_size, instrument_size).to(device)
    me = MelodyEncoder(melody_size, beat_size, bar_size, instrument_size).to(device)
    sim = SongInfoModel(n_rhythm_features, style_size, rhythm_size, n_instruments).to(device)
    psa = PitchedStyleApplier(style_size, melody_size, rhythm_size, instrument_size).to(device)
    usa = UnpitchedStyleApplier(style_size, rhythm_size).to(device)
    return StyleTransferModel(pce, uce, se, me, pre, ure, sim, psa, usa)


def drop_silent(input):
    """Host-side version of train-model.py:105-109: None for a song without pitched notes, percussion
    removed when it is silent (within the `800 // C` bars the model will see)."""
    filename, (info, pitched, features, instruments, unpitched) = input
    max_n_bars = 800 // pitched.shape[0]
    if not np.any(pitched[:, :max_n_bars]):
        return None, max_n_bars
    if unpitched is not None and not np.any(unpitched[:, :max_n_bars]):
        unpitched = None
    return (filename, (info, pitched, features, instruments, unpitched)), max_n_bars


class LossLog:
    """Packed loss tensors of the iterations since the last flush."""

    def __init__(self, path, pbar, flush_every, health=None):
        self.path, self.pbar, self.flush_every = path, pbar, flush_every
        self.health = health          # callable raising on a device-side failure (StyleTransferModel.check_device_status)
        self.pending = []

    def add(self, iteration, packed):
        self.pending.append((iteration, packed))
        if len(self.pending) >= self.flush_every:
            self.flush()

    def flush(self):
        if not self.pending:
            return
        if self.health is not None:
            self.health()             # joins the model's accumulation lanes; a kernel-reported failure names itself here
        values = torch.stack([p for _, p in self.pending]).cpu().numpy()          # the one D2H copy
        rows = []
        for (iteration, _), v in zip(self.pending, values):
            leaf = dict(zip(_native.LOSS_KEYS, v.tolist()))
            assert not math.isnan(leaf['total']), f'loss is NaN at iteration {iteration}'
            has_u = not math.isnan(leaf['channels_loss_unpitched_total'])
            rows.append(dict(iteration=iteration, **{k: ('' if math.isnan(x) else x) for k, x in leaf.items()}))
            if self.pbar is not None:
                self.pbar.add(1, total_loss=leaf['total'], pitched_loss=leaf['channels_loss_pitched_total'],
                              pitched_notes_loss=leaf['channels_loss_pitched_notes_loss'],
                              song_info_loss=leaf['song_info_loss_total'],
                              instruments_loss=leaf['song_info_loss_instruments_loss'],
                              channelss_loss=leaf['channels_loss_total'], mode_loss=leaf['song_info_loss_mode_loss'],
                              bpm_loss=leaf['song_info_loss_bpm_loss'])
                if has_u:
                    self.pbar.update_values(1, unpitched_loss=leaf['channels_loss_unpitched_total'],
                                            unpitched_notes_loss=leaf['channels_loss_unpitched_notes_loss'])
        if self.path:
            self._append_rows(rows)
        self.pending = []

    def _append_rows(self, rows):
        """One CSV row per iteration with the flattened loss leaves (train-model.py:148-149), fixed columns
        CSV_FIELDS; the header goes in only when this call creates the file."""
        assert_dir(self.path)
        fresh = not os.path.exists(self.path) or os.path.getsize(self.path) == 0
        with open(self.path, 'a', newline='', encoding='utf-8') as f:
            out = csv.writer(f)
            if fresh:
                out.writerow(CSV_FIELDS)
            out.writerows([row[k] for k in CSV_FIELDS] for row in rows)


def train(model, inputs, n_iterations=5000, iter_size=2, training_info_path='training.csv', save_path='snapshots/',
          save_interval=100, flush_every=20, progress=True, optimizer=None, fused=True):
    """`inputs`: iterator of (filename, get_input(...)) tuples, e.g. iter_parallel(iter_inputs(...)).
    fused=True runs a loop body as ONE C-ABI call (StyleTransferModel.train_iteration: same arithmetic, same gradients, no
    autograd graph); fused=False is the reference's own sequence model(...) -> get_total_loss -> backward."""
    optimizer = optimizer or FusedAdam(model, lr=.01, step_size=200, gamma=.9)
    optimizer.zero_grad()
    pbar = ProgressBar(n_iterations) if progress else None
    log = LossLog(training_info_path, pbar, flush_every, health=getattr(model, 'check_device_status', None))
    for iteration in range(n_iterations):
        input, max_n_bars = drop_silent(next(inputs))
        if input is None:
            if pbar is not None:
                pbar.n_iterations -= 1
            continue
        info = input[1][0]
        mode, bpm, pitched, features, unpitched = prepare_input(input, max_n_bars)
        used = get_used_instruments(features, unpitched)
        if fused:
            packed = model.train_iteration(mode, bpm, pitched, features, unpitched, used, info['bpm'])
        else:
            (instruments_pred, mode_pred, bpm_pred), pitched_pred, unpitched_pred = model(mode, bpm, pitched, features, unpitched)
            losses = get_total_loss(instruments_pred, used, bpm_pred, info['bpm'], mode_pred, mode, pitched_pred, pitched,
                                    unpitched_pred, unpitched, normalize=True)
            losses['total'].backward()
            packed = losses.packed
        log.add(iteration, packed)
        if (iteration + 1) % iter_size == 0:
            optimizer.step()                      # Adam + StepLR + zero_grad in one launch
        if iteration % save_interval == 0 and save_path:
            log.flush()
            path = os.path.join(save_path, f'{iteration}.pkl')
            assert_dir(path)
            with open(path, 'wb') as f:
                torch.save(model, f)
    log.flush()
    return model


def main(data_path='data/Lakh MIDI Dataset/clean_midi/', **kwargs):
    from style.utils.misc import iter_all_files
    print(f'Using {device}')
    print('Listing data files')
    files = list(iter_all_files(data_path, '**/*.mid'))
    print('Creating model')
    model = build_model()
    print('Training')
    inputs = iter_parallel(iter_inputs(files, included_instruments, shuffle=True, looped=True))
    return train(model, inputs, **kwargs)
