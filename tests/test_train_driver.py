"""Training driver (train-model.py:52-160): logging / prefetch / snapshot glue on the CPU,
and (-m gpu) the loop itself on songs read from the reference's example files against the oracle's
trajectory (same seed-108 weights, gradient accumulation over iter_size = 2, Adam + StepLR)."""
import csv
import math
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
MIDI = os.path.join(HERE, 'golden', 'midi')
SONGS = [os.path.join(MIDI, n) for n in ('Minuetto in sol magg. BWV App. 114.mid', 'Angie.4.mid',
                                         'Nocturne No. 1 in E minor, Op. 72_ Andante.mid', 'Vogue.3.mid')]


def test_progress_meter_matches_reference_formulas():
    from style.utils.misc import ProgressBar, dict_map, flatten_underscore
    p = ProgressBar(None, momentum=.9)
    p.pbar = None
    s = n = 0.
    for v in (1., 2., 4.):
        p.update_values(1, loss=v, other=None)
        s, n = s * .9 + v, n * .9 + 1
        assert p['loss'] == s / n
    assert p.min_values['loss'] == 1. and 'other' not in p.avg_values
    p.initial_values(loss=10.)
    p.update_values(1, loss=0.)
    assert p['loss'] == 10. * .9
    nested = dict(total=1., a=dict(b=2., c=dict(d=None)))
    assert flatten_underscore(nested) == {'total': 1., 'a_b': 2., 'a_c_d': None}
    assert dict_map(lambda x: None if x is None else x * 2, nested, recursive=True) == dict(total=2., a=dict(b=4., c=dict(d=None)))


def test_csv_log_and_prefetch(tmp_path):
    from style.train import LossLog, CSV_FIELDS
    from style.utils.parallel import iter_parallel
    path = str(tmp_path / 'log' / 'training.csv')
    log = LossLog(path, None, flush_every=100)
    blank = {k: '' for k in CSV_FIELDS}
    log._append_rows([dict(blank, iteration=0, total=1.5)])                       # creates the file: header + row
    log._append_rows([dict(blank, iteration=1, total=2.5), dict(blank, iteration=2)])    # appends: no second header
    rows = list(csv.DictReader(open(path)))
    assert [r['iteration'] for r in rows] == ['0', '1', '2'] and rows[1]['total'] == '2.5' and rows[2]['total'] == ''
    assert list(rows[0].keys()) == CSV_FIELDS
    assert list(iter_parallel(iter(range(7)))) == list(range(7))

    def failing():
        yield 'a'
        raise KeyError('boom')
    it = iter_parallel(failing())
    assert next(it) == 'a'
    with pytest.raises(KeyError):
        next(it)


def test_drop_silent_and_loss_log(tmp_path):
    from style import _native
    from style.train import drop_silent, LossLog, CSV_FIELDS
    pitched = np.zeros((2, 500, 4, 10, 56, 5))
    unpitched = np.zeros((1, 500, 4, 10, 47, 2))
    song = ('f', ({}, pitched, None, [0, 1], unpitched))
    assert drop_silent(song) == (None, 400)
    pitched[0, 450, 0, 0, 0, 1] = .5                      # beyond the 800 // C = 400 bars the model sees
    assert drop_silent(song)[0] is None
    pitched[0, 3, 0, 0, 0, 1] = .5
    kept, cap = drop_silent(song)
    assert cap == 400 and kept[1][4] is None                # silent percussion dropped
    unpitched[0, 1, 0, 0, 0, 1] = .3
    assert drop_silent(song)[0][1][4] is unpitched
    path = str(tmp_path / 'training.csv')
    log = LossLog(path, None, flush_every=2)
    a = torch.arange(15, dtype=torch.float32)
    b = a.clone()
    b[7:11] = float('nan')                                   # no percussion in that iteration
    log.add(0, a)
    assert not os.path.exists(path)                          # nothing read back before the flush
    log.add(1, b)
    rows = list(csv.DictReader(open(path)))
    assert list(rows[0].keys()) == CSV_FIELDS and CSV_FIELDS[1:] == _native.LOSS_KEYS
    assert rows[0]['total'] == '0.0' and rows[1]['channels_loss_unpitched_total'] == '' and rows[1]['iteration'] == '1'
    bad = a.clone()
    bad[0] = float('nan')
    log.add(2, bad)
    with pytest.raises(AssertionError):
        log.flush()
    # a device-side failure names itself at the flush, before the NaN assert gets to speak
    def health():
        raise _native.MstError('device status: ' + _native.describe_status(_native.DEV_LSTM_TIMEOUT))
    sick = LossLog(path, None, flush_every=1, health=health)
    with pytest.raises(_native.MstError, match='MST_DEV_LSTM_TIMEOUT'):
        sick.add(3, bad)


@pytest.mark.gpu
def test_lstm_exchange_timeout_surfaces_through_the_model(monkeypatch):
    """MST_LSTM_FLAVOUR=2 makes workgroup 0 of the 12-workgroup StyleEncoder LSTM publish its first step under a wrong epoch
    (mst_plan_options.lstm_flavour = 2): every consumer runs into the 0.2 s timeout, ORs MST_DEV_LSTM_TIMEOUT into the plan's
    device status word and the launch drains; StyleTransferModel.check_device_status (what LossLog.flush calls) raises."""
    import time
    from tools.synth import synth_clip
    from style import _native
    from style.train import build_model
    monkeypatch.setenv('MST_LSTM_FLAVOUR', '2')
    model = build_model(seed=108)
    clip = {k: (v.to('cuda:0') if torch.is_tensor(v) else v) for k, v in synth_clip(0, 2, 3, 2, True).items()}
    args = (clip['mode'], clip['bpm'], clip['pitched'], clip['instruments_features'], clip['unpitched'], clip['used_instruments'],
            clip['bpm_int'])
    t0 = time.time()
    packed = model.train_iteration(*args)
    torch.cuda.synchronize()
    assert time.time() - t0 < 20., 'the faulted launch must drain within its time bound, not hang'
    assert torch.isnan(packed[0])
    with pytest.raises(_native.MstError, match='MST_DEV_LSTM_TIMEOUT'):
        model.check_device_status()
    model.check_device_status()                      # read-and-cleared: healthy again as far as the status word goes
    monkeypatch.setenv('MST_LSTM_FLAVOUR', '0')      # a new plan (the options are part of the cache key) on healthy kernels
    model.zero_grad()
    packed = model.train_iteration(*args)
    model.check_device_status()
    assert torch.isfinite(packed[0])


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [True, False])
def test_train_loop_matches_oracle_trajectory(tmp_path, fused):
    from oracle import style_oracle as so
    from style import style_transfer as st
    from style.data import prepare_input, get_used_instruments
    from style.train import build_model, train, drop_silent
    songs = [st.get_model_input(p) for p in SONGS]
    model = build_model(seed=108)
    named = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in model.named_parameters()}
    start = torch.cat([p.detach().reshape(-1) for p in named.values()])
    csv_path, snap = str(tmp_path / 'training.csv'), str(tmp_path / 'snapshots')
    # fused: a loop body as one C-ABI call (the default); not fused: the reference's sequence through autograd
    train(model, iter(songs), n_iterations=4, iter_size=2, training_info_path=csv_path, save_path=snap, save_interval=2,
          flush_every=3, progress=False, fused=fused)
    rows = list(csv.DictReader(open(csv_path)))
    assert [int(r['iteration']) for r in rows] == [0, 1, 2, 3]
    assert sorted(os.listdir(snap)) == ['0.pkl', '2.pkl']
    # oracle: the same four iterations, two optimizer steps
    opt = so.Adam(named.values())
    for it, song in enumerate(songs):
        inp, cap = drop_silent(song)
        mode, bpm, pitched, features, unpitched = (None if t is None else t.cpu() for t in prepare_input(inp, cap))
        clip = dict(mode=mode, bpm=bpm, pitched=pitched, instruments_features=features, unpitched=unpitched,
                    used_instruments=get_used_instruments(features, unpitched).cpu(), bpm_int=inp[1][0]['bpm'])
        _, ref = so.iteration(named, clip, fast=True)
        for k, v in ref.items():
            assert abs(float(rows[it][k]) - v) < 5e-4 * max(1., abs(v)), (it, k, rows[it][k], v)
        if unpitched is None:
            assert rows[it]['channels_loss_unpitched_total'] == ''
        if (it + 1) % 2 == 0:
            opt.step()
    # Adam's first steps move every weight by ~lr * sign(g): where |g| is at rounding level the sign is noise, so the
    # comparison is on the update vector as a whole (2 steps of lr = .01 => |delta| <= .02 per element)
    got = torch.cat([p.detach().cpu().reshape(-1) for _, p in model.named_parameters()]) - start
    want = torch.cat([p.detach().reshape(-1) for p in named.values()]) - start
    assert float(want.abs().max()) <= .0201 and float(got.abs().max()) <= .0201
    assert float((got - want).norm() / want.norm()) < 2e-2
    assert float(((got - want).abs() > 1e-3).float().mean()) < 2e-3
    # whole-module snapshot loads back (train-model.py:156-160)
    loaded = torch.load(os.path.join(snap, '2.pkl'), weights_only=False)
    assert type(loaded).__name__ == 'StyleTransferModel'
