"""Host-side data pipeline with the reference's names (style/data.py:19-169): instrument vocabulary and
one-hot encoding, MIDI file -> model input (`iter_inputs`, `get_input`), `prepare_input`,
`get_used_instruments`.  Everything here runs on the host, per song; only `prepare_input` touches the
device (one H2D copy per tensor).  sklearn's OneHotEncoder and pandas are not used: categories are the
sorted distinct values, which is what `categories='auto'` produces (style/data.py:23-27).
"""
import numpy as np
import torch

from style.exceptions import MidiFormatError
from style.midi import load_midi_from_file, is_pitched, program2instrument, program2group, popular_instruments
from style.midi_conversion import read_midi, ChannelConverter, NoteTable, _key_weights
from style.model import device
from style.scales import key_names, get_scale, major_mode

included_instruments = popular_instruments
instrument_groups = [program2group[p] for p in included_instruments]
_instrument_categories = sorted(set(included_instruments))
_group_categories = sorted(set(instrument_groups))
n_instruments = len(included_instruments) + 1            # also percussion
instrument_size = len(_group_categories) + len(_instrument_categories)   # 51
percussion_id = len(included_instruments)


class CategoryOneHot:
    """The slice of sklearn's OneHotEncoder(sparse=False, categories='auto') the reference's callers use (style/data.py:23-27,
    124-125; style/style_transfer.py:115): `categories_`, `transform` of an (n, 1) column, `inverse_transform` of (n, k) rows."""

    def __init__(self, categories):
        self.categories_ = [np.array(categories)]
        self._index = {c: i for i, c in enumerate(categories)}

    def transform(self, column):
        column = np.asarray(column).reshape(-1)
        out = np.zeros((len(column), len(self._index)))
        for i, v in enumerate(column.tolist()):
            out[i, self._index[v]] = 1.
        return out

    def inverse_transform(self, rows):
        return self.categories_[0][np.asarray(rows).argmax(1)].reshape(-1, 1)


instruments_one_hot_encoder = CategoryOneHot(_instrument_categories)
groups_one_hot_encoder = CategoryOneHot(_group_categories)


def iter_all_midis(files, shuffle=False, looped=False):
    """(file, channels, info) of every readable file; unreadable files and MidiFormatErrors are skipped."""
    if shuffle:
        files = files[:]
        np.random.shuffle(files)
    while True:
        for file in files:
            mid = load_midi_from_file(file)
            if mid is None:
                continue
            try:
                channels, info = read_midi(mid)
            except MidiFormatError:
                continue
            yield file, channels, info
        if not looped:
            return


def iter_inputs(files, instruments, min_n_messages=100, *args, **kwargs):
    for filename, channels, info in iter_all_midis(files, *args, **kwargs):
        channels = [c for c in channels
                    if c['instrument_id'] in [-1, *instruments] and len(c['messages']) >= min_n_messages]
        if not any(is_pitched(c['instrument_id']) for c in channels):
            continue
        try:
            yield filename, get_input(channels, info)
        except Exception:
            print(filename)
            raise


def merge_nchannels(nchannels):
    """Channels playing the same instrument become one, notes ordered by onset (stable) (:103-114)."""
    instrument_ids = {n['instrument_id'] for n in nchannels}
    assert len(instrument_ids) == 1
    instrument_id = instrument_ids.pop()
    notes = NoteTable.concat([n['notes'] for n in nchannels])
    return {
        'channel_id': min(n['channel_id'] for n in nchannels),
        'instrument_id': instrument_id,
        'instrument_name': program2instrument[instrument_id],
        'notes': notes.take(np.argsort(notes.time, kind='stable')),
    }


def get_input(channels, info):
    """channels, info -> (info + detected scale, pitched rolls (C,R,T,10,56,5), instrument features
    (C,51), instrument ids, unpitched rolls (1,R,T,10,47,2) | None) (:66-100)."""
    cc = ChannelConverter(info)
    by_instrument = {}
    for channel in channels:
        by_instrument.setdefault(channel['instrument_id'], []).append(cc.channel2nchannel(channel))
    nchannels = [merge_nchannels(group) for group in by_instrument.values()]
    pitched = [n for n in nchannels if is_pitched(n['instrument_id'])]
    unpitched = [n for n in nchannels if not is_pitched(n['instrument_id'])]

    # seconds per pitch class: rows = keys, columns = channels; absent keys count 0 (DataFrame.sum skips NaN)
    seconds = np.zeros((len(key_names), len(pitched)))
    for j, nchannel in enumerate(pitched):
        weights, present = _key_weights(info, nchannel)
        seconds[present, j] = weights[present]
    keys_dist = seconds.sum(axis=1)
    keys_dist /= keys_dist.sum()
    info['scale'] = get_scale(keys_dist=keys_dist)

    pitched_vchannels = np.stack([cc.nchannel2vchannel(n) for n in pitched])
    unpitched_vchannels = np.stack([cc.nchannel2vchannel(n) for n in unpitched]) if unpitched else None
    instruments = [n['instrument_id'] for n in pitched]
    return info, pitched_vchannels, encode_instruments(instruments), instruments, unpitched_vchannels


def encode_instruments(instruments):
    """one-hot(program) ++ one-hot(GM family) per pitched channel -> (C, 51) float64 array."""
    x = np.zeros((len(instruments), instrument_size))
    for i, p in enumerate(instruments):
        x[i, _instrument_categories.index(p)] = 1.
        x[i, len(_instrument_categories) + _group_categories.index(program2group[p])] = 1.
    return x


def prepare_input(input, max_n_bars=None):
    """(filename, (info, pitched, instruments_features, instruments, unpitched)) -> device tensors
    (mode, bpm, pitched_channels, instruments_features, unpitched_channels), batch dim added."""
    _, (info, pitched_channels, instruments_features, _, unpitched_channels) = input
    if max_n_bars is None:
        max_n_bars = pitched_channels.shape[1]
    pitched_channels = torch.tensor(pitched_channels[:, :max_n_bars], dtype=torch.float).to(device).unsqueeze(0)
    instruments_features = torch.tensor(instruments_features, dtype=torch.float).to(device).unsqueeze(0)
    if unpitched_channels is not None:
        unpitched_channels = torch.tensor(unpitched_channels[:, :max_n_bars], dtype=torch.float).to(device).unsqueeze(0)
    is_major = info['scale']['mode'] in (major_mode, 'major')
    mode = torch.tensor([[1., 0.]] if is_major else [[0., 1.]]).to(device)
    bpm = torch.tensor(info['bpm'], dtype=torch.float).unsqueeze(0).to(device)
    return mode, bpm, pitched_channels, instruments_features, unpitched_channels


def get_used_instruments(instruments_features, unpitched_channels):
    used = (instruments_features[:, :, :len(included_instruments)].sum(1) > 0).float()
    percussion = torch.tensor(unpitched_channels is not None, dtype=torch.float).to(used.device).view(1, 1)
    return torch.cat([used, percussion], 1)
