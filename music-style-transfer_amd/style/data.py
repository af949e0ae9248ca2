"""Host-side tensor preparation with the reference's names (style/data.py:19-31,122-169).

Only the parts the model path needs are here: the instrument vocabulary / one-hot encoding,
`prepare_input` and `get_used_instruments`.  MIDI parsing and piano-roll conversion
(style/midi*.py, iter_inputs / get_input) stay on the host and are the next scope row;
they depend on `mido`, which this build does not use.
"""
import numpy as np
import torch

from style.model import device

# the 40 most popular General-MIDI programs, in the reference's order (style/midi.py:23-64)
included_instruments = [0, 25, 48, 33, 1, 27, 49, 29, 35, 30, 50, 24, 5, 4, 32, 52, 26, 18, 28, 89, 65, 53, 61, 2, 17, 73,
                        54, 62, 16, 39, 34, 51, 90, 56, 66, 38, 11, 81, 3, 57]
# General-MIDI family of a program = program // 8 (style/midi_programs.txt)
_GM_FAMILIES = ['Piano', 'Chromatic Percussion', 'Organ', 'Guitar', 'Bass', 'Strings', 'Ensemble', 'Brass', 'Reed', 'Pipe',
                'Synth Lead', 'Synth Pad', 'Synth Effects', 'Ethnic', 'Percussive', 'Sound effects']
program2group = {p: _GM_FAMILIES[p // 8] for p in range(128)}
instrument_groups = [program2group[p] for p in included_instruments]
# OneHotEncoder(categories='auto') orders categories by sorted value (style/data.py:23-27)
_instrument_categories = sorted(set(included_instruments))
_group_categories = sorted(set(instrument_groups))
n_instruments = len(included_instruments) + 1            # also percussion
instrument_size = len(_group_categories) + len(_instrument_categories)   # 51
percussion_id = len(included_instruments)
major_mode = 'major'


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
    mode_name = info['scale']['mode']
    is_major = mode_name == major_mode or getattr(mode_name, 'name', None) == major_mode
    mode = torch.tensor([[1., 0.]] if is_major else [[0., 1.]]).to(device)
    bpm = torch.tensor(info['bpm'], dtype=torch.float).unsqueeze(0).to(device)
    return mode, bpm, pitched_channels, instruments_features, unpitched_channels


def get_used_instruments(instruments_features, unpitched_channels):
    used = (instruments_features[:, :, :len(included_instruments)].sum(1) > 0).float()
    percussion = torch.tensor(unpitched_channels is not None, dtype=torch.float).to(used.device).view(1, 1)
    return torch.cat([used, percussion], 1)
