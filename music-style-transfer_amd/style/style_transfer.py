"""Style transfer between MIDI songs: the reference's inference driver (style/style_transfer.py:22-158)
over the HIP model.  Forward only, one pass per song:

    composition.mid --extract_style--> (style_A, melody_A, rhythm_A)
    style_k.mid     --extract_style--> style_k
    apply_style(style_k, melody_A, rhythm_A) --hard_output--> piano-rolls --> MIDI

The per-song host decisions are split out so they can be tested without a GPU:
`select_instruments` (top-n programs out of the 41 predicted logits, :105-116), `combine_info` (:134-142)
and `decode_rolls` (rolls -> MidiFile, the host half of `decode_midi`).  Behaviour kept from the
reference: songs are cut to `1000 // C` bars for the encoder (:69) while `original/*.mid` is written
from the full-length rolls; `apply_style` overwrites `info['tempo']` and `info['scale']['mode']` of the
dict it is given (the scale dict is shared with the style song's info); an existing output directory is
never cleared (the reference's `shutil.rmtree` raises NameError inside a bare `except`, :31-34).
"""
import os

import numpy as np
import torch

from style import smf
from style.data import (included_instruments, get_input, prepare_input, percussion_id, encode_instruments,
                        _instrument_categories)
from style.midi import load_midi_from_file, create_midi
from style.midi_conversion import ChannelConverter, read_midi
from style.model import device, hard_output
from style.scales import major_mode, minor_mode


def transfer_style(model, composition_path, style_paths, output_path):
    composition_name = os.path.splitext(os.path.basename(composition_path))[0]
    composition_input = get_model_input(composition_path)
    _, (composition_info, composition_pitched, _, composition_instruments, composition_unpitched) = composition_input
    composition_cc = ChannelConverter(composition_info)
    style_, melody, rhythm = extract_style(model, composition_input)
    output_path = os.path.join(output_path, composition_name)

    save(composition_cc, composition_pitched, composition_unpitched, composition_instruments,
         os.path.join(output_path, f'original/{composition_name}.mid'))
    apply_style(model, composition_info, style_, melody, rhythm, len(composition_instruments),
                os.path.join(output_path, f'{composition_name} (reconstructed).mid'))
    for style_path in style_paths:
        style_name = os.path.splitext(os.path.basename(style_path))[0]
        style_input = get_model_input(style_path)
        _, (style_info, style_pitched, _, style_instruments, style_unpitched) = style_input
        style, _, _ = extract_style(model, style_input)
        save(ChannelConverter(style_info), style_pitched, style_unpitched, style_instruments,
             os.path.join(output_path, f'original/{style_name}.mid'))
        info = combine_info(style_info=style_info, melody_info=composition_info)
        apply_style(model, info, style, melody, rhythm, len(style_instruments),
                    os.path.join(output_path, f'{composition_name} ({style_name} style).mid'))


def get_model_input(path):
    mid = load_midi_from_file(path)
    if mid is None:
        return None
    channels, info = read_midi(mid)
    channels = [c for c in channels if c['instrument_id'] in [-1, *included_instruments]]
    return path, get_input(channels, info)


def extract_style(model, input):
    max_n_bars = 1000 // input[1][1].shape[0]
    mode, bpm, pitched_channels, instruments_features, unpitched_channels = prepare_input(input, max_n_bars)
    with torch.no_grad():
        style, melody, rhythm = model.extract_style(mode, bpm, pitched_channels, instruments_features, unpitched_channels)
    return style.detach(), melody.detach(), rhythm.detach()


def channel_slots(instruments):
    """MIDI channels 0-8, 10-15 for the pitched instruments in order; 9 is percussion (:78-86)."""
    slots = [i for i in range(16) if i != 9]
    pitched = [{'channel_id': slot, 'instrument_id': int(program)} for slot, program in zip(slots, instruments)]
    return pitched, {'channel_id': 9, 'instrument_id': -1}


def save(cc, pitched_channels, unpitched_channels, instruments, save_path):
    channels_info, unpitched_info = channel_slots(instruments)
    channels_info = channels_info[:pitched_channels.shape[1]]     # (for numpy rolls shape[1] is the bar count: reference quirk)
    os.makedirs(os.path.dirname(save_path) or '.', exist_ok=True)
    if len(pitched_channels.shape) == 6:                       # numpy rolls straight from get_input
        pitched_channels = torch.tensor(pitched_channels, dtype=torch.float).unsqueeze(0).to(device)
        if unpitched_channels is not None:
            unpitched_channels = torch.tensor(unpitched_channels, dtype=torch.float).unsqueeze(0).to(device)
    mid = decode_midi(cc, channels_info, pitched_channels, unpitched_info, unpitched_channels)
    mid.save(save_path)


def select_instruments(instruments_pred, n_instruments):
    """Top-n of the 41 logits -> (GM programs of the pitched picks, percussion picked?).  A lone
    percussion pick is widened by one so at least one pitched instrument plays (:105-116)."""
    ranked = np.argsort(-np.asarray(instruments_pred))
    picked = ranked[:n_instruments]
    if len(picked) == 1 and picked[0] == percussion_id:
        picked = ranked[:n_instruments + 1]
    unpitched = percussion_id in picked
    programs = [_instrument_categories[i] for i in picked if i != percussion_id]
    return programs, unpitched


def apply_style(model, info, style, melody, rhythm, n_instruments, save_path):
    with torch.no_grad():
        instruments_pred, mode, bpm = model.predict_song_info(style, rhythm)
    info['tempo'] = smf.bpm2tempo(round(float(bpm)))
    instruments, unpitched = select_instruments(instruments_pred.detach().cpu().numpy()[0], n_instruments)
    info['scale']['mode'] = major_mode if int(mode[0].argmax()) == 0 else minor_mode
    cc = ChannelConverter(info)
    instruments_features = torch.tensor(encode_instruments(instruments), dtype=torch.float).to(device).unsqueeze(0)
    with torch.no_grad():
        pitched_pred, unpitched_pred = model.apply_style(style, melody, rhythm, instruments_features, unpitched)
    save(cc, pitched_pred, unpitched_pred, instruments, save_path)


def combine_info(style_info, melody_info):
    return {
        'time_signature': melody_info['time_signature'],
        'scale': style_info['scale'],
        'ticks_per_beat': melody_info['ticks_per_beat'],
        'ticks_per_bar': melody_info['ticks_per_bar'],
        'tempo': style_info['tempo'],
    }


def decode_rolls(channel_converter, channels_info, pitched_rolls, unpitched_channel_info=None, unpitched_roll=None):
    """Hard (C,R,T,10,56,5) / (R,T,10,47,2) numpy rolls -> MidiFile; gaps are capped at one second (:145-158)."""
    channels = [channel_converter.vchannel2channel(channel_info, roll)
                for channel_info, roll in zip(channels_info, pitched_rolls)]
    if unpitched_roll is not None:
        channels.append(channel_converter.vchannel2channel(unpitched_channel_info, unpitched_roll))
    return create_midi(channel_converter.info, *channels, max_delta_time=1)


def decode_midi(channel_converter, channels_info, pitched_channels, unpitched_channel_info=None, unpitched_channels=None):
    pitched_rolls = hard_output(pitched_channels).cpu().detach().numpy()[0]
    unpitched_roll = None
    if unpitched_channels is not None:
        unpitched_roll = hard_output(unpitched_channels).cpu().detach().numpy()[0, 0]
    return decode_rolls(channel_converter, channels_info, pitched_rolls, unpitched_channel_info, unpitched_roll)
