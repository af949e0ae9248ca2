"""MIDI constants, instrument vocabulary and file creation (style/midi.py:19-64,90-108,120-168),
on this package's own SMF implementation (style/smf.py) instead of `mido`."""
import math

from style import smf
from style.smf import Message, MetaMessage, MidiFile

default_tempo = 500000
default_volume = 96
max_volume = 127
max_velocity = 127

# the 40 most popular General-MIDI programs (0-based), reference order (style/midi.py:23-64)
popular_instruments = [0, 25, 48, 33, 1, 27, 49, 29, 35, 30, 50, 24, 5, 4, 32, 52, 26, 18, 28, 89, 65, 53, 61, 2, 17, 73,
                       54, 62, 16, 39, 34, 51, 90, 56, 66, 38, 11, 81, 3, 57]
_GM_FAMILIES = ['Piano', 'Chromatic Percussion', 'Organ', 'Guitar', 'Bass', 'Strings', 'Ensemble', 'Brass', 'Reed', 'Pipe',
                'Synth Lead', 'Synth Pad', 'Synth Effects', 'Ethnic', 'Percussive', 'Sound effects']
program2group = {p: _GM_FAMILIES[p // 8] for p in range(128)}
program2instrument = {p: f'GM program {p + 1}' for p in range(128)}     # display names only
program2instrument[-1] = 'Percussion'


def get_instrument_id(program, channel=0):
    return -1 if channel == 9 else program


def is_sound_effect(instrument_id):
    return instrument_id > 119


def is_pitched(instrument_id):
    return instrument_id >= 0 and not is_sound_effect(instrument_id)


def load_midi_from_file(path):
    """None for unreadable files, like the reference (style/midi.py:104-108)."""
    try:
        return MidiFile(path)
    except (OSError, ValueError, KeyError, EOFError, IndexError):
        return None


def create_midi(info, *instruments, max_delta_time=math.inf):
    """Single-track format-1 file: time signature, tempo, program changes, then all notes sorted by
    time with delta times capped at `max_delta_time` seconds (style/midi.py:120-168)."""
    cap = smf.second2tick(max_delta_time, info['ticks_per_beat'], info['tempo'])
    if math.isfinite(cap):
        cap = int(cap)
    mid = MidiFile(ticks_per_beat=info['ticks_per_beat'])
    track = []
    mid.tracks.append(track)
    ts = info['time_signature']
    track.append(MetaMessage('time_signature', 0, numerator=ts['numerator'], denominator=ts['denominator']))
    track.append(MetaMessage('set_tempo', 0, tempo=info['tempo']))
    notes = []
    for ins in instruments:
        if ins['channel_id'] != 9:
            track.append(Message('program_change', 0, channel=ins['channel_id'], program=ins['instrument_id']))
        for m in ins['messages']:
            velocity = int(m.velocity * max_velocity)
            assert velocity <= 127, (velocity, m.velocity)
            notes.append(Message(m.type, m.time, channel=ins['channel_id'], note=m.note, velocity=velocity))
    notes.sort(key=lambda m: m.time)                       # stable, like sorted()
    duration = info.get('duration', notes[-1].time + info['ticks_per_bar'])
    notes.append(MetaMessage('end_of_track', duration))
    now = 0
    for m in notes:
        m = m.copy()
        delta = min(m.time - now, cap)
        now = m.time
        m.time = max(0, delta)
        track.append(m)
    return mid
