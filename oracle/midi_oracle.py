"""TEST INFRASTRUCTURE — scalar CPU restatement of the reference's MIDI <-> piano-roll path.

One Python object per note, one loop iteration per message, `fractions.Fraction` beat positions and
list-of-arrays bars: the shape of the reference's algorithm (style/midi_conversion.py:31-232,244-283,
349-609; style/midi.py:120-168; style/data.py:66-114; style/utils/math.py:14-19), restated so the
product's column-store implementation (music-style-transfer_amd/style/midi_conversion.py, data.py)
can be checked note for note, index for index, tick for tick.

Pinning.  `mido` is absent from the build container, so the reference's own module cannot be imported
and this oracle cannot be diffed against it directly.  It is pinned by the reference's example
outputs (tests/golden/midi/*.mid, written by the reference's `create_midi`): files without
time-capped gaps are fixed points of read -> roll -> write under this oracle, and the SMF byte layer
round-trips all of them.  Beyond that: PARITY UNPINNED (stated in DESIGN.md).

Only tests/ may import this module.  Parsed files come in as lists of message objects with mido's
attribute names (`type`, `time`, `channel`, `note`, `velocity`, `program`, `control`, `value`,
`tempo`, `numerator`, `denominator`, `key`).
"""
from fractions import Fraction
import math

import numpy as np

DEFAULT_TEMPO, DEFAULT_VOLUME, MAX_VOLUME, MAX_VELOCITY = 500000, 96, 127, 127
KEY_NAMES = ['C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B']
IGNORED = {'smpte_offset', 'midi_port', 'sysex', 'end_of_track', 'track_name', 'copyright', 'lyrics', 'marker',
           'sequencer_specific', 'channel_prefix', 'text', 'instrument_name', 'aftertouch', 'polytouch', 'cue_marker',
           'unknown_meta', 'sequence_number'}
KNOWN = IGNORED | {'note_on', 'note_off', 'time_signature', 'key_signature', 'set_tempo', 'program_change',
                   'control_change', 'pitchwheel'}


class FormatError(Exception):
    pass


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def is_pitched(instrument_id):                           # style/midi.py:96-101
    return 0 <= instrument_id <= 119


# ------------------------------------------------------------------------------------------------ modes
class OracleMode:                                         # style/scales.py:27-91
    def __init__(self, steps, shift=0):
        self.steps, self.shift = list(steps), shift
        self.tonic_intervals = [0]
        for s in self.steps:
            self.tonic_intervals.append(self.tonic_intervals[-1] + s)
        self.absolute_intervals = self.tonic_intervals[:7]
        self.degree_of = {}
        last = 1
        for semitone in range(12):
            if semitone in self.absolute_intervals:
                last = self.absolute_intervals.index(semitone) + 1
                self.degree_of[semitone] = last
            else:
                self.degree_of[semitone] = last + .5

    def get_degree(self, interval):
        return self.degree_of[interval % 12]


MAJOR = OracleMode([2, 2, 1, 2, 2, 2, 1])
MINOR = OracleMode([2, 1, 2, 2, 1, 2, 2], shift=-2)        # create_mode(major, -2), style/scales.py:100-102,125
MAJOR_CHROMATIC = {1.5: 'flat', 2.5: 'flat', 4.5: 'sharp', 5.5: 'sharp', 6.5: 'flat'}   # midi_conversion.py:235-241


def spell(note_id, mode, tonic):
    """MIDI note number -> (scale octave, degree, accidental)  (midi_conversion.py:244-265,312-320)."""
    octave, pitch_class = divmod(note_id, 12)
    octave -= 1
    interval = pitch_class - KEY_NAMES.index(tonic)
    degree = mode.get_degree(interval)
    accidental = 'none'
    if not isinstance(degree, int):
        in_major = MAJOR.get_degree(interval + MAJOR.tonic_intervals[(mode.shift - MAJOR.shift) % 7])   # scales.py:117-121
        accidental = MAJOR_CHROMATIC[in_major]
        degree = math.floor(degree) if accidental == 'sharp' else math.ceil(degree)
    if interval < 0:
        octave -= 1
    return octave, degree, accidental


def unspell(octave, degree, accidental, mode, tonic):
    """Inverse of spell -> MIDI note number (midi_conversion.py:268-283,323-327)."""
    interval = mode.absolute_intervals[degree - 1] + KEY_NAMES.index(tonic)
    if accidental == 'sharp':
        interval += 1
    elif accidental == 'flat':
        interval -= 1
    if interval < 0:
        octave, interval = octave - 1, interval + 12
    elif interval >= 12:
        octave, interval = octave + 1, interval - 12
    return 12 * (octave + 1) + interval


# ------------------------------------------------------------------------------------------ file -> notes
def read_song(tracks, ticks_per_beat):
    """tracks (lists of messages with delta times) -> (channels, info)  (midi_conversion.py:31-66,117-232)."""
    merged = []
    for track in tracks:
        now = 0
        for m in track:
            now += m.time
            c = Obj(**m.__dict__)
            c.time = now
            merged.append(c)
    merged.sort(key=lambda m: m.time)
    song_msgs, by_channel = [], {}
    for m in merged:
        if m.time > 1e7:
            continue
        if hasattr(m, 'channel'):
            by_channel.setdefault(m.channel, []).append(m)
        else:
            song_msgs.append(m)
    every = [m for ms in by_channel.values() for m in ms]
    ons = [m.time for m in every if m.type == 'note_on' and m.velocity > 0]
    first, last = min(ons), max(ons)
    duration = max(m.time for m in every)
    info = dict(ticks_per_beat=ticks_per_beat, time_signature=dict(numerator=4, denominator=4, value=1.), key=None,
                duration=duration)
    tempo, since, held = DEFAULT_TEMPO, 0, {}
    for m in song_msgs:
        if m.type in IGNORED:
            continue
        if m.type == 'time_signature':
            ts = dict(numerator=m.numerator, denominator=m.denominator, value=m.numerator / m.denominator)
            if ts != info['time_signature']:
                if first <= m.time <= last:
                    raise FormatError('Time signature changed')
                info['time_signature'] = ts
        elif m.type == 'key_signature':
            if m.key != info['key']:
                if first <= m.time <= last:
                    raise FormatError('Key signature changed')
                info['key'] = m.key
        elif m.type == 'set_tempo':
            if m.tempo != tempo:
                held[tempo] = held.get(tempo, 0) + m.time - since
                tempo, since = m.tempo, m.time
        elif m.type not in KNOWN:
            raise FormatError(m.type)
    info['ticks_per_bar'] = int(ticks_per_beat * info['time_signature']['numerator'])
    info['n_bars'] = duration / info['ticks_per_bar']
    info['n_beats'] = info['time_signature']['numerator']
    held[tempo] = held.get(tempo, 0) + duration - since
    held = {k: v for k, v in held.items() if v}
    info['tempo2time'] = held
    info['tempo'] = max(held.items(), key=lambda kv: kv[1])[0]
    info['bpm'] = round(60e6 / info['tempo'])
    channels = []
    for msgs in by_channel.values():
        channel_id = msgs[0].channel
        instrument = -1 if channel_id == 9 else 0
        volume = DEFAULT_VOLUME
        per_instrument = {}
        for m in msgs:
            if m.type in IGNORED:
                continue
            if m.type not in KNOWN:
                raise FormatError(m.type)
            if m.type == 'program_change':
                instrument = -1 if channel_id == 9 else m.program
            elif m.type == 'control_change' and m.control == 7:
                volume = m.value
            elif m.type in ('note_on', 'note_off'):
                velocity = m.velocity * volume / (MAX_VELOCITY * MAX_VOLUME)
                kind = 'note_off' if velocity == 0 else m.type
                per_instrument.setdefault(instrument, []).append(Obj(type=kind, note=m.note, velocity=velocity, time=m.time))
        for instrument_id, ms in per_instrument.items():
            if any(x.type == 'note_on' for x in ms):
                channels.append(dict(channel_id=channel_id, instrument_id=instrument_id, messages=ms))
    return channels, info


def notes_of(channel):
    """note_on/off messages -> notes with end times (midi_conversion.py:366-401)."""
    notes, sounding = [], {}
    for m in channel['messages']:
        if m.note in sounding:
            sounding.pop(m.note).end_time = m.time
        if m.type == 'note_on':
            n = Obj(note_id=m.note, velocity=m.velocity, time=m.time, end_time=m.time)
            notes.append(n)
            sounding[m.note] = n
    for n in notes:
        n.duration = n.end_time - n.time
    return notes


def round_number(number, precision):                      # style/utils/math.py:14-19
    down = number % precision
    up = abs(down - precision)
    if down < up:
        return number - down, down
    return number + up, -up


class Converter:
    def __init__(self, info, beat_divisors=(8, 3), n_octaves=8, min_percussion=35, max_percussion=81):
        self.info, self.beat_divisors = info, beat_divisors
        self.fractions = sorted({Fraction(i, d) for d in beat_divisors for i in range(d)})
        self.n_notes, self.lo, self.hi = n_octaves * 7, min_percussion, max_percussion

    @property
    def n_bars(self):
        return math.ceil(self.info['n_bars'])

    def roll(self, notes, instrument_id, mode=None, tonic=None):
        """notes -> dense roll (midi_conversion.py:403-446,490-516,597-609)."""
        pitched = is_pitched(instrument_id)
        tpb, tpbar = self.info['ticks_per_beat'], self.info['ticks_per_bar']
        rows, n_feat = (self.n_notes, 5) if pitched else (self.hi - self.lo + 1, 2)
        bars = [[np.zeros([len(self.fractions), rows, n_feat]) for _ in range(self.info['n_beats'])]
                for _ in range(self.n_bars + 1)]
        for n in notes:
            best = None
            for divisor in self.beat_divisors:
                snapped, err = round_number(n.time, tpb / divisor)
                if best is None or abs(err) < best[0]:
                    best = (abs(err), snapped, divisor)
            _, snapped, divisor = best
            n.qtime = int(snapped)
            n.qduration = n.end_time - n.qtime
            bar, rest = divmod(n.qtime, tpbar)
            beat, ticks = divmod(rest, tpb)
            n.bar, n.beat = int(bar), int(beat)
            n.beat_fraction = Fraction(int(ticks // (tpb / divisor)), divisor)
            if pitched:
                n.scale_octave, n.scale_degree, n.accidental = spell(n.note_id, mode, tonic)
                row = n.scale_octave * 7 + n.scale_degree - 1
                if not 0 <= row < self.n_notes:
                    continue
            else:
                if not self.lo <= n.note_id <= self.hi:
                    continue
                row = n.note_id - self.lo
            features = [n.qduration / tpb, n.velocity]
            if pitched:
                features += {'flat': [1., 0., 0.], 'none': [0., 1., 0.], 'sharp': [0., 0., 1.]}[n.accidental]
            cell = np.zeros([len(self.fractions), rows, n_feat])
            cell[self.fractions.index(n.beat_fraction)][row] = features
            bars[n.bar][n.beat] = np.maximum(bars[n.bar][n.beat], cell)
        return np.stack([np.stack(b) for b in bars])

    def unroll(self, roll, instrument_id, mode=None, tonic=None):
        """dense roll -> time-ordered note_on/note_off list (midi_conversion.py:448-488,518-566)."""
        pitched = is_pitched(instrument_id)
        tpb, tpbar = self.info['ticks_per_beat'], self.info['ticks_per_bar']
        out = []
        for b, bar in enumerate(roll):
            for t, beat in enumerate(bar):
                for fraction, cells in zip(self.fractions, beat):
                    for row in np.nonzero(cells[:, 1])[0]:
                        cell = cells[row]
                        if pitched:
                            duration, velocity, flat, natural, sharp = cell
                            accidental = 'flat' if flat else 'none' if natural else 'sharp' if sharp else 'none'
                            note_id = unspell(row // 7, row % 7 + 1, accidental, mode, tonic)
                        else:
                            duration, velocity = cell
                            note_id = int(row + self.lo)
                        qduration = int(duration * tpb)
                        time = b * tpbar + t * tpb + int(fraction * tpb)
                        out.append(Obj(type='note_on', note=int(note_id), velocity=velocity, time=time))
                        out.append(Obj(type='note_off', note=int(note_id), velocity=0, time=time + qduration))
        return sorted(out, key=lambda m: m.time)


def key_seconds(notes, info):
    """{pitch class: seconds of velocity-weighted sound} (midi_conversion.py:337-346)."""
    ticks = {}
    for n in notes:
        ticks[n.note_id % 12] = ticks.get(n.note_id % 12, 0) + n.duration * n.velocity
    scale = info['tempo'] * 1e-6 / info['ticks_per_beat']
    return {k: v * scale for k, v in ticks.items()}


def track_events(info, channels, max_delta_time=math.inf):
    """[(delta, kind, channel, a, b)] of the single output track (style/midi.py:120-168); `channels` =
    [(channel_id, instrument_id, messages)]."""
    cap = max_delta_time / (info['tempo'] * 1e-6 / info['ticks_per_beat'])
    if math.isfinite(cap):
        cap = int(cap)
    head = [(0, 'time_signature', None, info['time_signature']['numerator'], info['time_signature']['denominator']),
            (0, 'set_tempo', None, info['tempo'], None)]
    notes = []
    for channel_id, instrument_id, messages in channels:
        if channel_id != 9:
            head.append((0, 'program_change', channel_id, instrument_id, None))
        for m in messages:
            v = int(m.velocity * MAX_VELOCITY)
            assert v <= 127
            notes.append((m.time, m.type, channel_id, m.note, v))
    notes.sort(key=lambda e: e[0])
    notes.append((info.get('duration', notes[-1][0] + info['ticks_per_bar']), 'end_of_track', None, None, None))
    now, body = 0, []
    for time, kind, channel_id, a, b in notes:
        delta = min(time - now, cap)
        now = time
        body.append((max(0, delta), kind, channel_id, a, b))
    return head + body
