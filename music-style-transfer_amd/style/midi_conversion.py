"""MIDI messages <-> notes <-> dense piano-roll (`vchannel`), host side, bit-exact indexing.

Restates style/midi_conversion.py:31-232 (read_midi and its helpers) and :349-609
(ChannelConverter) on column stores: where the reference walks lists of `Note` dataclasses one
Python object at a time, a channel here is a `NoteTable` (one numpy column per attribute) and every
conversion stage is a handful of array operations.  The arithmetic that decides an index is kept
operation for operation (float64 `%` / `//` of the quantiser, `int()` truncations, the element-wise
`max` merge), so piano-roll indices and MIDI ticks are identical to the reference's:

  beat fractions  sorted {i/8} U {i/3} = [0,1/8,1/4,1/3,3/8,1/2,5/8,2/3,3/4,7/8]   (:358-364)
  pitched note    octave*7 + (degree-1) in [0,56), out-of-range notes dropped        (:597-604)
  percussion      key-35 in [0,47)                                                   (:605-609)
  features        [duration(beats), velocity, flat, natural, sharp]                  (:501-511)
  bars            ceil(n_bars)+1                                                     (:492-493)

`mido` is replaced by style/smf.py.
"""
from collections import namedtuple
from fractions import Fraction
import math

import numpy as np

from style import smf
from style.exceptions import MidiFormatError
from style.midi import (get_instrument_id, program2instrument, is_pitched, default_tempo, default_volume, max_volume,
                        max_velocity)
from style.scales import key_names, key2interval, get_relative_degree, major_mode

max_msg_time = 1e7

messages_to_include = {'note_on', 'note_off', 'time_signature', 'key_signature', 'set_tempo', 'program_change',
                       'control_change', 'pitchwheel'}
messages_to_ignore = {'smpte_offset', 'midi_port', 'sysex', 'end_of_track', 'track_name', 'copyright', 'lyrics', 'marker',
                      'sequencer_specific', 'channel_prefix', 'text', 'instrument_name', 'aftertouch', 'polytouch',
                      'cue_marker', 'unknown_meta', 'sequence_number'}
known_messages = messages_to_include | messages_to_ignore

FLAT, NATURAL, SHARP, NO_ACCIDENTAL = 0, 1, 2, -1         # column order of the accidental one-hot
accidental_names = {FLAT: 'flat', NATURAL: 'none', SHARP: 'sharp', NO_ACCIDENTAL: None}
# chromatic degree inside the major mode -> spelling (style/midi_conversion.py:235-241)
degree2accidental = {1.5: FLAT, 2.5: FLAT, 4.5: SHARP, 5.5: SHARP, 6.5: FLAT}

NoteMessage = namedtuple('NoteMessage', 'type note time velocity')


def check(condition, error_message):
    if not condition:
        raise MidiFormatError(error_message)


class NoteMessages:
    """note_on / note_off stream of one (channel, instrument): columns `on`, `note`, `time` (absolute
    ticks), `velocity` (0..1).  Iterates as NoteMessage tuples, so `create_midi` and `len()` users see
    the reference's list of messages."""

    def __init__(self, on, note, time, velocity):
        self.on = np.asarray(on, dtype=bool)
        self.note = np.asarray(note, dtype=np.int64)
        self.time = np.asarray(time, dtype=np.int64)
        self.velocity = np.asarray(velocity)

    def __len__(self):
        return len(self.note)

    def __iter__(self):
        for on, note, time, velocity in zip(self.on.tolist(), self.note.tolist(), self.time.tolist(), self.velocity):
            yield NoteMessage('note_on' if on else 'note_off', note, time, velocity)

    def sorted_by_time(self):
        order = np.argsort(self.time, kind='stable')
        return NoteMessages(self.on[order], self.note[order], self.time[order], self.velocity[order])


class NoteTable:
    """Notes of one channel, one column per attribute of the reference's `Note` (:286-309)."""
    int_columns = ('note_id', 'time', 'end_time', 'scale_octave', 'scale_degree', 'accidental', 'qtime', 'qduration', 'bar',
                   'beat', 'fraction')

    def __init__(self, n=0, **columns):
        for name in self.int_columns:
            setattr(self, name, np.zeros(n, dtype=np.int64))
        self.velocity = np.zeros(n)
        for name, values in columns.items():
            setattr(self, name, np.asarray(values))

    def __len__(self):
        return len(self.velocity)

    def copy(self):
        t = NoteTable()
        t.__dict__.update({k: v.copy() for k, v in self.__dict__.items()})
        return t

    def take(self, index):
        t = NoteTable()
        t.__dict__.update({k: v[index] for k, v in self.__dict__.items()})
        return t

    @staticmethod
    def concat(tables):
        t = NoteTable()
        for name in tables[0].__dict__:
            setattr(t, name, np.concatenate([getattr(x, name) for x in tables]))
        return t

    @property
    def duration(self):
        return self.end_time - self.time


# ------------------------------------------------------------------------------------- MIDI -> channels
def get_channel_info(channel):
    info = {k: v for k, v in channel.items() if k != 'messages'}
    info['pitched'] = is_pitched(info['instrument_id'])
    return info


def merge_tracks(tracks, apply_global_timing=False):
    """All messages of all tracks; with global timing, delta times become absolute ticks and the
    result is ordered by them (stable: track order, then position in the track)."""
    msgs = []
    for track in tracks:
        now = 0
        for msg in track:
            msg = msg.copy()
            if apply_global_timing:
                now += msg.time
                msg.time = now
            msgs.append(msg)
    if apply_global_timing:
        msgs.sort(key=lambda m: m.time)
    return msgs


def split_channels(mid):
    """(song-level messages, [messages of channel a, of channel b, ...]) in order of first appearance."""
    global_messages, channels = [], {}
    for msg in merge_tracks(mid.tracks, apply_global_timing=True):
        if msg.time > max_msg_time:
            continue
        if hasattr(msg, 'channel'):
            channels.setdefault(msg.channel, []).append(msg)
        else:
            global_messages.append(msg)
    return global_messages, list(channels.values())


def get_midi_info(global_messages, channels, ticks_per_beat):
    """Song constants; a time/key signature change between the first and the last note-on is a
    MidiFormatError, the tempo is the one held for the most ticks (:117-180)."""
    channel_messages = [m for channel in channels for m in channel]
    on_times = [m.time for m in channel_messages if m.type == 'note_on' and m.velocity > 0]
    first_note, last_note = min(on_times), max(on_times)
    duration = max(m.time for m in channel_messages)
    info = {
        'ticks_per_beat': ticks_per_beat,
        'time_signature': {'numerator': 4, 'denominator': 4, 'value': 1.},
        'key': None,
        'duration': duration,
    }
    tempo, tempo_since, tempo2ticks = default_tempo, 0, {}
    for msg in global_messages:
        kind = msg.type
        if kind in messages_to_ignore:
            continue
        if kind == 'time_signature':
            ts = {'numerator': msg.numerator, 'denominator': msg.denominator, 'value': msg.numerator / msg.denominator}
            if ts != info['time_signature']:
                check(not first_note <= msg.time <= last_note, "Time signature changed")
                info['time_signature'] = ts
        elif kind == 'key_signature':
            if msg.key != info['key']:
                check(not first_note <= msg.time <= last_note, "Key signature changed")
                info['key'] = msg.key
        elif kind == 'set_tempo':
            if msg.tempo != tempo:
                tempo2ticks[tempo] = tempo2ticks.get(tempo, 0) + msg.time - tempo_since
                tempo, tempo_since = msg.tempo, msg.time
        elif kind not in known_messages:
            raise MidiFormatError(f"Unknown message type: {kind}")
    info['ticks_per_bar'] = int(ticks_per_beat * info['time_signature']['numerator'])
    info['n_bars'] = duration / info['ticks_per_bar']
    info['n_beats'] = info['time_signature']['numerator']
    tempo2ticks[tempo] = tempo2ticks.get(tempo, 0) + duration - tempo_since
    info['tempo2time'] = {k: v for k, v in tempo2ticks.items() if v}
    info['tempo'] = max(info['tempo2time'].items(), key=lambda kv: kv[1])[0]
    info['bpm'] = round(smf.tempo2bpm(info['tempo']))
    return info


def group_channel_messages(channel_messages, channel_id):
    """{instrument id: NoteMessages}; program changes switch the instrument, controller 7 scales the
    velocities that follow, a note-on of velocity 0 is a note-off (:183-209)."""
    instrument_id = get_instrument_id(0, channel_id)
    volume = default_volume
    rows = {}
    for msg in channel_messages:
        kind = msg.type
        if kind in messages_to_ignore:
            continue
        if kind not in known_messages:
            raise MidiFormatError(f"Unknown message type: {kind}")
        if kind == 'program_change':
            instrument_id = get_instrument_id(msg.program, channel_id)
        elif kind == 'control_change':
            if msg.control == 7:
                volume = msg.value
        elif kind == 'note_on' or kind == 'note_off':
            velocity = msg.velocity * volume / (max_velocity * max_volume)
            on = kind == 'note_on' and velocity != 0
            if on:
                assert 0 < velocity <= 1, msg.velocity
            rows.setdefault(instrument_id, []).append((on, msg.note, msg.time, velocity))
    return {k: NoteMessages(*zip(*v)) for k, v in rows.items()}


def read_midi(mid):
    """MidiFile -> ([{channel_id, instrument_id, instrument_name, messages}], info) (:216-232)."""
    global_messages, channels_messages = split_channels(mid)
    info = get_midi_info(global_messages, channels_messages, mid.ticks_per_beat)
    channels = []
    for channel_messages in channels_messages:
        channel_id = channel_messages[0].channel
        for instrument_id, messages in group_channel_messages(channel_messages, channel_id).items():
            if messages.on.any():
                channels.append({
                    'channel_id': channel_id,
                    'instrument_id': instrument_id,
                    'instrument_name': program2instrument[instrument_id],
                    'messages': messages,
                })
    return channels, info


# --------------------------------------------------------------------------------- key-relative spelling
def note2scale_loc(key, octave, mode, tonic):
    """One (key name, octave) -> dict(octave, degree, accidental name) in the scale (mode, tonic) (:244-265)."""
    o, d, a = _scale_table(mode, tonic)
    k = key2interval[key]
    return dict(octave=octave + int(o[k]), degree=int(d[k]), accidental=accidental_names[int(a[k])])


def _scale_table(mode, tonic):
    """For each of the 12 pitch classes: (octave adjustment, degree 1..8, accidental code).
    A chromatic tone is spelled as the neighbouring scale degree: sharp -> the degree below,
    flat -> the degree above (which may be 8 = degree 1 of the next octave)."""
    tonic_interval = key2interval[tonic]
    octave_adj, degrees, accidentals = np.zeros(12, np.int64), np.zeros(12, np.int64), np.zeros(12, np.int64)
    for k in range(12):
        interval = k - tonic_interval
        degree = mode.get_degree(interval)
        accidental = NATURAL
        if not isinstance(degree, int):
            accidental = degree2accidental[get_relative_degree(interval, mode, major_mode)]
            degree = math.floor(degree) if accidental == SHARP else math.ceil(degree)
        octave_adj[k], degrees[k], accidentals[k] = -1 if interval < 0 else 0, degree, accidental
    return octave_adj, degrees, accidentals


def scale_loc2key_octave(octave, degree, mode, tonic, accidental=None):
    """Inverse spelling for one note (:268-283)."""
    interval = mode.absolute_intervals[degree - 1] + key2interval[tonic]
    interval += {'sharp': 1, 'flat': -1}.get(accidental, 0)
    octave += interval // 12
    return key_names[interval % 12], octave


def note_id2key_octave(note_id, pitched=True):
    if pitched:
        octave, interval = divmod(note_id, 12)
        return key_names[interval], octave - 1
    return str(note_id), None


def _key_weights(info, nchannel):
    """Seconds of velocity-weighted sound per pitch class, in note order (sequential accumulation,
    like the reference's `sum()` over each group)."""
    notes = nchannel['notes']
    ticks = np.bincount(notes.note_id % 12, weights=notes.duration * notes.velocity, minlength=12)
    present = np.bincount(notes.note_id % 12, minlength=12) > 0
    return smf.tick2second(ticks, info['ticks_per_beat'], info['tempo']), present


def get_keys_dist(info, nchannel):
    """{key name: seconds, 'instrument': name} for the keys that occur (:337-343)."""
    seconds, present = _key_weights(info, nchannel)
    dist = {key_names[k]: float(seconds[k]) for k in np.flatnonzero(present)}
    dist['instrument'] = nchannel['instrument_name']
    return dist


# ---------------------------------------------------------------------------------------- the converter
class ChannelConverter:
    def __init__(self, info, beat_divisors=(8, 3), n_octaves=8, min_percussion=35, max_percussion=81):
        self.info = info
        self.beat_divisors = beat_divisors
        self.n_octaves = n_octaves
        self.min_percussion = min_percussion
        self.max_percussion = max_percussion
        self.beat_fractions = sorted({Fraction(i, d) for d in beat_divisors for i in range(d)})
        self.beat_fraction2idx = {f: i for i, f in enumerate(self.beat_fractions)}
        # (divisor position, quants) -> fraction index, -1 where the reference would raise KeyError
        self._quant2idx = np.full((len(beat_divisors), max(beat_divisors) + 1), -1, dtype=np.int64)
        for j, d in enumerate(beat_divisors):
            for q in range(d):
                self._quant2idx[j, q] = self.beat_fraction2idx[Fraction(q, d)]
        self.n_notes = n_octaves * 7
        self.n_unpitched = max_percussion - min_percussion + 1
        self.n_note_features = 5        # duration, velocity, flat, natural, sharp
        self.n_unpitched_features = 2   # duration, velocity

    # .................................................................................. messages -> notes
    def channel2nchannel(self, channel):
        """A note ends at the next message (on or off) of the same pitch, or has zero length (:366-401)."""
        nchannel = {k: v for k, v in channel.items() if k != 'messages'}
        m = channel['messages']
        order = np.argsort(m.note, kind='stable')
        pitch, when = m.note[order], m.time[order]
        end = when.copy()
        same = pitch[1:] == pitch[:-1]
        end[:-1][same] = when[1:][same]
        end_time = np.empty_like(end)
        end_time[order] = end
        on = m.on
        nchannel['notes'] = NoteTable(int(on.sum()), note_id=m.note[on], time=m.time[on], end_time=end_time[on],
                                      velocity=np.asarray(m.velocity[on], dtype=np.float64))
        return nchannel

    def nchannel2kchannel(self, nchannel, in_place=False):
        kchannel = nchannel if in_place else dict(nchannel, notes=nchannel['notes'].copy())
        notes = kchannel['notes']
        if is_pitched(nchannel['instrument_id']):
            octave_adj, degrees, accidentals = _scale_table(self.mode, self.key)
            pc = notes.note_id % 12
            notes.scale_octave = notes.note_id // 12 - 1 + octave_adj[pc]
            notes.scale_degree = degrees[pc]
            notes.accidental = accidentals[pc]
        else:
            notes.scale_octave = np.zeros(len(notes), np.int64)
            notes.scale_degree = np.zeros(len(notes), np.int64)
            notes.accidental = np.full(len(notes), NO_ACCIDENTAL, np.int64)
        return kchannel

    def kchannel2qchannel(self, kchannel, in_place=False):
        """Snap onsets to the nearest 1/8 or 1/3 of a beat; the first divisor wins ties (:416-446).
        round_number (style/utils/math.py:14-19) in float64: r = t % g; down if r < |r - g| else up."""
        qchannel = kchannel if in_place else dict(kchannel, notes=kchannel['notes'].copy())
        notes = qchannel['notes']
        tpb, tpbar = self.info['ticks_per_beat'], self.info['ticks_per_bar']
        time = notes.time.astype(np.float64)
        best_err = best_q = best_div = None
        for j, divisor in enumerate(self.beat_divisors):
            grid = tpb / divisor
            down = np.mod(time, grid)
            up = np.abs(down - grid)
            snapped = np.where(down < up, time - down, time + up)
            err = np.where(down < up, down, up)                # |time error|
            if best_err is None:
                best_err, best_q, best_div = err, snapped, np.full(len(time), j, np.int64)
            else:
                better = err < best_err
                best_err = np.where(better, err, best_err)
                best_q = np.where(better, snapped, best_q)
                best_div = np.where(better, j, best_div)
        notes.qtime = best_q.astype(np.int64)                  # int(): truncation
        notes.qduration = notes.end_time - notes.qtime
        notes.bar, rest = np.divmod(notes.qtime, tpbar)
        notes.beat, ticks = np.divmod(rest, tpb)
        grids = np.array([tpb / d for d in self.beat_divisors])[best_div]
        quants = np.floor_divide(ticks.astype(np.float64), grids).astype(np.int64)
        notes.fraction = self._quant2idx[best_div, quants]
        if (notes.fraction < 0).any():
            raise KeyError('beat fraction outside the grid')
        return qchannel

    # ................................................................................ notes -> piano-roll
    def note2idx(self, notes, pitched):
        """(row index per note, mask of the notes that fit the roll)."""
        if pitched:
            idx = notes.scale_octave * 7 + (notes.scale_degree - 1)
            return idx, (idx >= 0) & (idx < self.n_notes)
        idx = notes.note_id - self.min_percussion
        return idx, (notes.note_id >= self.min_percussion) & (notes.note_id <= self.max_percussion)

    def qchannel2vchannel(self, qchannel):
        """(n_bars+1, n_beats, 10, notes, features) float64; colliding notes merge by element-wise max."""
        pitched = is_pitched(qchannel['instrument_id'])
        notes = qchannel['notes']
        n_rows = self.n_notes if pitched else self.n_unpitched
        roll = np.zeros([self.n_bars + 1, self.info['n_beats'], len(self.beat_fractions), n_rows, self.n_features(pitched)])
        idx, fits = self.note2idx(notes, pitched)
        notes, idx = notes.take(fits), idx[fits]
        features = np.zeros([len(notes), self.n_features(pitched)])
        features[:, 0] = notes.qduration / self.info['ticks_per_beat']
        features[:, 1] = notes.velocity
        if pitched:
            features[np.arange(len(notes)), 2 + notes.accidental] = 1.
        np.maximum.at(roll, (notes.bar, notes.beat, notes.fraction, idx), features)
        return roll

    def vchannel2qchannel(self, channel_info, vchannel):
        """Every cell with non-zero velocity is a note, in (bar, beat, fraction, row) order (:518-566).
        Arithmetic runs in the roll's own dtype (float32 for model outputs)."""
        pitched = is_pitched(channel_info['instrument_id'])
        vchannel = np.asarray(vchannel)
        bar, beat, fraction, row = np.nonzero(vchannel[..., 1])
        cells = vchannel[bar, beat, fraction, row]
        tpb = self.info['ticks_per_beat']
        notes = NoteTable(len(bar), bar=bar, beat=beat, fraction=fraction, velocity=cells[:, 1],
                          qduration=(cells[:, 0] * tpb).astype(np.int64))
        if pitched:
            flat, natural, sharp = cells[:, 2] != 0, cells[:, 3] != 0, cells[:, 4] != 0
            notes.accidental = np.where(flat, FLAT, np.where(natural, NATURAL, np.where(sharp, SHARP, NATURAL)))
            notes.scale_degree = row % 7 + 1
            notes.scale_octave = row // 7
        else:
            notes.accidental = np.full(len(bar), NO_ACCIDENTAL, np.int64)
            notes.note_id = row + self.min_percussion
        qchannel = dict(channel_info)
        qchannel['notes'] = notes
        return qchannel

    def qchannel2channel(self, channel_info, qchannel):
        """Notes -> note_on/note_off stream ordered by tick (stable over on0, off0, on1, off1, ...)."""
        notes = qchannel['notes']
        tpb, tpbar = self.info['ticks_per_beat'], self.info['ticks_per_bar']
        if is_pitched(channel_info['instrument_id']):
            absolute = np.asarray(self.mode.absolute_intervals, np.int64)
            spelled = np.array([-1, 0, 1, 0], np.int64)[notes.accidental]            # flat, none, sharp, (None)
            note_id = 12 * (notes.scale_octave + 1) + absolute[notes.scale_degree - 1] + key2interval[self.key] + spelled
        else:
            note_id = notes.note_id
        numer = np.array([f.numerator for f in self.beat_fractions], np.int64)[notes.fraction]
        denom = np.array([f.denominator for f in self.beat_fractions], np.int64)[notes.fraction]
        time = notes.bar * tpbar + notes.beat * tpb + (numer * tpb) // denom          # int(Fraction * int)
        n = len(notes)
        on = np.zeros(2 * n, bool)
        on[0::2] = True
        velocity = np.zeros(2 * n, dtype=notes.velocity.dtype)
        velocity[0::2] = notes.velocity
        messages = NoteMessages(on, np.repeat(note_id, 2), np.stack([time, time + notes.qduration], 1).reshape(-1), velocity)
        channel = dict(channel_info)
        channel['messages'] = messages.sorted_by_time()
        return channel

    def nchannel2vchannel(self, nchannel):
        kchannel = self.nchannel2kchannel(nchannel)
        qchannel = self.kchannel2qchannel(kchannel, in_place=True)
        return self.qchannel2vchannel(qchannel)

    def vchannel2channel(self, channel_info, vchannel):
        return self.qchannel2channel(channel_info, self.vchannel2qchannel(channel_info, vchannel))

    @property
    def mode(self):
        return self.info['scale']['mode']

    @property
    def key(self):
        return self.info['scale']['key']

    @property
    def n_bars(self):
        return math.ceil(self.info['n_bars'])

    def n_features(self, pitched):
        return self.n_note_features if pitched else self.n_unpitched_features

    def get_empty_beat(self, pitched):
        return np.zeros([len(self.beat_fractions), self.n_notes if pitched else self.n_unpitched, self.n_features(pitched)])

    def get_empty_bar(self, pitched):
        return [self.get_empty_beat(pitched) for _ in range(self.info['n_beats'])]
