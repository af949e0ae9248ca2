"""Standard MIDI File reader / writer — the subset of `mido` the reference uses
(style/midi.py:106,120-168; style/midi_conversion.py:37-66,177,333,343,394; style/style_transfer.py:103),
written from the SMF specification so the host path needs no third-party package.

Messages are plain objects with the attribute names mido gives them (`type`, `time`, `channel`,
`note`, `velocity`, `program`, `control`, `value`, `pitch`, `tempo`, `numerator`, `denominator`,
`key`), `time` being the delta in ticks inside a track.  The writer reproduces mido's byte stream
(running status for channel messages, reset by meta events), which the reference's example
files pin: reading one and writing it back is bit-identical (tests/test_midi_host.py).
"""
import math

CHANNEL_TYPES = {0x80: 'note_off', 0x90: 'note_on', 0xA0: 'polytouch', 0xB0: 'control_change',
                 0xC0: 'program_change', 0xD0: 'aftertouch', 0xE0: 'pitchwheel'}
_STATUS = {v: k for k, v in CHANNEL_TYPES.items()}
META_TYPES = {0x00: 'sequence_number', 0x01: 'text', 0x02: 'copyright', 0x03: 'track_name', 0x04: 'instrument_name',
              0x05: 'lyrics', 0x06: 'marker', 0x07: 'cue_marker', 0x20: 'channel_prefix', 0x21: 'midi_port',
              0x2F: 'end_of_track', 0x51: 'set_tempo', 0x54: 'smpte_offset', 0x58: 'time_signature',
              0x59: 'key_signature', 0x7F: 'sequencer_specific'}
_KEYS_MAJOR = ['Cb', 'Gb', 'Db', 'Ab', 'Eb', 'Bb', 'F', 'C', 'G', 'D', 'A', 'E', 'B', 'F#', 'C#']
_KEYS_MINOR = ['Abm', 'Ebm', 'Bbm', 'Fm', 'Cm', 'Gm', 'Dm', 'Am', 'Em', 'Bm', 'F#m', 'C#m', 'G#m', 'D#m', 'A#m']


class MidiError(ValueError):
    pass


class Message:
    """A channel message or a meta message (is_meta)."""
    is_meta = False

    def __init__(self, type, time=0, **fields):
        self.type = type
        self.time = time
        self.__dict__.update(fields)

    def copy(self):
        m = Message.__new__(type(self))
        m.__dict__.update(self.__dict__)
        return m

    __copy__ = copy

    def __repr__(self):
        f = ' '.join(f'{k}={v}' for k, v in self.__dict__.items() if k != 'type')
        return f'<{self.type} {f}>'


class MetaMessage(Message):
    is_meta = True


def tempo2bpm(tempo):
    return 60e6 / tempo


def bpm2tempo(bpm):
    return int(round(60e6 / bpm))


def tick2second(tick, ticks_per_beat, tempo):
    return tick * (tempo * 1e-6 / ticks_per_beat)


def second2tick(second, ticks_per_beat, tempo):
    return second / (tempo * 1e-6 / ticks_per_beat)


# ------------------------------------------------------------------------------------------- reading
def _vlq(data, i):
    value = 0
    for _ in range(4):
        if i >= len(data):
            raise EOFError('truncated variable-length quantity')
        c = data[i]
        i += 1
        value = (value << 7) | (c & 0x7F)
        if c < 0x80:
            return value, i
    raise MidiError('variable-length quantity longer than 4 bytes')


def _read_track(data):
    msgs, i, running = [], 0, None
    while i < len(data):
        delta, i = _vlq(data, i)
        if i >= len(data):
            raise EOFError('truncated track')
        status = data[i]
        if status == 0xFF:
            kind = data[i + 1]
            length, j = _vlq(data, i + 2)
            body = data[j:j + length]
            i = j + length
            running = None
            name = META_TYPES.get(kind, 'unknown_meta')
            m = MetaMessage(name, delta)
            if name == 'set_tempo' and length == 3:
                m.tempo = int.from_bytes(body, 'big')
            elif name == 'time_signature' and length >= 2:
                m.numerator, m.denominator = body[0], 2 ** body[1]
                m.clocks_per_click = body[2] if length > 2 else 24
                m.notated_32nd_notes_per_beat = body[3] if length > 3 else 8
            elif name == 'key_signature' and length == 2:
                sf = body[0] - 256 if body[0] > 127 else body[0]
                if not -7 <= sf <= 7 or body[1] not in (0, 1):
                    raise MidiError('bad key signature')          # mido raises KeySignatureError here
                m.key = (_KEYS_MINOR if body[1] else _KEYS_MAJOR)[sf + 7]
            elif name == 'channel_prefix' and length == 1:
                m.channel = body[0]                                   # like mido: lands in that channel's stream
                m.data = bytes(body)
            else:
                m.data = bytes(body)
            msgs.append(m)
            continue
        if status in (0xF0, 0xF7):
            length, j = _vlq(data, i + 1)
            msgs.append(Message('sysex', delta, data=bytes(data[j:j + length])))
            i = j + length
            running = None
            continue
        if status >= 0x80:
            running = status
            i += 1
        elif running is None:
            raise MidiError('data byte without running status')
        else:
            status = running
        kind = status & 0xF0
        if kind not in CHANNEL_TYPES:
            raise MidiError(f'unsupported status byte {status:#x}')
        n = 1 if kind in (0xC0, 0xD0) else 2
        if i + n > len(data):
            raise EOFError('truncated channel message')
        a = data[i]
        b = data[i + 1] if n == 2 else 0
        i += n
        name = CHANNEL_TYPES[kind]
        m = Message(name, delta, channel=status & 0x0F)
        if name in ('note_on', 'note_off'):
            m.note, m.velocity = a, b
        elif name == 'polytouch':
            m.note, m.value = a, b
        elif name == 'control_change':
            m.control, m.value = a, b
        elif name == 'program_change':
            m.program = a
        elif name == 'aftertouch':
            m.value = a
        else:
            m.pitch = (a | (b << 7)) - 8192
        msgs.append(m)
    return msgs


class MidiFile:
    def __init__(self, filename=None, type=1, ticks_per_beat=480):
        self.filename, self.type, self.ticks_per_beat, self.tracks = filename, type, ticks_per_beat, []
        if filename is not None:
            with open(filename, 'rb') as f:
                self._parse(f.read())

    def _parse(self, data):
        if data[:4] != b'MThd':
            raise OSError('MThd not found. Probably not a MIDI file')
        hlen = int.from_bytes(data[4:8], 'big')
        self.type = int.from_bytes(data[8:10], 'big')
        ntracks = int.from_bytes(data[10:12], 'big')
        division = int.from_bytes(data[12:14], 'big')
        if division & 0x8000:
            raise MidiError('SMPTE time division is not supported')
        self.ticks_per_beat = division
        i = 8 + hlen
        for _ in range(ntracks):
            if data[i:i + 4] != b'MTrk':
                raise OSError('MTrk not found')
            tlen = int.from_bytes(data[i + 4:i + 8], 'big')
            if i + 8 + tlen > len(data):
                raise EOFError('truncated track chunk')
            self.tracks.append(_read_track(data[i + 8:i + 8 + tlen]))
            i += 8 + tlen

    # ---------------------------------------------------------------------------------------- writing
    def save(self, filename):
        with open(filename, 'wb') as f:
            f.write(self.to_bytes())

    def to_bytes(self):
        out = bytearray(b'MThd' + (6).to_bytes(4, 'big') + self.type.to_bytes(2, 'big') +
                        len(self.tracks).to_bytes(2, 'big') + self.ticks_per_beat.to_bytes(2, 'big'))
        for track in self.tracks:
            body = _write_track(track)
            out += b'MTrk' + len(body).to_bytes(4, 'big') + body
        return bytes(out)


def _enc_vlq(v):
    v = int(v)
    if v < 0:
        raise MidiError('negative delta time')
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    return bytes(reversed(out))


def _write_track(track):
    msgs = list(track)
    if not msgs or msgs[-1].type != 'end_of_track':
        msgs.append(MetaMessage('end_of_track', 0))
    out, running = bytearray(), None
    for m in msgs:
        out += _enc_vlq(m.time)
        if m.is_meta:
            if m.type == 'set_tempo':
                kind, body = 0x51, int(m.tempo).to_bytes(3, 'big')
            elif m.type == 'time_signature':
                kind = 0x58
                body = bytes([m.numerator, int(math.log2(m.denominator)), getattr(m, 'clocks_per_click', 24),
                              getattr(m, 'notated_32nd_notes_per_beat', 8)])
            elif m.type == 'end_of_track':
                kind, body = 0x2F, b''
            elif m.type == 'key_signature':
                minor = m.key.endswith('m')
                sf = (_KEYS_MINOR if minor else _KEYS_MAJOR).index(m.key) - 7
                kind, body = 0x59, bytes([sf & 0xFF, int(minor)])
            else:
                kind = {v: k for k, v in META_TYPES.items()}[m.type]
                body = getattr(m, 'data', b'')
            out += bytes([0xFF, kind]) + _enc_vlq(len(body)) + body
            running = None
            continue
        if m.type == 'sysex':
            out += b'\xF0' + _enc_vlq(len(m.data)) + m.data
            running = None
            continue
        status = _STATUS[m.type] | m.channel
        if m.type in ('note_on', 'note_off'):
            data = bytes([m.note, m.velocity])
        elif m.type == 'polytouch':
            data = bytes([m.note, m.value])
        elif m.type == 'control_change':
            data = bytes([m.control, m.value])
        elif m.type == 'program_change':
            data = bytes([m.program])
        elif m.type == 'aftertouch':
            data = bytes([m.value])
        else:
            v = m.pitch + 8192
            data = bytes([v & 0x7F, v >> 7])
        if status != running:
            out.append(status)
            running = status
        out += data
    return bytes(out)
