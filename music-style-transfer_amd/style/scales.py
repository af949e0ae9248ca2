"""Diatonic mode algebra and Krumhansl-style key finding (host side, per song; style/scales.py:27-221).

Restated, not imported: `Mode` keeps the reference's attribute names because `ChannelConverter`
and `transfer_style` use `mode.get_degree`, `mode.absolute_intervals`, `major_mode` / `minor_mode`.
The reference's `ndcg` column (from the absent `py_utils`) never enters the loss (`scales.py:188`)
and is not computed.
"""
import numpy as np

key_names = ['C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B']
interval2key = dict(enumerate(key_names))
key2interval = {k: i for i, k in interval2key.items()}
_MODE_NAMES = ['Ionian', 'Dorian', 'Phrygian', 'Lydian', 'Mixolydian', 'Aeolian', 'Locrian']


def normalize_dist(dist):
    """style/utils/math.py:4-11."""
    dist = np.array(dist, dtype=float)
    total = dist.sum()
    if total > 0:
        return dist / total
    return np.full_like(dist, 1.0 / len(dist))


def cross_entropy(dist, target, epsilon=1e-12):
    """style/utils/metrics.py:4-8."""
    dist = np.clip(dist, epsilon, 1.)
    return -np.sum(target * np.log(dist)) / dist.shape[0]


class Mode:
    names = _MODE_NAMES

    def __init__(self, intervals, shift=0):
        assert len(intervals) == 7
        self.intervals, self.shift = list(intervals), shift
        self.tonic_intervals = list(np.concatenate([[0], np.cumsum(intervals)]).astype(int))
        self.absolute_intervals = [int(v) for v in self.tonic_intervals[:7]]
        # scale tones get integer degrees 1..7, chromatic tones the half step above the previous tone
        self.interval2degree, degree = {}, 1
        for semitone in range(12):
            if semitone in self.absolute_intervals:
                degree = self.absolute_intervals.index(semitone) + 1
                self.interval2degree[semitone] = degree
            else:
                self.interval2degree[semitone] = degree + .5

    @property
    def name(self):
        return _MODE_NAMES[self.shift % 7]

    def __len__(self):
        return 7

    def get_degree(self, interval):
        return self.interval2degree[interval % 12]

    def __repr__(self):
        return f'{self.name} mode'


def create_mode(mode, shift):
    iv = mode.intervals
    return Mode(iv[shift:] + iv[:shift], shift)


def get_relative_degree(interval, source_scale, target_scale):
    """Degree of `interval` (relative to the source tonic) inside the target mode (style/scales.py:117-121)."""
    rel = (source_scale.shift - target_scale.shift) % 7
    return target_scale.get_degree(interval + target_scale.tonic_intervals[rel])


major_mode = Mode([2, 2, 1, 2, 2, 2, 1])
minor_mode = create_mode(major_mode, shift=-2)

# Krumhansl-Kessler key profiles (style/scales.py:128-132)
major_dist = normalize_dist([6.35, 2.23, 3.48, 2.33, 4.38, 4.09, 2.52, 5.19, 2.39, 3.66, 2.29, 2.88])
minor_dist = normalize_dist([6.33, 2.68, 3.52, 5.38, 2.60, 3.53, 2.54, 4.75, 3.98, 2.69, 3.34, 3.17])
_TYPICAL = {'major': [0, 2, 4, 5, 6, 7, 9, 10, 11], 'minor': [0, 1, 2, 3, 5, 7, 8, 9, 10, 11]}


def get_scales(key2time=None, keys_dist=None):
    """All 24 (tonic, mode) candidates with loss = CE(profile) * (1.5 - coverage) * (2 - loose coverage)
    (style/scales.py:160-211); majors for C..B first, then minors, which is also the tie-break order."""
    if keys_dist is None:
        keys_dist = normalize_dist([key2time.get(k, 0) for k in key_names])
    keys_dist = np.asarray(keys_dist, dtype=float)
    out = []
    for mode_name, profile, mode in (('major', major_dist, major_mode), ('minor', minor_dist, minor_mode)):
        rotated = keys_dist
        for key in key_names:
            cov = rotated[mode.absolute_intervals].sum()
            loose = rotated[_TYPICAL[mode_name]].sum()
            ce = cross_entropy(rotated, profile)
            out.append(dict(key=key, mode=mode_name, coverage=cov, loose_coverage=loose, cross_entropy=ce,
                            loss=ce * (1.5 - cov) * (2 - loose)))
            rotated = np.concatenate([rotated[1:], rotated[:1]])
    return out


def get_scale(*args, **kwargs):
    best = min(get_scales(*args, **kwargs), key=lambda s: s['loss'])
    best['mode'] = major_mode if best['mode'] == 'major' else minor_mode
    return best
