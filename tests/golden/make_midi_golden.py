"""Generate golden fixtures for the host (MIDI) path by RUNNING the importable part of the reference
(build container only):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_midi_golden.py

`style.scales`, `style.utils.math`, `style.utils.metrics` import without `mido`; `style.midi*`,
`style.data`, `style.style_transfer` do not (mido / flatten_dict / py_utils are absent), so for those
the fixtures are the reference's own example outputs: a selection of examples/**/*.mid is copied
byte for byte into tests/golden/midi/ (data files written by the reference's create_midi).
Outputs: tests/golden/host_scales.json (inputs + expected outputs only).
"""
import json
import os
import shutil

import numpy as np

import style.scales as ref_scales                       # the REFERENCE
from style.utils.math import round_number, normalize_dist
from style.utils.metrics import cross_entropy

assert os.path.realpath(ref_scales.__file__).startswith('/root/reference/'), ref_scales.__file__
HERE = os.path.dirname(os.path.abspath(__file__))

out = {}
modes = {'major': ref_scales.major_mode, 'minor': ref_scales.minor_mode}
out['modes'] = {
    name: dict(intervals=list(m.intervals), shift=m.shift, name=m.name, tonic_intervals=[int(v) for v in m.tonic_intervals],
               absolute_intervals=[int(v) for v in m.absolute_intervals],
               degrees=[m.get_degree(i) for i in range(-12, 24)],
               relative_to_major=[ref_scales.get_relative_degree(i, m, ref_scales.major_mode) for i in range(-12, 24)])
    for name, m in modes.items()}
out['all_mode_names'] = [m.name for m in ref_scales.all_modes]
out['major_dist'] = ref_scales.major_dist.tolist()
out['minor_dist'] = ref_scales.minor_dist.tolist()

rng = np.random.default_rng(7)
cases = []
for _ in range(400):
    number = int(rng.integers(0, 200000))
    tpb = int(rng.choice([96, 100, 120, 192, 220, 384, 480, 960, 1000]))
    divisor = int(rng.choice([8, 3]))
    precision = tpb / divisor
    value, err = round_number(number, precision)
    cases.append([number, tpb, divisor, float(value), float(err)])
out['round_number'] = cases

dists = []
for _ in range(20):
    d = rng.random(12) * (rng.random(12) > .3)
    nd = normalize_dist(d)
    dists.append(dict(raw=d.tolist(), normalized=nd.tolist(),
                      ce_major=float(cross_entropy(nd, ref_scales.major_dist)),
                      ce_minor=float(cross_entropy(nd, ref_scales.minor_dist))))
dists.append(dict(raw=[0.] * 12, normalized=normalize_dist(np.zeros(12)).tolist(), ce_major=None, ce_minor=None))
out['dists'] = dists

with open(os.path.join(HERE, 'host_scales.json'), 'w') as f:
    json.dump(out, f)

EXAMPLES = '/root/reference/examples'
PICK = [
    'style transfer - midi/Orient Express/original/Minuetto in sol magg. BWV App. 114.mid',
    'style transfer - midi/Orient Express/original/Nocturne No. 1 in E minor, Op. 72_ Andante.mid',
    'drums - midi/Welcome to the Jungle.2 (300 it).mid',
    'style transfer - midi/My Way/original/Angie.4.mid',
    'style transfer - midi/My Way/original/Dancing in the Moonlight.mid',
    'style transfer - midi/It Must Have Been Love/original/Vogue.3.mid',
    'style transfer - midi/Sweet Dreams/original/Sweet Dreams.mid',
    'style transfer - midi/Sweet Dreams/Sweet Dreams (Tico Tico No Fuba style).mid',
    'style transfer - midi/It Must Have Been Love/original/Kashmir.2.mid',
    'style transfer - midi/My Way/original/Heroic Polonaise No. 6 in A flat, Opus 53.mid',
]
for rel in PICK:
    shutil.copyfile(os.path.join(EXAMPLES, rel), os.path.join(HERE, 'midi', os.path.basename(rel)))
print('wrote host_scales.json and', len(PICK), 'midi fixtures')
