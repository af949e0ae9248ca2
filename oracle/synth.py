"""Synthetic clip generator for the tests (TEST INFRASTRUCTURE): re-exported from tools/synth.py, where bench.py takes it from."""
from tools.synth import *  # noqa: F401,F403
from tools.synth import synth_clip, INSTRUMENT_SIZE, N_INSTRUMENTS  # noqa: F401
