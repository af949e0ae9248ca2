#!/usr/bin/env python3
"""Developer tool: per-launch-step timing table of one training iteration on the bench clip
(HIP events through mst_plan_time_steps). Usage on the GPU box: python tools/step_profile.py [C R T]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'music-style-transfer_amd')]
import numpy as np
import torch

from bench import CLIP, WIDTHS, KIND_NAMES, init_params
from tools.synth import synth_clip
from style import _native as nat

shape = dict(CLIP)
if len(sys.argv) >= 4:
    shape = dict(C=int(sys.argv[1]), R=int(sys.argv[2]), T=int(sys.argv[3]))
K = int(sys.argv[4]) if len(sys.argv) >= 5 else 1          # clips per launch (batched plan)
dev = torch.device('cuda:0')
native = nat.get()
native.lib.mst_plan_step_info.restype = C.c_int32
dims = nat.Dims(**shape, **WIDTHS, instr=51, n_instruments=41, has_unpitched=1, clips=K)
flat, table = init_params(native, dims)
clips = [synth_clip(k, shape['C'], shape['R'], shape['T'], True) for k in range(K)]
plan = native.plan(dims, dev)
for k, clip in enumerate(clips):
    plan.set_inputs(mode=clip['mode'], bpm=clip['bpm'], instr=clip['instruments_features'], used=clip['used_instruments'],
                    bpm_target=120., clip=k)
params = flat.to(dev); g = torch.zeros_like(params)
xp = torch.cat([c['pitched'] for c in clips]).contiguous().to(dev)
xu = torch.cat([c['unpitched'] for c in clips]).contiguous().to(dev)
plan.train_iteration(params, g, xp, xu)
torch.cuda.synchronize()
tot = 0
for bwd in (False, True):
    steps = plan.time_steps(7, bwd, params, g, xp, xu, reps=20)
    info = np.zeros(8 * len(steps), np.int32)
    native.lib.mst_plan_step_info(C.c_void_p(plan.handle), 7, int(bwd), C.c_void_p(info.ctypes.data))
    info = info.reshape(-1, 8)
    print('==== backward' if bwd else '==== forward')
    for i, ((kind, ms, fl, by), inf) in enumerate(zip(steps, info)):
        tot += ms
        print(f'{i:3d} {KIND_NAMES[kind]:22s} {ms*1e3:8.1f} us  {fl/1e6:9.2f} MFLOP {by/1e6:8.2f} MB  {fl/ms/1e9:7.2f} TF {by/ms/1e6:7.1f} GB/s  info={inf.tolist()}')
print('sum of steps: %.3f ms' % tot)
