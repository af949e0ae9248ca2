#!/usr/bin/env python3
"""Run the audio-extension kernels a few times (for rocprofv3 --kernel-trace --stats): STFT, Gram, style iteration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'music-style-transfer_amd')):
    sys.path.insert(0, p)
import torch
from style.audio import AudioPlan

dev = torch.device('cuda:0')
n = 30 * 44100
plan = AudioPlan(n, 1024, 256, device=dev)
g = torch.Generator().manual_seed(3)
audio = (torch.rand(n, generator=g) * 2 - 1).to(dev)
_, mag = plan.stft(audio, want_spec=False)
gs = plan.gram(mag).clone() * 1.1
x = mag.clone()
opt = plan.optimizer_state()
for _ in range(20):
    plan.stft(audio)
    plan.gram(mag)
    plan.style_iteration(x, gs, opt)
torch.cuda.synchronize()
print('loss', float(opt['loss'].cpu()[0]), 'splits', plan.splits, 'tiles', plan.tiles)
