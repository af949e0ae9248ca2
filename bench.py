#!/usr/bin/env python3
"""bench.py — style-transfer optimisation iterations/sec (BASELINE.json metric).

A "step" is one body of the reference's training loop (train-model.py:97-154): forward ->
get_total_loss -> backward (gradients accumulate), with the Adam + StepLR step every
iter_size = 2 iterations, on one synthetic "30 s" clip (C=4 pitched channels, R=16 bars, T=4
beats, percussion channel present; BASELINE.json configs[1]).  Clips and
parameters are resident in HBM before the timed region.  With N > 1 GPUs every rank owns its
own clip (weak scaling) and the flat fp32 gradient buffer is all-reduced (SUM, RCCL over xGMI)
before every optimizer step — the data-parallel form of the reference's gradient accumulation.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'music-style-transfer_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

WIDTHS = dict(beat=64, bar=128, nrf=8, style=256, melody=8, rhythm=32)
CLIP = dict(C=4, R=16, T=4)
ITER_SIZE = 2
PEAK_F32_TFLOPS = 157.3       # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0
KIND_NAMES = {0: 'gemm_kernel', 1: 'gather_kernel', 2: 'segred_kernel', 3: 'lstm_fwd_kernel', 4: 'lstm_bwd_kernel',
              5: 'combine_fwd', 6: 'combine_bwd', 7: 'me_notes_fwd_kernel', 8: 'me_notes_bwd_kernel',
              9: 'psa_notes_fwd_kernel', 10: 'psa_notes_bwd_kernel', 11: 'lstm_transpose_kernel', 12: 'rowlin_fwd_kernel',
              13: 'rowlin_bwd_kernel', 14: 'me_reduce_kernel(fwd)', 15: 'me_reduce_kernel(bwd)', 26: 'conv_prep_kernel',
              27: 'conv_fwd_kernel', 28: 'conv_dw_kernel', 29: 'lin_fwd_kernel', 30: 'lin_dx_kernel', 31: 'lin_dw_kernel'}


def algorithmic_flops_per_iter(C, R, T, U=1):
    """FLOP_fwd over the reference's (unfused) layer shapes; fwd+bwd = 3x."""
    P, Q = C * R * T, R * T
    fwd = 2470880 * P + 209288 * Q + 651328 * R + U * (934400 * Q + 141312 * R)
    return 3 * fwd


N_PARAMS = 980325


def algorithmic_bytes_per_iter(C, R, T, iter_size, U=1):
    """SURVEY.md §8(d), fused ideal, fp32: the note tensors read twice (encode + loss target) and the prediction written once,
    melody / rhythm written + read, plus the parameter traffic of an optimizer step (read fwd + read bwd + gradient write = 3x,
    Adam reads p, g, m, v and writes p, m, v = 7x: 10 x 4 B x 980 325 = 39.2 MB) AMORTISED over the iter_size clip-iterations
    that share the step (train-model.py:95,151-154): 11.8 + 39.2 / iter_size MB for the bench clip."""
    P, Q = C * R * T, R * T
    per_clip = 3 * 11200 * P + U * 3 * 3760 * Q + (17920 + 1280) * 2 * Q
    return per_clip + 10 * 4 * N_PARAMS / iter_size


def pmc_traffic_per_iteration(clips=1):
    """HBM bytes per clip-iteration over ALL kernels of the timed loop, from the committed rocprofv3 --pmc passes, or None."""
    path = _latest_profile('pmc_hbm_traffic.json' if clips == 1 else f'pmc_hbm_traffic_{clips}_clips.json')
    try:
        return float(json.load(open(path))['per_iteration']['hbm_bytes'])
    except (OSError, ValueError, KeyError, TypeError):
        return None


def init_params(native, dims, seed=108):
    """Random-init weights of the reference architecture (uniform +-1/sqrt(fan_in), like nn.Linear)."""
    table = native.param_table(dims)
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(native.param_floats(dims))
    for name, off, shape in table:
        n = int(np.prod(shape))
        fan = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        flat[off:off + n] = (torch.rand(n, generator=g) * 2 - 1) / fan ** 0.5
    return flat, table


def cpu_baseline(flat, table, clip, seconds):
    """The oracle (torch-CPU port of the reference's path, pinned to its fixtures) on this host."""
    from oracle import style_oracle as so
    # a 1-GPU box owns 16 host cores; torch's default (all 128 visible cores) oversubscribes badly
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    named = {n: flat[o:o + int(np.prod(s))].view(*s).clone().requires_grad_(True) for n, o, s in table}
    opt = so.Adam(named.values())
    so.iteration(named, clip, fast=True)          # warm
    opt.step()
    t0 = time.perf_counter()
    it = 0
    while True:
        so.iteration(named, clip, fast=True)
        it += 1
        if it % ITER_SIZE == 0:
            opt.step()
        if time.perf_counter() - t0 >= seconds and it % ITER_SIZE == 0:
            break
    dt = time.perf_counter() - t0
    return dict(value=it / dt, unit='iters/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{it} iterations of the same clip/config through oracle/style_oracle.py (torch CPU fp32, '
                       f'mkldnn LSTM), {dt:.1f} s')


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round NN that has one (profiles are committed per round)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_{suffix}')))
    return hits[-1] if hits else None


def pmc_traffic(kernel, clips=1):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/), or None.
    bench.py cannot run the profiler on itself; the counters are collected by the command recorded in the file
    (one file per workload: one clip per launch, 64 clips per launch)."""
    path = _latest_profile('pmc_hbm_traffic.json' if clips == 1 else f'pmc_hbm_traffic_{clips}_clips.json')
    try:
        d = json.load(open(path))
        return d[kernel]['hbm_bytes_per_launch'] if kernel in d else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def rocprof_avg_us(kernel, clips=1):
    """Average duration (us) of `kernel` in the committed `rocprofv3 --kernel-trace --stats` summary of this bench
    command (profiles/rNN_rocprofv3_kernel_stats[_K_clips].csv), or None."""
    import csv
    path = _latest_profile('rocprofv3_kernel_stats.csv' if clips == 1 else f'rocprofv3_kernel_stats_{clips}_clips.csv')
    try:
        for row in csv.DictReader(open(path)):
            name = row.get('Name', '')
            if name == kernel or name.startswith(kernel + '(') or name.startswith('void ' + kernel):
                return float(row['AverageNs']) / 1e3
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None


def roofline_leg(plan, nat, params, gparams, xp, xu, K, dt_per_clip_iter, breakdown=False, iter_size=None):
    """Every launch step of one pass timed with HIP events on the current stream (mst_plan_time_steps: 20 back-to-back
    launches of the step on an otherwise idle GPU), aggregated per kernel; the roofline object is for the kernel with
    the largest share.  achieved = sum of its descriptors' algorithmic FLOPs / sum of its launch durations."""
    steps = plan.time_steps(nat.STAGE_ALL, False, params, gparams, xp, xu, reps=20) + \
        plan.time_steps(nat.STAGE_ALL, True, params, gparams, xp, xu, reps=20)
    agg = {}
    for kind, ms, fl, by in steps:
        a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl; a[3] += by
    total_ms = sum(a[1] for a in agg.values())
    names = dict(KIND_NAMES)
    if plan.gemm_tile == 64:
        names[0] = 'gemm_mfma_kernel'            # plans with >= 6 clips per launch use the 64x64-tile GEMM
    rows = [dict(kernel=names[kind], launches=cnt, total_us=round(ms * 1e3, 1), avg_us=round(ms * 1e3 / cnt, 2),
                 share=round(ms / total_ms, 3), gflop=round(fl / 1e9, 4), mbytes=round(by / 1e6, 2))
            for kind, (cnt, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    kind, (cnt, ms, fl, by) = max(agg.items(), key=lambda kv: kv[1][1])
    achieved = fl / (ms * 1e-3) / 1e12
    prof_us = rocprof_avg_us(names[kind], K)
    roof = dict(bound='mfma', kernel=names[kind], launches_per_iter=cnt, avg_launch_us=round(ms * 1e3 / cnt, 2),
                timer='HIP events (hipEventRecord on the launch stream) around 20 back-to-back launches of each step, GPU otherwise idle',
                flop_per_launch=fl / cnt, achieved=achieved, peak=PEAK_F32_TFLOPS, unit='TFLOP/s',
                frac=achieved / PEAK_F32_TFLOPS, traffic=pmc_traffic(names[kind], K), clips_per_launch=K,
                # the same kernel inside the committed rocprofv3 run of this command (there launches of the two concurrent
                # accumulation iterations share the chip, so its average is longer than the isolated one)
                rocprof_avg_launch_us=prof_us,
                frac_from_rocprof=(fl / cnt / (prof_us * 1e-6) / 1e12 / PEAK_F32_TFLOPS) if prof_us else None,
                whole_iteration=dict(algorithmic_gflop=algorithmic_flops_per_iter(**CLIP) / 1e9,
                                     achieved_tflops=algorithmic_flops_per_iter(**CLIP) / dt_per_clip_iter / 1e12))
    # whole clip-iteration against the HBM roof: algorithmic bytes with the optimizer step's parameter traffic amortised over
    # the iter_size clip-iterations that share it, beside the PMC-measured bytes of every kernel of the loop
    ab = algorithmic_bytes_per_iter(**CLIP, iter_size=iter_size or max(K, ITER_SIZE))
    tb = pmc_traffic_per_iteration(K)
    roof.update(algorithmic_bytes=ab, traffic_per_iteration=tb, traffic_ratio=(tb / ab) if tb else None,
                algorithmic_bytes_note='per clip-iteration, whole loop body; 11.8 MB of note / melody / rhythm traffic + 39.2 MB of '
                                       'parameter + Adam traffic per optimizer step / iter_size; traffic_per_iteration = FETCH_SIZE + '
                                       'WRITE_SIZE of all kernels (profiles/, raw counters: 16-byte loads are under-reported)')
    if breakdown:
        for r in rows:
            print(json.dumps(r), file=sys.stderr)
    return roof, rows


def batched_leg(native, nat, dev, flat, K, passes, warm):
    """BASELINE.json configs[2] beside the headline: K different clips in ONE batched plan (every launch carries all K),
    Adam after every pass; one hipGraph replay = one pass + optimizer step.  Returns clip-iterations/s and the roofline
    of the pass's dominant kernel."""
    from tools.synth import synth_clip
    dims = nat.Dims(**CLIP, **WIDTHS, instr=51, n_instruments=41, has_unpitched=1, clips=K)
    plan = nat.Plan(native, dims, dev)
    clips = [synth_clip(k, CLIP['C'], CLIP['R'], CLIP['T'], True) for k in range(K)]
    for k, c in enumerate(clips):
        plan.set_inputs(mode=c['mode'], bpm=c['bpm'], instr=c['instruments_features'], used=c['used_instruments'],
                        bpm_target=float(c['bpm_int']), clip=k)
    params = flat.to(dev)
    g, m, v = torch.zeros_like(params), torch.zeros_like(params), torch.zeros_like(params)
    state = torch.zeros(4, device=dev)
    xp = torch.cat([c['pitched'] for c in clips]).contiguous().to(dev)
    xu = torch.cat([c['unpitched'] for c in clips]).contiguous().to(dev)
    losses = torch.zeros(K, nat.N_LOSSES, device=dev)
    P = nat.ptr
    stream = torch.cuda.Stream(dev)

    def one_pass():
        plan.train_iteration(params, g, xp, xu, losses)
        nat.check(native.lib.mst_adam_step(P(params), P(g), P(m), P(v), params.numel(), P(state), .01, .9, .999, 1e-8, 200, .9,
                                           1, nat.current_stream(dev)), 'mst_adam_step')

    with torch.cuda.stream(stream):
        one_pass()
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            one_pass()
        for _ in range(warm):
            graph.replay()
        times = []
        while len(times) < 3 or (sum(times) < 0.25 and len(times) < 50):      # >= 3 regions of `passes` passes, >= 0.25 s in all
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(passes):
                graph.replay()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        dt = float(np.median(times))
        roof, rows = roofline_leg(plan, nat, params, g, xp, xu, K, dt / (passes * K), iter_size=K)
    return dict(value=passes * K / dt, unit='clip-iterations/s', ms_per_pass=dt / passes * 1e3, passes=passes, warmup_passes=warm,
                spread=dict(repeats=len(times), min_ms_per_pass=min(times) / passes * 1e3, max_ms_per_pass=max(times) / passes * 1e3,
                            value='median over the repeats'),
                config=dict(workload=f'{K} different 30 s clips (C=4,R=16,T=4 +percussion) on one GPU in one batched plan '
                                     f'(BASELINE.json configs[2]): fwd+loss+bwd of all {K} clips per pass, gradients summed, '
                                     'Adam+StepLR after every pass', clips_per_launch=K, iter_size=K, hip_graph=True,
                            launches_per_pass=plan.launch_count(7, False) + plan.launch_count(7, True) + 3,
                            final_total_loss_clip0=float(losses.cpu()[0, 0])),
                roofline=roof, kernel_breakdown=rows)


def surface_leg(dev, clip, iters, warm):
    """The drop-in path a user of the reference gets by changing the import (train-model.py:97-154 -> style/train.py): the
    nn.Module surface — model(...) -> get_total_loss(...) -> losses['total'].backward() -> FusedAdam.step() every iter_size
    iterations — on the bench clip, seed-108 weights, eager launches through torch.autograd (no hipGraph, the 15 loss leaves
    stay on the device like in style/train.py)."""
    import style.model as sm
    from style.optim import FusedAdam
    from style.train import build_model
    model = build_model().to(dev)
    opt = FusedAdam(model, lr=.01, step_size=200, gamma=.9)
    c = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in clip.items()}

    def body(it):
        (ip, mp, bp), xp, xu = model(c['mode'], c['bpm'], c['pitched'], c['instruments_features'], c['unpitched'])
        losses = sm.get_total_loss(ip, c['used_instruments'], bp, c['bpm_int'], mp, c['mode'], xp, c['pitched'], xu, c['unpitched'],
                                   normalize=True)
        losses['total'].backward()
        if (it + 1) % ITER_SIZE == 0:
            opt.step()
        return losses.packed

    def fused_body(it):      # style/train.py's default: the loop body as one C-ABI call (StyleTransferModel.train_iteration)
        packed = model.train_iteration(c['mode'], c['bpm'], c['pitched'], c['instruments_features'], c['unpitched'],
                                       c['used_instruments'], c['bpm_int'])
        if (it + 1) % ITER_SIZE == 0:
            opt.step()
        return packed

    def timed(fn):
        for it in range(warm):
            fn(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(iters):
            packed = fn(it)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, packed

    dt_f, _ = timed(fused_body)
    dt, packed = timed(body)

    # songs differ in (C, R): the training loop sees a new shape almost every iteration.  Six synthetic clips of different
    # shapes in rotation through the fused body — one plan + one captured graph per shape, built during the warm-up
    # rotations and reused afterwards (plan cache: 16 shapes)
    from tools.synth import synth_clip
    shapes = [(4, 16), (3, 12), (4, 24), (2, 20), (5, 8), (4, 12)]
    songs = []
    for i, (C_, R_) in enumerate(shapes):
        sc = synth_clip(20 + i, C_, R_, CLIP['T'], True)
        songs.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()})

    def varying_body(it):
        sg = songs[it % len(songs)]
        packed = model.train_iteration(sg['mode'], sg['bpm'], sg['pitched'], sg['instruments_features'], sg['unpitched'],
                                       sg['used_instruments'], sg['bpm_int'])
        if (it + 1) % ITER_SIZE == 0:
            opt.step()
        return packed

    for it in range(3 * len(songs)):
        varying_body(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        varying_body(it)
    torch.cuda.synchronize()
    dt_v = time.perf_counter() - t0
    return dict(value=iters / dt, fused_value=iters / dt_f, fused_ms_per_step=dt_f / iters * 1e3,
                fused_varying_shapes_value=iters / dt_v, varying_shapes='(C,R) in ' + str(shapes) + ' in rotation, T=4, percussion',
                unit='iters/s', ms_per_step=dt / iters * 1e3, steps=iters, warmup=warm,
                config=dict(workload='the same clip through the reference\'s Python surface (style.model.StyleTransferModel.forward, '
                                     'get_total_loss, autograd backward, FusedAdam every 2nd iteration), eager, no hipGraph',
                            final_total_loss=float(packed.cpu()[0])))


def audio_leg(nat, dev, iters, warm, cpu_seconds):
    """AUDIO EXTENSION — not the headline, not reference parity (the reference has no audio path).  BASELINE.json's
    metric text taken literally: a 30 s @ 44.1 kHz clip, STFT(1024/256) -> 5168 x 513 magnitudes, one optimisation iteration =
    Gram of x, loss || G(x) - G_style ||^2, gradient (4/T) x (G - G_s), Adam on x; replayed from a hipGraph.  Reported under its
    own key with its own rooflines (STFT: HBM; Gram / gradient GEMMs: f32 MFMA)."""
    from style.audio import AudioPlan
    n = 30 * 44100
    plan = AudioPlan(n, 1024, 256, device=dev)
    g = torch.Generator().manual_seed(3)
    audio = (torch.rand(n, generator=g) * 2 - 1).to(dev)
    style_audio = (torch.rand(n, generator=g) * 2 - 1).to(dev)
    stream = torch.cuda.Stream(dev)

    def timed_us(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(reps):
            fn()
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    with torch.cuda.stream(stream):
        stft_us = timed_us(lambda: plan.stft(audio), 50)
        stft_mag_us = timed_us(lambda: plan.stft(audio, want_spec=False), 50)
        _, mag = plan.stft(audio, want_spec=False)
        _, smag = plan.stft(style_audio, want_spec=False)
        gram_us = timed_us(lambda: plan.gram(mag), 50)
        gs = plan.gram(smag).clone()
        x = mag.clone()
        opt = plan.optimizer_state()
        plan.style_iteration(x, gs, opt)
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            plan.style_iteration(x, gs, opt)
        for _ in range(warm):
            graph.replay()
        torch.cuda.synchronize()
        first_loss = float(opt['loss'].cpu()[0])
        regions = []
        for _ in range(3):                              # median of three regions: one 200-step region is ~25 ms
            t0 = time.perf_counter()
            for _ in range(iters):
                graph.replay()
            torch.cuda.synchronize()
            regions.append(time.perf_counter() - t0)
        dt = sorted(regions)[1]
        last_loss = float(opt['loss'].cpu()[0])
    T, F, ld = plan.frames, plan.bins, plan.ld
    stft_bytes = 4 * n + 8 * T * F
    gram_flops = 2.0 * T * F * (F + 1) / 2            # the symmetric half that is computed (lower-triangle tiles)
    it_flops = gram_flops + 2.0 * T * ld * ld
    out = dict(label='AUDIO EXTENSION - not reference parity (the reference has no audio path); oracle build-defined, parity unpinned',
               metric='style-transfer opt iters/sec (spectrogram Gram loss, 30 s @ 44.1 kHz, STFT 1024/256)', value=iters / dt, unit='iters/s',
               ms_per_step=dt / iters * 1e3, steps=iters, warmup=warm, hip_graph=True, dtype='f32', data='synthetic (uniform noise clips)',
               config=dict(workload='x (5168 x 513 magnitudes, resident in HBM) optimised towards the Gram of a second clip: Gram GEMM '
                                    '(k-split, lower-triangle 128x128 tiles) + finalise/loss + gradient GEMM + Adam per iteration',
                           frames=T, bins=F, ld=ld, gram_k_splits=plan.splits, loss_first=first_loss, loss_last=last_loss),
               stft=dict(us_complex_and_magnitude=stft_us, us_magnitude_only=stft_mag_us,
                         roofline=dict(bound='hbm', achieved=stft_bytes / (stft_us * 1e-6) / 1e9, peak=PEAK_HBM_GBS, unit='GB/s',
                                       frac=stft_bytes / (stft_us * 1e-6) / 1e9 / PEAK_HBM_GBS, algorithmic_bytes=stft_bytes,
                                       timer='HIP events around 50 back-to-back launches', traffic=None)),
               gram=dict(us=gram_us, roofline=dict(bound='mfma', achieved=gram_flops / (gram_us * 1e-6) / 1e12, peak=PEAK_F32_TFLOPS,
                                                   unit='TFLOP/s', frac=gram_flops / (gram_us * 1e-6) / 1e12 / PEAK_F32_TFLOPS,
                                                   flop_per_launch=gram_flops, timer='HIP events around 50 launches (GEMM + finalise)',
                                                   traffic=None)),
               iteration_tflops=it_flops / (dt / iters) / 1e12)
    if cpu_seconds > 0:
        from oracle import audio_oracle as ao
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
        feat = mag.cpu()[:, :F].contiguous()
        gsr = ao.gram(smag.cpu()[:, :F].contiguous())
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < cpu_seconds:
            ao.style_iterations(feat, gsr, 2, 1e-2)
            k += 2
        cdt = time.perf_counter() - t0
        out['cpu_baseline'] = dict(value=k / cdt, unit='iters/s', cores=torch.get_num_threads(), kind='port',
                                   sample=f'{k} iterations of the same optimisation through oracle/audio_oracle.py (torch CPU), {cdt:.1f} s')
    return out


LONG_CLIP = dict(C=8, R=151, T=4)       # BASELINE.json configs[4] mapped per SURVEY.md 8(d): 5 min at 120 bpm + 1 = 151 bars, 8 channels


def long_clip_untiled(native, nat, dev, flat, iters, warm):
    """The same long clip (C=8, R=151, T=4) as ONE untiled plan on one GPU, hipGraph-replayed: the denominator of the tiled
    leg's strong-scaling efficiency (what one GPU does without any exchange)."""
    from tools.synth import synth_clip
    dims = nat.Dims(**LONG_CLIP, **WIDTHS, instr=51, n_instruments=41, has_unpitched=1)
    plan = nat.Plan(native, dims, dev)
    c = synth_clip(9, LONG_CLIP['C'], LONG_CLIP['R'], LONG_CLIP['T'], True)
    plan.set_inputs(mode=c['mode'], bpm=c['bpm'], instr=c['instruments_features'], used=c['used_instruments'], bpm_target=float(c['bpm_int']))
    xp, xu = c['pitched'].contiguous().to(dev), c['unpitched'].contiguous().to(dev)
    params = flat.to(dev)
    g, m, v = torch.zeros_like(params), torch.zeros_like(params), torch.zeros_like(params)
    state = torch.zeros(4, device=dev)
    losses = torch.zeros(nat.N_LOSSES, device=dev)
    P = nat.ptr
    stream = torch.cuda.Stream(dev)

    def iteration():
        plan.train_iteration(params, g, xp, xu, losses)
        nat.check(native.lib.mst_adam_step(P(params), P(g), P(m), P(v), params.numel(), P(state), .01, .9, .999, 1e-8, 200, .9, 1,
                                           nat.current_stream(dev)), 'mst_adam_step')

    with torch.cuda.stream(stream):
        iteration()
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            iteration()
        for _ in range(warm):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            graph.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dict(value=iters / dt, unit='iters/s', ms_per_step=dt / iters * 1e3, steps=iters, hip_graph=True,
                final_total_loss=float(losses.cpu()[0]))


def tiled_leg(native, nat, dev, flat, dist, rank, world, iters, warm):
    """BASELINE.json configs[4]: ONE long clip (C=8, R=151, T=4) whose bars are tiled over the ranks: every
    rank computes the per-position work of its contiguous bar tile, the bar-level chains run replicated, and the iteration
    is a sequence of phases with small all-reduces (SUM) in between (mst_tiled_phase), then the flat-gradient all-reduce and
    Adam.  A step = one training iteration of the whole clip (strong scaling: the clip is fixed, the tiles shrink)."""
    from tools.synth import synth_clip
    C_, R_, T_ = LONG_CLIP['C'], LONG_CLIP['R'], LONG_CLIP['T']
    base, extra = divmod(R_, world)
    rows = base + (1 if rank < extra else 0)
    r0 = rank * base + min(rank, extra)
    dims = nat.Dims(**LONG_CLIP, **WIDTHS, instr=51, n_instruments=41, has_unpitched=1)
    plan = nat.Plan(native, dims, dev, tile_r0=r0, tile_rows=rows)
    c = synth_clip(9, C_, R_, T_, True)
    plan.set_inputs(mode=c['mode'], bpm=c['bpm'], instr=c['instruments_features'], used=c['used_instruments'], bpm_target=float(c['bpm_int']))
    xp = c['pitched'][:, :, r0:r0 + rows].contiguous().to(dev)
    xu = c['unpitched'][:, :, r0:r0 + rows].contiguous().to(dev)
    params = flat.to(dev)
    g, m, v = torch.zeros_like(params), torch.zeros_like(params), torch.zeros_like(params)
    state = torch.zeros(4, device=dev)
    losses = torch.zeros(nat.N_LOSSES, device=dev)
    P = nat.ptr
    n_x = [0]

    def all_reduce(t):
        n_x[0] += 1
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def iteration():
        plan.tiled_train_iteration(params, g, xp, xu, losses, is_root=rank == 0, all_reduce=all_reduce)
        if dist is not None:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        nat.check(native.lib.mst_adam_step(P(params), P(g), P(m), P(v), params.numel(), P(state), .01, .9, .999, 1e-8, 200, .9, 1,
                                           nat.current_stream(dev)), 'mst_adam_step')

    for _ in range(warm):
        iteration()
    n_x[0] = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.cpu()[0])
    return dict(value=iters / dt, unit='iters/s', ms_per_step=dt / iters * 1e3, steps=iters, warmup=warm, scaling='strong',
                config=dict(workload=f'one 5-min clip C={C_},R={R_},T={T_} (+percussion), bars tiled over {world} rank(s) '
                                     f'(BASELINE.json configs[4]): fwd+loss+bwd+Adam every step, {n_x[0] // max(iters, 1)} in-iteration '
                                     'all-reduces (SUM) of small workspace ranges + the flat-gradient all-reduce',
                            bars_per_rank=rows, hip_graph=False, collectives_per_iteration=n_x[0] // max(iters, 1) + (1 if dist is not None else 0),
                            final_total_loss=float(losses.cpu()[0])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown', action='store_true', help='print the per-kernel time table to stderr')
    ap.add_argument('--no-roofline', action='store_true',
                    help='skip the per-step timing leg (PMC runs: its 2 x 21 extra launches of every step would count as traffic)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL); gloo only to rehearse N>1 on one GPU')
    ap.add_argument('--share-device', action='store_true', help='rehearsal: every rank uses cuda:0')
    ap.add_argument('--clips-per-gpu', type=int, default=1,
                    help='B > 1: B different clips per GPU in one batched plan (BASELINE.json configs[2]/[3]); a step is then one '
                         'clip-iteration, the optimizer steps once per pass over the B clips (iter_size = B per GPU)')
    ap.add_argument('--batched-clips', type=int, default=64,
                    help='with the default one-clip workload on 1 GPU: also time this many clips in one batched plan (configs[2]) for '
                         'a few passes and report it as "batched" in the same JSON line; 0 = skip')
    ap.add_argument('--batched-passes', type=int, default=12)
    ap.add_argument('--surface-steps', type=int, default=200,
                    help='with the default workload on 1 GPU: also time this many loop bodies through the nn.Module surface '
                         '(the drop-in path behind train-model.py) and report them as "surface"; 0 = skip')
    ap.add_argument('--audio-steps', type=int, default=200,
                    help='with the default workload on 1 GPU: also time this many iterations of the AUDIO EXTENSION (no reference '
                         'counterpart; reported as "audio_extension", never part of the headline); 0 = skip')
    ap.add_argument('--tile-bars', action='store_true',
                    help='headline = BASELINE.json configs[4] instead: ONE long clip (C=8, R=151, T=4) with its bars tiled over the '
                         '--gpus ranks (strong scaling)')
    ap.add_argument('--graphs', choices=['per-lane', 'joint'], default='joint',
                    help="'streams' accumulation: one graph holding both lanes and the Adam step (default), or one hipGraph per lane "
                         'replayed on its own stream + eager Adam (experiment, slower)')
    ap.add_argument('--accum', choices=['streams', 'batched'], default='streams',
                    help='B = 1: run the iter_size = 2 accumulation iterations on two streams (default) or as one 2-clip batched pass')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    dev = torch.device('cuda', 0 if args.share_device else local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from tools.synth import synth_clip
    from style import _native as nat
    native = nat.get()                     # raises if libmst_amd.so is missing: no fallback
    B = max(1, args.clips_per_gpu)
    batched = B > 1 or args.accum == 'batched'
    K = B if B > 1 else (ITER_SIZE if batched else 1)          # clips carried by every launch of the plan
    iter_size = B if B > 1 else ITER_SIZE                        # clip-iterations between optimizer steps, per GPU
    dims1 = nat.Dims(**CLIP, **WIDTHS, instr=51, n_instruments=41, has_unpitched=1)
    dims = nat.Dims(**CLIP, **WIDTHS, instr=51, n_instruments=41, has_unpitched=1, clips=K)
    flat, table = init_params(native, dims1)
    if args.tile_bars:
        r = tiled_leg(native, nat, dev, flat, dist, rank, world, args.steps, args.warmup)
        # the untiled clip on ONE GPU (every rank measures its own card; rank 0 reports): strong-scaling denominator
        un = long_clip_untiled(native, nat, dev, flat, max(10, args.steps // 2), 3)
        r['config']['untiled_one_gpu'] = un
        r['config']['speedup_vs_untiled_one_gpu'] = r['value'] / un['value']
        if rank == 0:
            print(json.dumps(dict(metric='style-transfer opt iters/sec', value=r['value'], unit='iters/s', n_gpus=world, steps=args.steps,
                                  warmup=args.warmup, ms_per_step=r['ms_per_step'], higher_is_better=True, scaling='strong',
                                  vs_baseline=None, dtype='f32', data='synthetic', config=r['config'])))
        if dist is not None:
            dist.destroy_process_group()
        return
    # B = 1: the one clip of this GPU (both accumulation iterations run on it, like the reference looping over a
    # one-song dataset); B > 1: B different clips
    clips = [synth_clip(rank * B + (k if B > 1 else 0), CLIP['C'], CLIP['R'], CLIP['T'], True) for k in range(K)]
    clip = clips[0]
    plan = native.plan(dims, dev)

    def set_all(ws=None):
        for k, c in enumerate(clips):
            plan.set_inputs(mode=c['mode'], bpm=c['bpm'], instr=c['instruments_features'], used=c['used_instruments'],
                            bpm_target=float(c['bpm_int']), ws=ws, clip=k)

    set_all()
    params = flat.to(dev)
    gparams = torch.zeros_like(params)
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    state = torch.zeros(4, device=dev)
    xp = torch.cat([c['pitched'] for c in clips]).contiguous().to(dev)
    xu = torch.cat([c['unpitched'] for c in clips]).contiguous().to(dev)
    losses = torch.zeros(K, nat.N_LOSSES, device=dev)
    n = params.numel()
    stream = torch.cuda.Stream(dev)
    side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    # The iter_size = 2 accumulation iterations between optimizer steps are independent (same parameters, gradients
    # summed).  'streams': they run CONCURRENTLY on two side streams with two workspaces and two gradient buffers that
    # meet in the Adam kernel (g + g2 is bitwise what in-place accumulation gives).  'batched': they are the two clips of
    # one 2-clip plan, every launch carrying both.  One graph replay = iter_size iterations + Adam either way.
    ws = [plan.ws] if batched else [plan.ws, plan.new_ws()]
    if not batched:
        set_all(ws[1])
    grads = [gparams] if batched else [gparams, torch.zeros_like(gparams)]
    loss_out = [losses] if batched else [losses, torch.zeros_like(losses)]
    P = nat.ptr

    def iteration(j=0):
        plan.train_iteration(params, grads[j], xp, xu, loss_out[j], ws=ws[j])

    def pair():
        if batched:
            iteration(0)
            return
        if os.environ.get('MST_BENCH_SEQ'):      # experiment: accumulation iterations back to back on one stream
            iteration(0); iteration(1)
        else:
            for j in (0, 1):
                side[j].wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side[j]):
                    iteration(j)
            for j in (0, 1):
                torch.cuda.current_stream(dev).wait_stream(side[j])
        if dist is not None:                     # one buffer for the collective; part of the replayed graph
            grads[0].add_(grads[1])
            grads[1].zero_()

    def optimizer_step():
        st = nat.current_stream(dev)
        if dist is not None:
            dist.all_reduce(grads[0], op=dist.ReduceOp.SUM)          # sum, not mean: train-model.py:126,151-153
        if dist is not None or batched:
            nat.check(native.lib.mst_adam_step(P(params), P(grads[0]), P(m), P(v), n, P(state), .01, .9, .999, 1e-8, 200, .9,
                                               1, st), 'mst_adam_step')
        else:
            nat.check(native.lib.mst_adam_step2(P(params), P(grads[0]), P(grads[1]), P(m), P(v), n, P(state), .01, .9, .999,
                                                1e-8, 200, .9, 1, st), 'mst_adam_step2')

    graph_pair = None
    with torch.cuda.stream(stream):
        pair(); optimizer_step()                                      # load code objects, size RCCL buffers
        stream.synchronize()
        # --graphs per-lane (experiment): one hipGraph PER accumulation lane, each a single dependency chain replayed on its own
        # stream, and the Adam launches behind a two-event join.  hipGraph on ROCm 7.2 replays a single chain with ~2 us per node
        # boundary but a graph with parallel branches with ~5 us (tools/probe/capture_probe.cpp), so this was meant to keep the
        # cheap boundary and the overlap; measured 1998 it/s against 2298 for the one joint graph (three replays, two joins and
        # two launches per pair on the host side cost more than the boundaries save), so joint stays the default.
        lane_graphs = None
        if not args.no_graph and not batched and not os.environ.get('MST_BENCH_SEQ') and args.graphs == 'per-lane':
            lane_graphs = []
            for j in (0, 1):
                side[j].wait_stream(stream)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side[j]):
                    iteration(j)
                lane_graphs.append(g)
                stream.wait_stream(side[j])
        elif not args.no_graph:
            graph_pair = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_pair, stream=stream):
                pair()
                if dist is None:
                    optimizer_step()

        def two_steps():
            if lane_graphs is not None:
                for j in (0, 1):
                    side[j].wait_stream(stream)
                    with torch.cuda.stream(side[j]):
                        lane_graphs[j].replay()
                for j in (0, 1):
                    stream.wait_stream(side[j])
                if dist is not None:
                    grads[0].add_(grads[1])
                    grads[1].zero_()
                optimizer_step()
            elif graph_pair is not None:
                graph_pair.replay()
                if dist is not None:
                    optimizer_step()
            else:
                pair()
                optimizer_step()

        def run(nsteps):
            for _ in range(nsteps // iter_size):
                two_steps()
            if not batched:
                for _ in range(nsteps % iter_size):
                    iteration(0)                                      # odd tail: an accumulation iteration without a step

        if batched and args.steps % iter_size:
            raise SystemExit(f'--steps must be a multiple of {iter_size} clip-iterations in batched mode')
        run(args.warmup)

        def timed_region():
            """EXACTLY args.steps steps between barrier + synchronize on both sides; wall time, MAX over the ranks."""
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            run(args.steps)
            ev1.record(stream)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            dt_ = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt_], device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ = float(t.cpu()[0])
            return dt_, ev0.elapsed_time(ev1)

        # The region is repeated (whole regions of args.steps steps, never partial ones) until at least 0.25 s of timed work
        # and 3 regions have been seen: at the driver's --steps 20 a single region is ~9 ms, i.e. ten graph replays.  value is
        # the MEDIAN region; spread keeps the fastest and the slowest.  (Every rank derives the same repeat count from the
        # MAX-reduced time of the first region.)
        regions = [timed_region()]
        want = max(3, min(50, int(np.ceil(0.25 / max(regions[0][0], 1e-6)))))
        while len(regions) < want:
            regions.append(timed_region())
        order = sorted(range(len(regions)), key=lambda i: regions[i][0])
        dt, dev_ms = regions[order[len(order) // 2]]
        spread = dict(repeats=len(regions), min_ms_per_step=regions[order[0]][0] / args.steps * 1e3,
                      max_ms_per_step=regions[order[-1]][0] / args.steps * 1e3, value='median region')
        final_loss = float(losses.cpu()[0, 0])

        # ---- the same loop with the clip uploaded inside it (the reference's prepare_input moves every song to the device,
        # train-model.py:102-103): pinned host copies of the note tensors go to the device on a copy stream, double-buffered so
        # that the upload of pair i + 1 overlaps the compute of pair i; one graph per input buffer
        with_upload = None
        if rank == 0 and world == 1 and not batched and (graph_pair is not None or lane_graphs is not None):
            hxp, hxu = xp.cpu().pin_memory(), xu.cpu().pin_memory()
            bufs = [(xp, xu), (torch.empty_like(xp), torch.empty_like(xu))]
            graphs = [graph_pair, torch.cuda.CUDAGraph()]             # (this leg replays one joint graph per input buffer)
            xp_main, xu_main = xp, xu
            for j in (0, 1):
                if graphs[j] is not None:
                    continue
                graphs[j] = torch.cuda.CUDAGraph()
            for j in (0, 1):
                if graphs[j] is graph_pair:
                    continue
                xp, xu = bufs[j]
                with torch.cuda.graph(graphs[j], stream=stream):
                    pair()
                    optimizer_step()
            xp, xu = xp_main, xu_main
            copy = torch.cuda.Stream(dev)
            up_done = [torch.cuda.Event(), torch.cuda.Event()]
            read_done = [torch.cuda.Event(), torch.cuda.Event()]

            def run_with_upload(npairs):
                for i in range(npairs):
                    j = i & 1
                    with torch.cuda.stream(copy):
                        if i >= 2:
                            copy.wait_event(read_done[j])          # the replay that last read this buffer has finished
                        bufs[j][0].copy_(hxp, non_blocking=True)
                        bufs[j][1].copy_(hxu, non_blocking=True)
                        up_done[j].record(copy)
                    stream.wait_event(up_done[j])
                    graphs[j].replay()
                    read_done[j].record(stream)

            run_with_upload(max(2, args.warmup // ITER_SIZE))
            torch.cuda.synchronize()
            ups = []
            for _ in range(len(regions)):
                t0 = time.perf_counter()
                run_with_upload(args.steps // ITER_SIZE)
                torch.cuda.synchronize()
                ups.append(time.perf_counter() - t0)
            n_up = (args.steps // ITER_SIZE) * ITER_SIZE
            with_upload = dict(value=n_up / float(np.median(ups)), unit='iters/s', ms_per_step=float(np.median(ups)) / n_up * 1e3,
                               bytes_per_pair=(hxp.numel() + hxu.numel()) * 4,
                               note='clip H2D (pinned host memory, copy stream, double-buffered inputs) inside the timed loop; '
                                    'one upload per pair of accumulation iterations on the same clip')

        # ---- roofline leg: every launch step timed with HIP events on this stream (same workload)
        roof, table_rows = None, []
        if rank == 0 and not args.no_roofline:
            roof, table_rows = roofline_leg(plan, nat, params, gparams, xp, xu, K, dt / args.steps, args.breakdown, iter_size=iter_size)

    ips = world * args.steps / dt
    out = dict(metric='style-transfer opt iters/sec', value=ips, unit='iters/s', n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak',
               vs_baseline=None, dtype='f32', data='synthetic',
               config=dict(workload=(f'{B} different 30 s clips per GPU in one batched plan (BASELINE.json configs[2]/[3] shape), '
                                     f'optimizer step after every pass over the {B} clips; a step = one clip-iteration'
                                     if B > 1 else
                                     'one 30 s clip per GPU (BASELINE.json configs[1])') +
                                    ': piano-roll C=4,R=16,T=4 (+percussion), full widths (980325 params), fwd+loss+bwd every '
                                    'step, Adam+StepLR every iter_size steps',
                           clips_per_gpu=B, iter_size=iter_size, hip_graph=graph_pair is not None or lane_graphs is not None,
                           graphs=('one per accumulation lane (single chains) + Adam launches' if lane_graphs is not None else
                                   ('one joint graph' if graph_pair is not None else 'none')),
                           accumulation=('batched plan, %d clips per launch' % K) if batched else '2 concurrent streams',
                           launches_per_pass=plan.launch_count(7, False) + plan.launch_count(7, True) + 3,      # + 3 loss kernels
                           parallelism=f'dp{world} ({"RCCL" if args.backend == "nccl" else args.backend} all-reduce SUM of {n} fp32 grads per optimizer step)' if world > 1 else 'single GPU',
                           device_ms_per_step=dev_ms / args.steps, final_total_loss=final_loss),
               spread=spread)
    if with_upload is not None:
        out['value_with_upload'] = with_upload['value']
        out['with_upload'] = with_upload
    if world > 1 and B == 1 and not batched:
        # configs[4] beside the data-parallel headline: the long clip with its bars tiled over the same ranks (never lets the
        # headline line fail)
        try:
            tl = tiled_leg(native, nat, dev, flat, dist, rank, world, 10, 2)
        except Exception as e:                      # noqa: BLE001
            tl = dict(error=repr(e))
        if rank == 0:
            out['tiled'] = tl
    if rank == 0:
        out['roofline'] = roof
        out['kernel_breakdown'] = table_rows
        if world == 1 and B == 1 and not batched and args.batched_clips > 1:
            del plan, ws                     # the one-clip workspaces are no longer needed
            out['batched'] = batched_leg(native, nat, dev, flat, args.batched_clips, args.batched_passes, 3)
        if world == 1 and B == 1 and not batched and args.surface_steps > 0:
            out['surface'] = surface_leg(dev, clip, args.surface_steps, 20)
        if world == 1 and B == 1 and not batched and args.audio_steps > 0:
            try:
                out['audio_extension'] = audio_leg(nat, dev, args.audio_steps, 10, 0 if args.no_cpu_baseline else 4.0)
            except Exception as e:                      # noqa: BLE001  (an extension must never take the headline line down)
                out['audio_extension'] = dict(error=repr(e))
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(flat, table, clip, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
