"""Bar tiling of ONE long clip over several ranks (BASELINE.json configs[4]) on the hipsim build:
every "rank" is a plan for its bar tile; the phases run in lockstep and the exchanges are summed by hand (the
multi-process version over gloo is in test_dp_gloo.py).  The summed gradient and the losses must equal the one-rank
plan's and the oracle's."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity_cases as pc
from oracle import style_oracle as so
from tools.synth import synth_clip
from simutil import make_dims, rel, sim_native
from style import _native as nat


def run_tiled(native, device, widths, Cn, R, T, unp, tiles, clip, flat, poison=True):
    """Lockstep emulation of len(tiles) ranks; returns (sum of the ranks' gradients, per-rank losses, per-rank plans)."""
    dims = make_dims(widths, Cn, R, T, unp)
    plans, grads, losses, xp, xu = [], [], [], [], []
    params = flat.to(device)
    for r0, rows in tiles:
        p = nat.Plan(native, dims, device, tile_r0=r0, tile_rows=rows)
        pc.set_clip(p, clip)
        if poison:
            pc.poison(p)
        plans.append(p)
        grads.append(torch.zeros_like(params))
        losses.append(torch.zeros(nat.N_LOSSES, device=device))
        xp.append(clip['pitched'][:, :, r0:r0 + rows].contiguous().to(device))
        xu.append(clip['unpitched'][:, :, r0:r0 + rows].contiguous().to(device) if unp else None)
    n = native.lib.mst_tiled_phase_count(plans[0].handle)
    assert n > 3 and all(native.lib.mst_tiled_phase_count(p.handle) == n for p in plans)
    xoff, xlen, nx = (C.c_int64 * 8)(), (C.c_int64 * 8)(), C.c_int32()
    n_exchanges = 0
    for ph in range(n):
        spans = []
        for k, p in enumerate(plans):
            nat.check(native.lib.mst_tiled_phase(p.handle, ph, nat.ptr(params), nat.ptr(grads[k]), nat.ptr(p.ws), nat.ptr(xp[k]),
                                                 nat.ptr(xu[k]), nat.ptr(losses[k]), int(k == 0), nat.current_stream(device),
                                                 C.byref(xoff), C.byref(xlen), C.byref(nx)), 'mst_tiled_phase')
            spans.append([(xoff[q], xlen[q]) for q in range(nx.value)])
        # every rank exchanges the same buffers (their offsets may differ with the tile size: each workspace has its own layout)
        assert len(set(tuple(ln for _, ln in sp) for sp in spans)) == 1, spans
        if spans[0]:
            n_exchanges += 1                                    # the ranges of one phase end travel as one collective
        for q in range(len(spans[0])):
            ln = spans[0][q][1]
            total = sum(p.ws[sp[q][0]:sp[q][0] + ln] for p, sp in zip(plans, spans))
            for p, sp in zip(plans, spans):
                p.ws[sp[q][0]:sp[q][0] + ln] = total
    return sum(grads), losses, plans, n_exchanges


@pytest.mark.parametrize('tiles', [[(0, 3), (3, 3)], [(0, 2), (2, 3), (5, 1)]])
def test_tiled_clip_equals_one_rank_plan_and_oracle(tiles):
    native = sim_native()
    Cn, R, T, unp = 2, 6, 2, True
    dims = make_dims(pc.SMALL, Cn, R, T, unp)
    flat, named, table = pc.random_params(native, dims, 3)
    clip = synth_clip(21, Cn, R, T, unp, density=0.05)
    # one-rank plan
    plan = nat.Plan(native, dims, 'cpu')
    pc.set_clip(plan, clip)
    g1 = torch.zeros_like(flat)
    l1 = torch.zeros(nat.N_LOSSES)
    a, b = pc.dev_clip(clip, 'cpu')
    plan.train_iteration(flat.clone(), g1, a, b, l1)
    # tiled
    gt, lt, plans, nx = run_tiled(native, 'cpu', pc.SMALL, Cn, R, T, unp, tiles, clip, flat)
    for l in lt:
        assert torch.allclose(l, l1, atol=2e-6, equal_nan=True), (l, l1)
    assert rel(gt.numpy(), g1.numpy()) < 2e-5
    # every rank holds the same replicated tensors, and its own rows of the per-position ones
    for k, (r0, rows) in enumerate(tiles):
        assert rel(plans[k].view('style').numpy(), plan.view('style').numpy()) < 1e-5
        ref = plan.view('pitched_pred', (Cn, R, T * 10 * 56 * 5))[:, r0:r0 + rows]
        assert rel(plans[k].view('pitched_pred', (Cn, rows, T * 10 * 56 * 5)).numpy(), ref.numpy()) < 1e-5
    # oracle
    (info, xp_ref, xu_ref), ref_losses = so.iteration(named, clip)
    gref = torch.cat([(named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])).reshape(-1) for n, _, _ in table])
    assert rel(gt.numpy(), gref.numpy()) < pc.TOL
    for i, k in enumerate(nat.LOSS_KEYS):
        if k in ref_losses:
            assert abs(float(lt[0][i]) - ref_losses[k]) < 5e-5, k
    print("collectives per tiled iteration:", nx)
    assert nx <= 10         # 13 exchanged ranges travel as 10 collectives: an exchange is deferred until its first consumer


def test_tiled_pitched_only_single_tile_is_the_whole_clip():
    # one tile covering every bar: the tiled machinery (copies, folds, exchanges with nobody) must reproduce the plain plan
    native = sim_native()
    Cn, R, T = 3, 2, 2
    dims = make_dims(pc.SMALL, Cn, R, T, False)
    flat, named, table = pc.random_params(native, dims, 1)
    clip = synth_clip(4, Cn, R, T, False, density=0.05)
    gt, lt, plans, nx = run_tiled(native, 'cpu', pc.SMALL, Cn, R, T, False, [(0, 2)], clip, flat)
    (info, xp_ref, xu_ref), ref_losses = so.iteration(named, clip)
    gref = torch.cat([(named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])).reshape(-1) for n, _, _ in table])
    assert rel(gt.numpy(), gref.numpy()) < pc.TOL
