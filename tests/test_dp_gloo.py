"""World-size-2 data parallelism on CPU (gloo): one clip per rank, all-reduce(SUM) of the flat gradient
buffer, identical Adam step on every rank == the reference's 2-clip gradient accumulation + Adam step
(train-model.py:126,151-154), checked against the fixture the reference produced.  The per-rank
kernels are the product's .hip sources on the hipsim interpreter (no GPU here)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'music-style-transfer_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle.synth import synth_clip
    from parity_cases import SMALL, set_clip
    from simutil import GOLDEN, flat_from_named, make_dims, sim_native
    from style import _native as nat
    z = np.load(os.path.join(GOLDEN, 'small_unpitched.npz'))
    C, R, T = (int(v) for v in z['crt'])
    native = sim_native()
    dims = make_dims(SMALL, C, R, T, True)
    params, table = flat_from_named(native, dims, {k[3:]: z[k] for k in z.files if k.startswith('p0/')})
    plan = nat.Plan(native, dims, 'cpu')
    clip = synth_clip(rank, C, R, T, True, density=float(z['density']))        # one clip per rank
    set_clip(plan, clip)
    g = torch.zeros_like(params)
    plan.train_iteration(params, g, clip['pitched'].contiguous(), clip['unpitched'].contiguous())
    dist.all_reduce(g, op=dist.ReduceOp.SUM)                                   # sum, not mean
    m, v, state = torch.zeros_like(params), torch.zeros_like(params), torch.zeros(4)
    nat.check(native.lib.mst_adam_step(nat.ptr(params), nat.ptr(g), nat.ptr(m), nat.ptr(v), params.numel(), nat.ptr(state),
                                       .01, .9, .999, 1e-8, 200, .9, 1, None), 'adam')
    worst = max(float(np.abs(params[o:o + int(np.prod(s))].numpy() - z['p1/' + n].reshape(-1)).max()) for n, o, s in table)
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        torch.save(dict(worst=worst, same=same), out)
    dist.destroy_process_group()


def test_two_rank_sum_allreduce_equals_reference_accumulation(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from simutil import sim_native
    sim_native()                      # build the interpreter library once, before forking
    out = str(tmp_path / 'dp.pt')
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['same'], 'ranks diverged after the all-reduced step'
    assert r['worst'] < 3e-4, r['worst']
