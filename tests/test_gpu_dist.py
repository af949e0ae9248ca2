"""-m gpu: the RCCL call path on hardware.  One GPU is all a gpurun box has, so the process group has world size 1:
`init_process_group('nccl', device_id=...)` + all_reduce(SUM) on the flat gradient buffer is the code bench.py
and FusedAdam run with N ranks (bench.py:124-130,195-204; style/optim.py), and must leave a one-rank step unchanged.
The N = 2 arithmetic (sum, not mean; identical step on every rank) is covered on CPU by tests/test_dp_gloo.py."""
import os

import numpy as np
import pytest
import torch

from tools.synth import synth_clip
from simutil import GOLDEN
from test_gpu_model_surface import load_small, reference_call, to_dev

pytestmark = pytest.mark.gpu


def test_rccl_world1_allreduce_step_equals_local_step():
    import torch.distributed as dist
    from style.optim import FusedAdam
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29400 + os.getpid() % 500))
    dev = torch.device('cuda:0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        results = []
        for use_group in (False, True):
            z, model = load_small('small_unpitched')
            C, R, T = (int(v) for v in z['crt'])
            opt = FusedAdam(model, process_group=dist.group.WORLD if use_group else None)
            for k in (0, 1):
                _, losses = reference_call(model, to_dev(synth_clip(k, C, R, T, True, density=float(z['density']))))
                losses['total'].backward()
            if use_group:
                before = model._gflat.clone()
                opt.all_reduce_grads()                      # RCCL all-reduce(SUM) over one rank: identity
                torch.cuda.synchronize()
                assert torch.equal(before, model._gflat)
            opt.step()
            torch.cuda.synchronize()
            results.append(model._flat.clone())
            for n, p in model.named_parameters():
                assert np.abs(p.detach().cpu().numpy() - z['p1/' + n]).max() < 3e-4, n
        assert torch.equal(results[0], results[1])
        # the collective as bench.py issues it: on the flat buffer, on a side stream, inside the step sequence
        g = torch.arange(980325, dtype=torch.float32, device=dev)
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        s.synchronize()
        assert float(g[-1]) == 980324.0
    finally:
        dist.destroy_process_group()
