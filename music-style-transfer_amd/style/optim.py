"""Fused Adam + StepLR over the model's flat parameter buffer (train-model.py:89-90,151-154).

`torch.optim.Adam(model.parameters())` keeps working on this package's model (the parameters are
ordinary nn.Parameters whose .grad aliases the flat gradient buffer); this class is the HIP
equivalent of `optimizer.step(); optimizer.zero_grad(); scheduler.step()` in one launch, with
the step counter and the per-step scalars kept on the device so that it is graph-replayable.
For data parallelism pass `process_group`: gradients are all-reduced with SUM (RCCL) — the
reference accumulates gradients without averaging (train-model.py:126,151-153).
"""
import torch

from style import _native


class FusedAdam:
    def __init__(self, model, lr=.01, betas=(.9, .999), eps=1e-8, step_size=200, gamma=.9, process_group=None):
        model._sync_flat()
        self.model = model
        self.lr, self.betas, self.eps, self.step_size, self.gamma = lr, betas, eps, step_size, gamma
        self.exp_avg = torch.zeros_like(model._flat)
        self.exp_avg_sq = torch.zeros_like(model._flat)
        self.state = torch.zeros(4, dtype=torch.float32, device=model._flat.device)
        self.process_group = process_group
        # consecutive StyleTransferModel.train_iteration calls may now overlap on two lanes (two gradient buffers): this
        # optimizer joins them in step()
        model.concurrent_accumulation = True

    def all_reduce_grads(self):
        if self.process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                              and torch.distributed.get_world_size() > 1):
            torch.distributed.all_reduce(self.model._gflat, op=torch.distributed.ReduceOp.SUM, group=self.process_group)

    def step(self, zero_grad=True):
        m = self.model
        m._sync_flat()
        g2 = m.join_lanes() if hasattr(m, 'join_lanes') else None
        if g2 is not None and (self.process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                                  and torch.distributed.get_world_size() > 1)):
            m._gflat.add_(g2); g2.zero_(); g2 = None          # one buffer for the collective
        self.all_reduce_grads()
        n = m._flat.numel()
        P = _native.ptr
        if g2 is not None:
            _native.check(_native.get().lib.mst_adam_step2(P(m._flat), P(m._gflat), P(g2), P(self.exp_avg), P(self.exp_avg_sq), n,
                                                           P(self.state), self.lr, self.betas[0], self.betas[1], self.eps,
                                                           self.step_size, self.gamma, int(zero_grad),
                                                           _native.current_stream(m._flat.device)), 'mst_adam_step2')
            if not zero_grad:
                m._gflat.add_(g2); g2.zero_()                 # keep the accumulated sum visible in p.grad
            return
        _native.check(_native.get().lib.mst_adam_step(P(m._flat), P(m._gflat), P(self.exp_avg), P(self.exp_avg_sq), n,
                                                      P(self.state), self.lr, self.betas[0], self.betas[1], self.eps,
                                                      self.step_size, self.gamma, int(zero_grad),
                                                      _native.current_stream(m._flat.device)), 'mst_adam_step')

    def zero_grad(self):
        g2 = self.model.join_lanes() if hasattr(self.model, 'join_lanes') else None
        if g2 is not None:
            g2.zero_()
        self.model._gflat.zero_()
