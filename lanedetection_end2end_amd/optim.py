"""Fused Adam for the hot path's 226 trainable tensors: one HIP launch per step (``lf_adam_step``).

Same update rule and state names as ``torch.optim.Adam`` (``exp_avg``, ``exp_avg_sq``, ``step``), which is what the
reference's ``define_optim('adam', ...)`` returns (BEV/Networks/utils.py:411-420); amsgrad / maximize are not
supported.  Parameters without a gradient (``encoder.output_conv``) are skipped like torch does.
"""
import numpy as np
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)
        self._tables = {}

    def _table(self, gi, plist):
        """Device tables for one param group; rebuilt when a gradient buffer moved."""
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in plist)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2], cached[3]
        chunk = _lib.load().lf_adam_chunk()
        rec = np.zeros((len(plist), 5), dtype=np.int64)
        work = []
        for i, p in enumerate(plist):
            st = self.state[p]
            rec[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            work += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        dev = plist[0].device
        t_rec = torch.from_numpy(rec).to(dev)
        t_work = torch.tensor(work, dtype=torch.int32, device=dev)
        self._tables[gi] = (key, t_rec, t_work, len(work))
        return t_rec, t_work, len(work)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                    raise _lib.LaneFitLibraryError("FusedAdam needs contiguous fp32 parameters on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            steps = {self.state[p]["step"] for p in plist}
            assert len(steps) == 1, "parameters of one group must share the step count"
            step = steps.pop() + 1
            t_rec, t_work, nblocks = self._table(gi, plist)
            b1, b2 = group["betas"]
            _lib.check(lib.lf_adam_step(_lib.ptr(t_rec), _lib.ptr(t_work), nblocks, group["lr"], b1, b2, group["eps"],
                                        group["weight_decay"], step, self.grad_scale, _lib.stream()), "lf_adam_step")
            for p in plist:
                self.state[p]["step"] = step
        return loss
