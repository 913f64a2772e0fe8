"""Fused optimizer steps for the hot path's 226 trainable tensors: one HIP launch per step (``lf_adam_step`` /
``lf_sgd_step`` / ``lf_rmsprop_step``) -- the three optimizers the reference's ``define_optim`` builds
(BEV/Networks/utils.py:411-420): Adam, SGD(momentum 0.9), RMSprop(momentum 0.9); ``define_optim`` below mirrors it.

Same update rule and state names as ``torch.optim.Adam`` (``exp_avg``, ``exp_avg_sq``, ``step``), which is what the
reference's ``define_optim('adam', ...)`` returns (BEV/Networks/utils.py:411-420); amsgrad / maximize are not
supported.  Parameters without a gradient (``encoder.output_conv``) are skipped like torch does, and the step count
is PER PARAMETER like torch's: a tensor that gets its first gradient late (``decoder.output_conv`` when the reference's
pretrained schedule flips ``end_to_end`` under one optimizer, BEV/main.py get_flags) starts its own bias correction at 1.
A ``torch.optim.Adam`` state_dict (tensor-valued ``step``) loads as is.
"""
import numpy as np
import torch

from . import _lib


def _upload(host_array, device):
    """A small host table to the device WITHOUT stalling the stream: pinned staging + non-blocking copy.  (The engine hands out
    parameter gradients as views of a flat buffer that the caching allocator alternates between two addresses, so the record
    table is rebuilt every step; a pageable `.to(device)` there is a synchronous copy queued behind the whole backward pass --
    the host then waits for the GPU once per step and the next forward's launches start late: ~0.5 ms per step in
    `bench.py --workload epoch`, round 4.)  Returns (device tensor, pinned tensor to keep alive with it)."""
    pinned = torch.from_numpy(host_array).pin_memory()
    return pinned.to(device, non_blocking=True), pinned


def _work_list(numels, chunk, device, cache):
    """(tensor, chunk) work items for tensors of these sizes: depends on the sizes only, cached."""
    key = (tuple(numels), chunk, device)
    hit = cache.get(key)
    if hit is None:
        work = np.array([(i, c) for i, n in enumerate(numels) for c in range((n + chunk - 1) // chunk)], dtype=np.int32).reshape(-1, 2)
        dev, pin = _upload(work, device)
        hit = cache[key] = (dev, pin, len(work))
    return hit


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)
        self._tables = {}
        self._works = {}

    def load_state_dict(self, state_dict):
        """A loaded state replaces ``exp_avg`` / ``exp_avg_sq`` / ``step``: the cached device tables hold the OLD buffers' addresses
        and the old device-side step counts, so they are dropped (mid-run resume / rollback)."""
        super().load_state_dict(state_dict)
        self._tables = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = {}

    def _table(self, gi, plist):
        """Device tables for one param group: records {p, g, m, v, numel, step[2]} and the (tensor, chunk) work list.  Rebuilt --
        with the step counts of ``self.state`` in BOTH slots -- when the set of tensors with a gradient or any buffer address
        (parameter, gradient, moments) changed; between rebuilds the kernel advances the counts on the device (slot parity
        alternates per launch)."""
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                    for p in plist)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached
        chunk = _lib.load().lf_adam_chunk()
        rec = np.zeros((len(plist), 7), dtype=np.int64)
        for i, p in enumerate(plist):
            st = self.state[p]
            n = int(st["step"])
            rec[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), n, n)
        dev = plist[0].device
        t_work, _, nwork = _work_list([p.numel() for p in plist], chunk, dev, self._works)
        t_rec, pin = _upload(rec, dev)
        cached = [key, t_rec, t_work, nwork, 0, pin]
        self._tables[gi] = cached
        return cached

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
                elif torch.is_tensor(st["step"]):              # a loaded torch.optim.Adam state
                    st["step"] = int(st["step"])
            tab = self._table(gi, plist)
            _, t_rec, t_work, nblocks, parity = tab[:5]
            b1, b2 = group["betas"]
            _lib.check(lib.lf_adam_step(_lib.ptr(t_rec), _lib.ptr(t_work), nblocks, group["lr"], b1, b2, group["eps"],
                                        group["weight_decay"], parity, self.grad_scale, _lib.stream()), "lf_adam_step")
            tab[4] = parity ^ 1
            for p in plist:
                self.state[p]["step"] += 1
        return loss


class _FusedMomentum(torch.optim.Optimizer):
    """Shared plumbing of FusedSGD / FusedRMSprop: state buffers, device tables (the layout of FusedAdam's), one launch."""
    _state_names = ()

    def __init__(self, params, defaults, grad_scale):
        super().__init__(params, defaults)
        self.grad_scale = float(grad_scale)
        self._tables = {}
        self._works = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # new state buffers: the cached device tables point at the old ones
        self._tables = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = {}

    def _table(self, gi, plist):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) + tuple(self.state[p][n].data_ptr() for n in self._state_names) for p in plist)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached
        chunk = _lib.load().lf_adam_chunk()
        rec = np.zeros((len(plist), 7), dtype=np.int64)
        for i, p in enumerate(plist):
            st = self.state[p]
            m = st["momentum_buffer"]
            v = st["square_avg"] if "square_avg" in st else m
            rec[i] = (p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 0, 0)
        dev = plist[0].device
        t_work, _, nwork = _work_list([p.numel() for p in plist], chunk, dev, self._works)
        t_rec, pin = _upload(rec, dev)
        cached = (key, t_rec, t_work, nwork, pin)
        self._tables[gi] = cached
        return cached

    def _launch(self, group, t_rec, t_work, nblocks):
        raise NotImplementedError

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                    raise _lib.LaneFitLibraryError("fused optimizers need contiguous fp32 parameters on the GPU")
                st = self.state[p]
                for name in self._state_names:
                    if st.get(name) is None:
                        st[name] = torch.zeros_like(p)
                st["step"] = st.get("step", 0) + 1
            _, t_rec, t_work, nblocks = self._table(gi, plist)[:4]
            self._launch(group, t_rec, t_work, nblocks)
        return loss


class FusedSGD(_FusedMomentum):
    """``torch.optim.SGD(params, lr, momentum=0.9, weight_decay)`` as the reference builds it (utils.py:414-415): dampening 0,
    no Nesterov.  State name ``momentum_buffer`` as torch's."""
    _state_names = ("momentum_buffer",)

    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay), grad_scale)

    def _launch(self, group, t_rec, t_work, nblocks):
        _lib.check(_lib.load().lf_sgd_step(_lib.ptr(t_rec), _lib.ptr(t_work), nblocks, group["lr"], group["momentum"],
                                           group["weight_decay"], self.grad_scale, _lib.stream()), "lf_sgd_step")


class FusedRMSprop(_FusedMomentum):
    """``torch.optim.RMSprop(params, lr, momentum=0.9, weight_decay)`` (utils.py:416-417): alpha 0.99, eps 1e-8, not centered.
    State names ``square_avg`` / ``momentum_buffer`` as torch's."""
    _state_names = ("square_avg", "momentum_buffer")

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, momentum=0.9, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, momentum=momentum, weight_decay=weight_decay), grad_scale)

    def _launch(self, group, t_rec, t_work, nblocks):
        _lib.check(_lib.load().lf_rmsprop_step(_lib.ptr(t_rec), _lib.ptr(t_work), nblocks, group["lr"], group["alpha"],
                                               group["eps"], group["momentum"], group["weight_decay"], self.grad_scale,
                                               _lib.stream()), "lf_rmsprop_step")


def define_optim(optim, params, lr, weight_decay):
    """The reference's optimizer factory (BEV/Networks/utils.py:411-420) on the fused steps: same names, same settings."""
    if optim == 'adam':
        return FusedAdam(params, lr=lr, weight_decay=weight_decay)
    if optim == 'sgd':
        return FusedSGD(params, lr=lr, momentum=0.9, weight_decay=weight_decay)
    if optim == 'rmsprop':
        return FusedRMSprop(params, lr=lr, momentum=0.9, weight_decay=weight_decay)
    raise KeyError("The requested optimizer: {} is not implemented".format(optim))
