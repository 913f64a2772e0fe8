"""Loss modules with the reference's names and call signatures (BEV/Loss_crit.py, BP/Loss_crit.py),
computed by liblanefit_hip.so."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, geometry, ops


class polynomial():
    """Trapezoid-rule area metric of the reference (BEV/Loss_crit.py:12-35), vectorised.

    Works on whatever device the coefficients live on (the reference calls it on ``.cpu()``
    tensors after every step, BEV/main.py:273-280; passing GPU tensors removes those syncs).
    """

    def __init__(self, coeffs, a=0, b=0.7, n=100):
        c = coeffs.reshape(coeffs.shape[0], -1)
        self.a1, self.b1, self.c1 = c[:, 0], c[:, 1], c[:, 2]
        self.a, self.b, self.n = a, b, n

    def calc_pol(self, x):
        return self.a1 * x ** 2 + self.b1 * x + self.c1

    def trapezoidal(self, other):
        if self.a1.is_cuda:
            # one launch, no host sync: sums in the reference's order and dtype (lf_trapezoid)
            lib = _lib.load()
            dt = self.a1.dtype if self.a1.dtype in (torch.float32, torch.float64) else torch.float32
            p = torch.stack((self.a1, self.b1, self.c1), 1).to(dt).contiguous()
            q = torch.stack((other.a1, other.b1, other.c1), 1).to(dt).to(p.device).contiguous()
            out = torch.empty(p.shape[0], dtype=dt, device=p.device)
            _lib.check(lib.lf_trapezoid(_lib.ptr(p), _lib.ptr(q), p.shape[0], float(self.a), float(self.b), int(self.n),
                                        int(dt == torch.float64), _lib.ptr(out), _lib.stream()), "lf_trapezoid")
            return out
        # host tensors (the reference calls this on .cpu() copies, BEV/main.py:274-279): same rule, vectorised
        h = float(self.b - self.a) / self.n
        xs = self.a + h * torch.arange(0, self.n + 1, device=self.a1.device, dtype=self.a1.dtype)
        d = (self.calc_pol(xs[:, None]) - other.calc_pol(xs[:, None])).abs()
        w = torch.ones_like(xs)
        w[0] = w[-1] = 0.5
        return (d * w[:, None]).sum(0) * h


class Area_Loss(nn.Module):
    """Integral of the squared x-difference between fitted and ground-truth curves over
    y in [0, 0.7], three weightings; ``forward(params, gt_params, compute=True)``
    (BEV/Loss_crit.py:78-134).  Deviation: when no lane is kept the reference returns the
    Python int 0; this returns a zero tensor (still differentiable, gradient 0)."""

    def __init__(self, order, weight_funct):
        super().__init__()
        if order not in (1, 2):
            raise NotImplementedError('The requested order is not implemented')
        if weight_funct not in ops.WEIGHT_FUNCTS:
            raise NotImplementedError('The requested weight function is not implemented')
        self.order = order
        self.weight_funct = weight_funct

    def forward(self, params, gt_params, compute=True):
        return ops.AreaLossFn.apply(params, gt_params, self.order, ops.WEIGHT_FUNCTS[self.weight_funct])


class MSE_Loss(nn.Module):
    """``--loss_policy mse``: ``nn.MSELoss()`` of ``params.squeeze(-1)`` against ``gt_params`` -- the mean over all
    N x (order + 1) elements (BEV/Loss_crit.py:137-150, BP/Loss_crit.py:147-160); ``forward(params, gt_params, compute=True)``.
    One launch for the loss and its gradient (``lf_mse_loss``); like ``nn.MSELoss`` it wants equal shapes after the squeeze."""

    def __init__(self, options=None):
        super().__init__()

    def forward(self, params, gt_params, compute=True):
        return ops.MSELossFn.apply(params.squeeze(-1), gt_params)


class CrossEntropyLoss2d(nn.Module):
    """Class-weighted pixel cross entropy (BEV/Loss_crit.py:61-75): weights [1, w, w],
    target = targets[:, 0].  ``nclasses`` generalises to BP's [1] + [w]*nclasses (:64)."""

    def __init__(self, weight=None, size_average=True, seg=False, nclasses=2):
        super().__init__()
        w = [1.0] + [float(weight)] * nclasses if seg else [1.0] * (nclasses + 1)
        self.register_buffer("weights", torch.tensor(w, dtype=torch.float32), persistent=False)
        # A label outside [0, C) raises like nn.NLLLoss's device assert does -- and, like it, not inside the offending call:
        # the kernel counts such labels (they carry weight 0); the count of call k is copied to pinned host memory right behind
        # the kernel (``_lib.DeferredRead``) and read at the START of call k + 1 (or by flush()), waiting for that copy's event
        # only -- the loss adds no host sync to the step.  "always": read it in the call itself (one sync per step);
        # False: never (the count stays in ops.CrossEntropy2dFn.last_acc[2]).
        # The LAST call of a loop is inspected by flush(): the mirrors' loops call it after the last batch; train() / eval()
        # on the criterion flush too.
        self.check_targets = True
        self._pending = None

    def flush(self):
        """Raise now if the previous call saw a label outside [0, C)."""
        pend, self._pending = self._pending, None
        if pend is not None:
            acc = pend[0].get()
            if float(acc[2]) != 0.0:
                raise RuntimeError("cross entropy: %d target value(s) outside [0, %d)" % (int(acc[2]), pend[1]))

    def train(self, mode=True):
        self.flush()
        return super().train(mode)

    def forward(self, inputs, targets):
        if targets.dim() == 4:
            targets = targets[:, 0, :, :]
        if self.check_targets:
            self.flush()
        loss = ops.CrossEntropy2dFn.apply(inputs, targets.long(), self.weights, self.check_targets == "always")
        if self.check_targets:
            self._pending = (_lib.DeferredRead(ops.CrossEntropy2dFn.last_acc), inputs.shape[1])
        return loss


class backprojection_loss(nn.Module):
    """MSE in image space after back-projecting 56 sampled curve points through M^-1
    (BP/Loss_crit.py:161-218).  ``forward(params, x_gt, valid_samples) -> (loss, x_cal*valid)``."""

    def __init__(self, options):
        super().__init__()
        M, M_inv = geometry.get_homography(options.resize, getattr(options, "no_mapping", False))
        self.M, self.M_inv = torch.from_numpy(M), np.ascontiguousarray(M_inv, dtype=np.float64)
        order = options.order
        if order < 0 or order > 3:
            raise NotImplementedError(
                'Requested order {} for polynomial fit is not implemented'.format(order))
        y_d = (torch.arange(160, 720, 10) - 80).double() / 2.5           # :170-173 (literal 80 / 2.5)
        y_prime = (M[1, 1] * y_d + M[1, 2]) / (M[2, 1] * y_d + M[2, 2])   # :175
        y_eval = 255 - y_prime                                            # :176 (literal 255)
        Y = torch.stack([y_eval ** (order - j) for j in range(order + 1)], 1)
        dev = "cpu" if getattr(options, "no_cuda", False) else "cuda"
        self.Y = Y.contiguous().to(dev)
        self.y_prime = y_prime.contiguous().to(dev)
        self.order = order

    def forward(self, params, x_gt, valid_samples):
        return ops.BackprojLossFn.apply(params, x_gt, valid_samples, self.Y, self.y_prime, self.M_inv)


def define_loss_crit_bev(options):
    """BEV/Loss_crit.py:45-58."""
    if options.loss_policy == 'mse':
        crit = MSE_Loss(options)
    elif options.loss_policy == 'area':
        crit = Area_Loss(options.order, options.weight_funct)
    else:
        return NotImplementedError('The requested loss criterion is not implemented')
    seg = CrossEntropyLoss2d(options.weight_seg, seg=True)
    return crit, (seg if getattr(options, "no_cuda", False) else seg.cuda())


def define_loss_crit_bp(options):
    """BP/Loss_crit.py:47-67."""
    if options.loss_policy == 'mse':
        crit = MSE_Loss(options)
    elif options.loss_policy == 'backproject':
        crit = backprojection_loss(options)
    elif options.loss_policy == 'area':
        crit = Area_Loss(options.order, options.weight_funct)
    else:
        return NotImplementedError('The requested loss criterion is not implemented')
    seg = CrossEntropyLoss2d(options.weight_seg, seg=True, nclasses=options.nclasses)
    return crit, (seg if getattr(options, "no_cuda", False) else seg.cuda())
