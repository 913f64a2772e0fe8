"""torch.autograd.Function wrappers around the C ABI (fitting head and losses).

Each Function only marshals pointers: all arithmetic happens in liblanefit_hip.so.
"""
import ctypes

import torch

from . import _lib

ACT_KINDS = {"square": 0, "abs": 1, "relu": 2, "sigmoid": 3, "softplus": 4, "none": 5}
WEIGHT_FUNCTS = {"none": 0, "linear": 1, "quadratic": 2}


class SingularMatrixError(RuntimeError):
    """Raised like torch.inverse's error so `except RuntimeError` in the reference's main.py
    (BEV/main.py:213-219) keeps skipping the batch."""


def _raise_if_singular(status, solver):
    bad = int(status.max().item())       # D2H sync, by design: the reference raises synchronously
    if bad:
        idx = int((status != 0).nonzero()[0, 0])
        if bad == 2:
            raise SingularMatrixError("lanefit WLS: normal matrix of (image,lane) #%d is not positive-definite "
                                      "(Cholesky/GELS path)" % idx)
        raise SingularMatrixError("lanefit WLS: normal matrix of (image,lane) #%d is singular, the inversion "
                                  "could not be completed" % idx)


class WLSFit(torch.autograd.Function):
    """(logits NCHW fp32, grid (P,2)|(N,P,2) fp32) -> beta (N,K,order+1) fp64 [, masked (N,K,H,W) fp32]."""

    @staticmethod
    def forward(ctx, logits, grid, zero_rows, order, reg, y_offset, act_kind, solver, want_masked, check):
        lib = _lib.load()
        logits = logits.contiguous()
        assert logits.dtype == torch.float32 and logits.dim() == 4
        N, K, H, W = logits.shape
        grid = grid.contiguous()
        assert grid.dtype == torch.float32 and grid.shape[-2:] == (H * W, 2), (grid.shape, H, W)
        gbs = H * W * 2 if grid.dim() == 3 and grid.shape[0] > 1 else 0
        if grid.dim() == 3 and grid.shape[0] > 1:
            assert grid.shape[0] >= N
        D = order + 1
        dev = logits.device
        beta = torch.empty(N, K, D, dtype=torch.float64, device=dev)
        zinv = torch.empty(N, K, D * D, dtype=torch.float64, device=dev)
        status = torch.empty(N * K, dtype=torch.int32, device=dev)
        masked = torch.empty_like(logits) if want_masked else None
        ws = torch.empty(lib.lf_wls_workspace_bytes(N, K, order), dtype=torch.uint8, device=dev)
        _lib.check(lib.lf_wls_fwd(_lib.ptr(logits), _lib.ptr(grid), gbs, N, K, H, W, zero_rows, order, float(reg),
                                  float(y_offset), act_kind, solver, _lib.ptr(beta), _lib.ptr(zinv),
                                  _lib.ptr(masked), _lib.ptr(ws), _lib.ptr(status), _lib.stream()), "lf_wls_fwd")
        if check:
            _raise_if_singular(status, solver)
        ctx.save_for_backward(logits, grid, beta, zinv)
        ctx.cfg = (gbs, zero_rows, order, float(y_offset), act_kind)
        ctx.status = status
        # (no zero tensors for the outputs nobody differentiates: autograd otherwise fills an (N, K, H, W) fp32 gradient for `masked`
        # every step -- 205 MB, 27 us at config 3)
        ctx.set_materialize_grads(False)
        if want_masked:
            ctx.mark_non_differentiable(masked)
            return beta, masked, status
        return beta, None, status

    @staticmethod
    def backward(ctx, gbeta, _gm, _gs):
        lib = _lib.load()
        logits, grid, beta, zinv = ctx.saved_tensors
        gbs, zero_rows, order, y_offset, act_kind = ctx.cfg
        N, K, H, W = logits.shape
        if gbeta is None:          # the coefficients were not used downstream
            return torch.zeros_like(logits), None, None, None, None, None, None, None, None, None
        gbeta = gbeta.to(torch.float64).contiguous()
        gl = torch.empty_like(logits)
        _lib.check(lib.lf_wls_bwd(_lib.ptr(logits), _lib.ptr(grid), gbs, N, K, H, W, zero_rows, order, y_offset,
                                  act_kind, _lib.ptr(beta), _lib.ptr(zinv), _lib.ptr(gbeta), _lib.ptr(gl),
                                  _lib.stream()), "lf_wls_bwd")
        return gl, None, None, None, None, None, None, None, None, None


class AreaLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, gt, order, weight_funct):
        lib = _lib.load()
        p = params.squeeze(-1) if params.dim() == 3 else params
        if p.stride(1) != 1:       # (rows of one lane out of an (N, K, D) tensor go in as they are: the kernel takes the row stride)
            p = p.contiguous()
        gt = gt.to(p.dtype).contiguous()
        assert p.dtype in (torch.float32, torch.float64) and p.shape == gt.shape and p.shape[1] == order + 1
        loss = torch.empty((), dtype=p.dtype, device=p.device)
        grad = torch.empty(p.shape, dtype=p.dtype, device=p.device)
        _lib.check(lib.lf_area_loss(_lib.ptr(p, rows=True), p.stride(0), _lib.ptr(gt), p.shape[0], order, weight_funct,
                                    1 if p.dtype == torch.float64 else 0, _lib.ptr(loss), _lib.ptr(grad),
                                    _lib.stream()), "lf_area_loss")
        ctx.save_for_backward(grad)
        ctx.pshape = params.shape
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout).view(ctx.pshape), None, None, None


class MSELossFn(torch.autograd.Function):
    """mean((params - gt)^2) over all elements and its gradient in one launch (lf_mse_loss)."""

    @staticmethod
    def forward(ctx, params, gt):
        lib = _lib.load()
        p = params.contiguous()
        q = gt.to(p.dtype).contiguous()
        if p.dtype not in (torch.float32, torch.float64) or p.shape != q.shape:
            raise RuntimeError("MSE_Loss: params %s %s vs gt %s" % (tuple(p.shape), p.dtype, tuple(q.shape)))
        loss = torch.empty((), dtype=p.dtype, device=p.device)
        grad = torch.empty_like(p)
        _lib.check(lib.lf_mse_loss(_lib.ptr(p), _lib.ptr(q), p.numel(), 1 if p.dtype == torch.float64 else 0, _lib.ptr(loss),
                                   _lib.ptr(grad), _lib.stream()), "lf_mse_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return grad * gout, None


class BackprojLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, x_gt, valid, Y, y_prime, minv):
        lib = _lib.load()
        p = params.squeeze(-1) if params.dim() == 3 else params
        p = p.to(torch.float64)
        if p.stride(1) != 1:       # (a lane's rows out of an (N, K, D) tensor go in as they are: beta_stride)
            p = p.contiguous()
        x_gt = x_gt.to(torch.float64).contiguous()
        valid = valid.to(torch.float64).contiguous()
        N, D = p.shape
        S = x_gt.shape[1]
        loss = torch.empty((), dtype=torch.float64, device=p.device)
        xcv = torch.empty(N, S, dtype=torch.float64, device=p.device)
        grad = torch.empty(N, D, dtype=torch.float64, device=p.device)
        m = (ctypes.c_double * 9)(*[float(v) for v in minv.reshape(-1)])
        _lib.check(lib.lf_backproj_loss(_lib.ptr(p, rows=True), p.stride(0), _lib.ptr(x_gt), _lib.ptr(valid), _lib.ptr(Y),
                                        _lib.ptr(y_prime), ctypes.cast(m, ctypes.c_void_p), N, S, D - 1,
                                        _lib.ptr(loss), _lib.ptr(xcv), _lib.ptr(grad), _lib.stream()),
                   "lf_backproj_loss")
        ctx.save_for_backward(grad)
        ctx.pshape, ctx.pdtype = params.shape, params.dtype
        ctx.mark_non_differentiable(xcv)
        ctx.set_materialize_grads(False)       # (no zero-filled (N, S) gradient for xcv on every call)
        return loss, xcv

    @staticmethod
    def backward(ctx, gout, _gx):
        (grad,) = ctx.saved_tensors
        if gout is None:
            return torch.zeros(ctx.pshape, dtype=ctx.pdtype, device=grad.device), None, None, None, None, None
        return (grad * gout).view(ctx.pshape).to(ctx.pdtype), None, None, None, None, None


class CrossEntropy2dFn(torch.autograd.Function):
    """``check_targets``: read back the kernel's count of labels outside [0, C) (one D2H sync) and raise like torch's
    NLLLoss does; with False the count stays in ``CrossEntropy2dFn.last_acc[2]`` for the caller to inspect."""
    last_acc = None

    @staticmethod
    def forward(ctx, logits, target, weights, check_targets=True):
        lib = _lib.load()
        logits = logits.contiguous()
        target = target.contiguous()
        weights = weights.to(device=logits.device, dtype=torch.float32).contiguous()
        assert logits.dtype == torch.float32 and target.dtype == torch.int64
        N, C, H, W = logits.shape
        assert target.shape == (N, H, W), (target.shape, logits.shape)
        acc = torch.empty(3, dtype=torch.float64, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        _lib.check(lib.lf_ce2d_fwd(_lib.ptr(logits), _lib.ptr(target), _lib.ptr(weights), N, C, H, W,
                                   _lib.ptr(acc), _lib.ptr(loss), _lib.stream()), "lf_ce2d_fwd")
        CrossEntropy2dFn.last_acc = acc
        if check_targets and float(acc[2]) != 0.0:
            raise RuntimeError("cross entropy: %d target value(s) outside [0, %d)" % (int(acc[2]), C))
        ctx.save_for_backward(logits, target, weights, acc)
        return loss

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        logits, target, weights, acc = ctx.saved_tensors
        N, C, H, W = logits.shape
        up = gout.to(torch.float32).reshape(1).contiguous()
        g = torch.empty_like(logits)
        _lib.check(lib.lf_ce2d_bwd(_lib.ptr(logits), _lib.ptr(target), _lib.ptr(weights), N, C, H, W,
                                   _lib.ptr(acc), _lib.ptr(up), _lib.ptr(g), _lib.stream()), "lf_ce2d_bwd")
        return g, None, None, None


class LinearFn(torch.autograd.Function):
    """``act(x @ w.T + b)`` with ``act`` = identity or ReLU: the nn.Linear tails of the --clas heads on lf_linear_fwd / lf_linear_bwd
    (fp32, fixed summation order) instead of F.linear / rocBLAS."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        lib = _lib.load()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w = w.contiguous()
        assert x2.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[1] == x2.shape[1]
        N, K = x2.shape
        O = w.shape[0]
        bb = None if b is None else b.contiguous()
        y = torch.empty(N, O, dtype=torch.float32, device=x2.device)
        _lib.check(lib.lf_linear_fwd(_lib.ptr(x2), _lib.ptr(w), _lib.ptr(bb), _lib.ptr(y), N, K, O, int(bool(relu)), _lib.stream()),
                   "lf_linear_fwd")
        ctx.save_for_backward(x2, w, y)
        ctx.relu, ctx.xshape, ctx.has_bias = bool(relu), x.shape, b is not None
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x2, w, y = ctx.saved_tensors
        N, K = x2.shape
        O = w.shape[0]
        gy2 = gy.reshape(N, O).to(torch.float32).contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = torch.empty_like(x2) if need_x else None
        gw = torch.empty_like(w) if (need_w or need_b) else None
        gb = torch.empty(O, dtype=torch.float32, device=w.device) if need_b else None
        _lib.check(lib.lf_linear_bwd(_lib.ptr(x2), _lib.ptr(w), _lib.ptr(y), _lib.ptr(gy2), _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb),
                                     N, K, O, int(ctx.relu), _lib.stream()), "lf_linear_bwd")
        return (gx.view(ctx.xshape) if need_x else None), (gw if need_w else None), gb, None


def linear(x, w, b=None, relu=False):
    return LinearFn.apply(x, w, b, relu)


def seg_maps(logits, gt_line, zero_rows, lanes):
    """Segmentation-mode fit input (lf_seg_maps): arg-max of the class logits -> per-lane maps valued k at class k, masked rows
    zeroed, lanes flagged in ``gt_line`` (N, lanes; BP only) overwritten with map [0, 0].  No gradient (the reference detaches)."""
    lib = _lib.load()
    logits = logits.detach().contiguous()
    assert logits.dtype == torch.float32 and logits.dim() == 4
    N, C, H, W = logits.shape
    flags = None
    if gt_line is not None:
        flags = gt_line.to(device=logits.device, dtype=torch.float32).contiguous()
        if tuple(flags.shape) != (N, lanes):
            # the reference's expand_as(masked) fails here -- but only when gt_line.sum() != 0 lets it get that far
            if float(flags.sum()) != 0:
                raise RuntimeError("seg-mode fit: gt_line %s cannot be expanded to the (%d, %d) lane maps" % (tuple(flags.shape), N, lanes))
            flags = None
    maps = torch.empty(N, lanes, H, W, dtype=torch.float32, device=logits.device)
    _lib.check(lib.lf_seg_maps(_lib.ptr(logits), _lib.ptr(flags), _lib.ptr(maps), N, C, lanes, H, W, int(zero_rows), _lib.stream()),
               "lf_seg_maps")
    return maps
