"""The ``--clas`` line-type / horizon heads and the inference-side lane decoding.

``Classification`` mirrors BP/Networks/LSQ_layer.py:150-207: four Conv-BatchNorm-ReLU blocks on the encoder output, a
pooling layer, fully connected layers.  The BEV tree's class (BEV/Networks/LSQ_layer.py:170-228) has the same trunk and
the same horizon head but a DIFFERENT line head -- four ``fully_connected_line{1..4}`` ``Linear(128, 3)`` whose outputs
are concatenated to (N, 3, 4), consumed by ``nn.CrossEntropyLoss`` (BEV/main.py:88,252) -- ``ClassificationBEV`` below.
The conv trunk runs as ONE C-ABI call per direction (``lf_convchain_forward`` / ``_backward``) directly on
the NHWC encoder output inside the backbone's workspace; pooling + NCHW flatten is ``lf_poolflat_*``; the
``nn.Linear`` layers are plain library GEMMs.

``Projections`` mirrors BP/test.py:128-186 and ``decode_lanes`` fuses ``compute_coordinates`` for all lanes
with the gating of ``test_model`` (BP/test.py:72-88) into one launch (``lf_lane_decode``).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, geometry, ops


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class _ChainPlan:
    def __init__(self, N, H, W, channels, ksize):
        lib = _lib.load()
        ch = (ctypes.c_int * len(channels))(*channels)
        ks = (ctypes.c_int * len(ksize))(*ksize)
        self.handle = lib.lf_convchain_plan_create(N, H, W, len(ksize), ch, ks)
        if not self.handle:
            raise _lib.LaneFitLibraryError("lf_convchain_plan_create failed: %s" % lib.lf_last_error().decode())
        self.shape = (N, H, W)
        self.channels = tuple(channels)
        self.ws_bytes = lib.lf_convchain_workspace_bytes(self.handle)

    def __del__(self):
        try:
            _lib.load().lf_convchain_plan_destroy(self.handle)
        except Exception:
            pass


class _ConvChainFn(torch.autograd.Function):
    """x: (N,H,W,C0) NHWC fp32 -> relu(bn_L(conv_L(... relu(bn_1(conv_1(x)))))) NHWC."""

    @staticmethod
    def forward(ctx, mod, plan, x, training, *params):
        lib = _lib.load()
        N, H, W = plan.shape
        ws = torch.empty(plan.ws_bytes, dtype=torch.uint8, device=x.device)
        y = torch.empty(N, H, W, plan.channels[-1], dtype=torch.float32, device=x.device)
        params = [p.detach() for p in params]
        for p in params:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        bns = mod._batchnorms()
        running = _ptr_array([b for bn in bns for b in (bn.running_mean, bn.running_var)])
        _lib.check(lib.lf_convchain_forward(plan.handle, _lib.ptr(x), _ptr_array(params), _lib.ptr(mod._ptr_table(params)),
                                            running, int(training), float(bns[0].momentum), float(bns[0].eps), _lib.ptr(y),
                                            _lib.ptr(ws), plan.ws_bytes, _lib.stream()), "lf_convchain_forward")
        ctx.plan, ctx.ws, ctx.x, ctx.y, ctx.params = plan, ws, x, y, params
        ctx.training = int(training)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        plan, params = ctx.plan, ctx.params
        gy = gy.contiguous()
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=gy.device)
        grads, off = [], 0
        for p in params:
            grads.append(flat[off: off + p.numel()].view(p.shape))
            off += p.numel()
        gx = torch.empty_like(ctx.x) if ctx.needs_input_grad[2] else None
        _lib.check(lib.lf_convchain_backward(plan.handle, _lib.ptr(ctx.x), _lib.ptr(ctx.y), _lib.ptr(gy), _ptr_array(params),
                                             _ptr_array(grads), _lib.ptr(gx), ctx.training, _lib.ptr(ctx.ws), plan.ws_bytes,
                                             _lib.stream()), "lf_convchain_backward")
        ctx.ws = None
        return (None, None, gx, None) + tuple(grads)


class _PoolFlatFn(torch.autograd.Function):
    """NHWC (N,H,W,C) -> (N, features) in NCHW flatten order; mode 0 = MaxPool2d(2,2), 1 = AvgPool2d((1,W))."""

    @staticmethod
    def forward(ctx, y, mode):
        lib = _lib.load()
        N, H, W, C = y.shape
        feat = C * (H // 2) * (W // 2) if mode == 0 else C * H
        out = torch.empty(N, feat, dtype=torch.float32, device=y.device)
        _lib.check(lib.lf_poolflat_fwd(_lib.ptr(y), N, H, W, C, mode, _lib.ptr(out), _lib.stream()), "lf_poolflat_fwd")
        ctx.save_for_backward(y)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        N, H, W, C = y.shape
        gy = torch.empty_like(y)
        _lib.check(lib.lf_poolflat_bwd(_lib.ptr(y), _lib.ptr(g.contiguous()), N, H, W, C, ctx.mode, _lib.ptr(gy),
                                       _lib.stream()), "lf_poolflat_bwd")
        return gy, None


class Classification(nn.Module):
    """``Classification(class_type, size, channels_in, resize)`` with the reference's submodule names, so its
    state_dict loads into / from the reference's."""

    def __init__(self, class_type, size, channels_in, resize):
        super().__init__()
        self.class_type = class_type
        self.conv1 = nn.Conv2d(channels_in, 128, 1, stride=1, padding=0, bias=True)
        self.conv1_bn = nn.BatchNorm2d(128)
        self.conv2 = nn.Conv2d(128, 128, 3, stride=1, padding=1, bias=True)
        self.conv2_bn = nn.BatchNorm2d(128)
        self.conv3 = nn.Conv2d(128, 64, 3, stride=1, padding=1, bias=True)
        self.conv3_bn = nn.BatchNorm2d(64)
        self.conv4 = nn.Conv2d(64, 64, 3, stride=1, padding=1, bias=True)
        self.conv4_bn = nn.BatchNorm2d(64)
        rows, cols = size
        self.size = (rows, cols)
        self.avgpool = nn.AvgPool2d((1, cols))
        self.maxpool = nn.MaxPool2d((2, 2), stride=2)
        if class_type == 'line':
            self.fully_connected1 = nn.Linear(64 * rows * cols // 4, 128)
            self._make_line_heads()
        else:
            self.fully_connected_horizon = nn.Linear(64 * rows, resize)
        self._channels = (channels_in, 128, 128, 64, 64)
        self._plans = {}
        self._ptr_cache = (None, None)

    def _make_line_heads(self):
        """BP: one 4-way head (BP/Networks/LSQ_layer.py:186-187)."""
        self.fully_connected_line1 = nn.Linear(128, 4)

    def _line_logits(self, f):
        return ops.linear(f, self.fully_connected_line1.weight, self.fully_connected_line1.bias)

    def _batchnorms(self):
        return [self.conv1_bn, self.conv2_bn, self.conv3_bn, self.conv4_bn]

    def _trunk_params(self):
        out = []
        for conv, bn in zip((self.conv1, self.conv2, self.conv3, self.conv4), self._batchnorms()):
            out += [conv.weight, conv.bias, bn.weight, bn.bias]
        return out

    def _ptr_table(self, params):
        key = tuple(p.data_ptr() for p in params)
        if self._ptr_cache[0] != key:
            self._ptr_cache = (key, torch.tensor(key, dtype=torch.int64, device=params[0].device))
        return self._ptr_cache[1]

    def trunk(self, x):
        """conv1..conv4 (+BN+ReLU) on a logical-NCHW tensor; returns NHWC (N,H,W,64)."""
        if not x.is_cuda:
            raise _lib.LaneFitLibraryError("lanefit Classification needs its input on the MI355X; there is no CPU path")
        xh = x.permute(0, 2, 3, 1)
        if not xh.is_contiguous():           # the backbone hands out channels-last memory: normally a no-op
            xh = xh.contiguous()
        xh = xh.float()
        N, H, W, C = xh.shape
        assert C == self._channels[0]
        key = (N, H, W)
        if key not in self._plans:
            self._plans[key] = _ChainPlan(N, H, W, self._channels, (1, 3, 3, 3))
        y = _ConvChainFn.apply(self, self._plans[key], xh, self.training, *self._trunk_params())
        if self.training:
            torch._foreach_add_([m.num_batches_tracked for m in self._batchnorms()], 1)
        return y

    def forward(self, x):
        y = self.trunk(x)
        if self.class_type == 'line':
            f = _PoolFlatFn.apply(y, 0)
            f = ops.linear(f, self.fully_connected1.weight, self.fully_connected1.bias, relu=True)      # F.relu(fc1(f)), one launch
            return self._line_logits(f)
        f = _PoolFlatFn.apply(y, 1)
        return ops.linear(f, self.fully_connected_horizon.weight, self.fully_connected_horizon.bias)


class ClassificationBEV(Classification):
    """The BEV tree's ``Classification`` (BEV/Networks/LSQ_layer.py:170-228): same trunk and horizon head; the line
    head is four 3-way classifiers ``fully_connected_line1..4`` (``Linear(128, 3)`` each, :198-205) whose logits are
    stacked to (N, 3, 4) -- class axis 1, lane axis 2 (:218-226) -- for ``nn.CrossEntropyLoss`` (BEV/main.py:88,252).
    The four heads keep their own parameters (``state_dict`` keys of the reference) and run as ONE (128 -> 12) GEMM."""

    def _make_line_heads(self):
        for i in range(1, 5):
            setattr(self, "fully_connected_line%d" % i, nn.Linear(128, 3))

    def _line_logits(self, f):
        heads = [getattr(self, "fully_connected_line%d" % i) for i in range(1, 5)]
        y = torch.stack([ops.linear(f, h.weight, h.bias) for h in heads], 2)      # four (N, 3) GEMVs (lf_linear_fwd) -> (N, 3, 4)
        return y


def resize_coordinates(array):
    """BP/test.py:20-21."""
    return array * 2.5


class Projections:
    """``Projections(options)``: sample heights 160..710 step 10 of the 1280x720 frame mapped into the
    bird's-eye view; ``compute_coordinates(params)`` evaluates one lane's polynomial there and projects
    back (BP/test.py:128-186).  ``decode_lanes`` does all lanes + the test-time gating in one launch."""

    def __init__(self, options):
        M, M_inv = geometry.get_homography(resize=options.resize, no_mapping=False)
        self.M, self.M_inv = torch.from_numpy(M), torch.from_numpy(M_inv)
        self.order = options.order
        if self.order < 0 or self.order > 3:
            raise NotImplementedError(
                'Requested order {} for polynomial fit is not implemented'.format(self.order))
        start, delta = 160, 10
        self.num_heights = (720 - start) // delta
        self.y_d = (torch.arange(start, 720, delta) - 80).double() / 2.5
        self.y_prime = (self.M[1, 1:2] * self.y_d + self.M[1, 2:]) / (self.M[2, 1:2] * self.y_d + self.M[2, 2:])
        self.y_eval = 255 - self.y_prime
        self._minv = (ctypes.c_double * 9)(*[float(v) for v in self.M_inv.double().reshape(-1)])
        self._dev = None

    def _device_consts(self, device):
        if self._dev is None or self._dev[0].device != device:
            self._dev = (self.y_eval.to(device).contiguous(), self.y_prime.double().to(device).contiguous())
        return self._dev

    def _decode(self, beta, line_flag, bound, lo, hi, fill, want_int):
        lib = _lib.load()
        if not beta.is_cuda:
            raise _lib.LaneFitLibraryError("lanefit Projections needs its input on the MI355X; there is no CPU path")
        N, L, K = beta.shape
        assert K == self.order + 1
        beta = beta.double().contiguous()
        y_eval, y_prime = self._device_consts(beta.device)
        S = self.num_heights
        x = torch.empty(N, L, S, dtype=torch.float64, device=beta.device)
        xi = torch.empty(N, L, S, dtype=torch.int32, device=beta.device) if want_int else None
        if line_flag is not None:
            line_flag = line_flag.float().contiguous()
        if bound is not None:
            bound = bound.to(torch.int32).contiguous()
        _lib.check(lib.lf_lane_decode(_lib.ptr(beta), _lib.ptr(y_eval), _lib.ptr(y_prime), self._minv, 2.5,
                                      _lib.ptr(line_flag), _lib.ptr(bound), lo, hi, fill, N, L, S, self.order,
                                      _lib.ptr(x), _lib.ptr(xi), _lib.stream()), "lf_lane_decode")
        return x, xi

    def compute_coordinates(self, params):
        """params (N, order+1, 1) fp64 -> x coordinates (N, 56) in the 1280-wide frame."""
        x, _ = self._decode(params.reshape(params.size(0), 1, -1), None, None, 1.0, 0.0, -2.0, False)
        return x[:, 0]

    def decode_lanes(self, betas, line_pred=None, horizon_pred=None):
        """``betas``: the per-lane (N, order+1, 1) tensors in model order; ``line_pred`` (N, 4) the rounded
        line-type sigmoid in the dataset's order (re-indexed [1,2,0,3] as test_model does);
        ``horizon_pred`` (N,) int horizon row.  Returns (lanes (N,L,56) fp64, rounded int32 copy) with -2
        wherever test_model writes -2 (BP/test.py:77-88)."""
        beta = torch.stack([b.reshape(b.size(0), -1) for b in betas], 1)
        flag = None if line_pred is None else line_pred[:, [1, 2, 0, 3]][:, : beta.size(1)]
        bound = None
        if horizon_pred is not None:
            bound = torch.div(horizon_pred.to(torch.int64) - 160, 10, rounding_mode='trunc')
        return self._decode(beta, flag, bound, 0.0, 1279.0, -2.0, True)


def horizon_row(outputs_horizon):
    """BP/test.py:62-63: sigmoid row votes -> horizon row in the 720-high frame, snapped to the 10 px grid."""
    pred = torch.sigmoid(outputs_horizon).sum(dim=1)
    return (torch.round((resize_coordinates(pred) + 80) / 10) * 10).int()


def line_flags(outputs_line):
    """BP/test.py:64."""
    return torch.round(torch.sigmoid(outputs_line))
