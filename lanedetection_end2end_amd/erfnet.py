"""ERFNet backbone as an ``nn.Module`` whose forward/backward run in liblanefit_hip.so.

Drop-in for the reference's ``Networks.ERFNet.Net`` (BEV/Networks/ERFNet.py:145-157): same
constructor, same ``forward(input, flag, only_encode=False)``, same ``state_dict()`` keys and
shapes (so reference checkpoints load and ``define_init_weights`` -- which matches on the class
names 'Conv' / 'BatchNorm2d', BEV/Networks/utils.py:458-503 -- initialises it identically).
The torch sub-modules below only HOLD parameters and buffers; no torch convolution is ever
called.  One ``torch.autograd.Function`` spans the whole backbone: its forward and backward
are single C-ABI calls (``lf_erfnet_forward`` / ``lf_erfnet_backward``).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib

_BN_EPS = 1e-3


# cumulative down-sampling in front of every layer of the plan (module order: encoder.initial_block, encoder.layers[0..14],
# decoder.layers[0..5]): input size of layer i = (H / s_i, W / s_i) of the network input
_LAYER_IN_STRIDE = [1, 2] + [4] * 5 + [4] + [8] * 8 + [8, 4, 4, 4, 2, 2]


class _Holder(nn.Module):
    """A sub-module of the backbone.  It holds its parameters with the reference's names; ``forward`` runs the module's layer
    range of the engine plan (``lf_erfnet_forward_range``) at the input's size -- the same kernels as the whole-network pass --
    so ``model.net.encoder(x)``, ``model.net.decoder.layers[1](y)`` ... work like the reference's modules
    (BEV/Networks/ERFNet.py:19-22,44-60,86-95,104-107,129-142).  Bound to its ``Net`` by ``Net._bind_blocks``."""
    _lf_range = None

    def _owner(self):
        ref = self.__dict__.get("_lf_owner")
        net = ref() if ref is not None else None
        if net is None or self._lf_range is None:
            raise RuntimeError("lanefit ERFNet block is not part of a Net (blocks run inside their network's engine plan)")
        return net

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_lf_owner", None)            # weak reference: re-bound by Net.__setstate__
        return st

    def _run_range(self, x, head=-1):
        return self._owner()._forward_range(x, self._lf_range[0], self._lf_range[1], head)

    def forward(self, input):
        return self._run_range(input)


class DownsamplerBlock(_Holder):
    """conv3x3 s2 (ninput -> noutput-ninput) || maxpool2 -> BN -> ReLU (ERFNet.py:11-22)."""

    def __init__(self, ninput, noutput):
        super().__init__()
        self.conv = nn.Conv2d(ninput, noutput - ninput, (3, 3), stride=2, padding=1, bias=True)
        self.pool = nn.MaxPool2d(2, stride=2)
        self.bn = nn.BatchNorm2d(noutput, eps=_BN_EPS)


class non_bottleneck_1d(_Holder):
    """Factorised residual block: 3x1, 1x3, BN, dilated 3x1, 1x3, BN, Dropout2d (ERFNet.py:25-60)."""

    def __init__(self, chann, dropprob, dilated):
        super().__init__()
        d = dilated
        self.conv3x1_1 = nn.Conv2d(chann, chann, (3, 1), padding=(1, 0), bias=True)
        self.conv1x3_1 = nn.Conv2d(chann, chann, (1, 3), padding=(0, 1), bias=True)
        self.bn1 = nn.BatchNorm2d(chann, eps=_BN_EPS)
        self.conv3x1_2 = nn.Conv2d(chann, chann, (3, 1), padding=(d, 0), bias=True, dilation=(d, 1))
        self.conv1x3_2 = nn.Conv2d(chann, chann, (1, 3), padding=(0, d), bias=True, dilation=(1, d))
        self.bn2 = nn.BatchNorm2d(chann, eps=_BN_EPS)
        self.dropout = nn.Dropout2d(dropprob)
        self._built_with_dropout = dropprob > 0      # the engine plan has a mask slot for these blocks


class UpsamplerBlock(_Holder):
    """ConvTranspose2d 3x3 s2 -> BN -> ReLU (ERFNet.py:98-107)."""

    def __init__(self, ninput, noutput):
        super().__init__()
        self.conv = nn.ConvTranspose2d(ninput, noutput, 3, stride=2, padding=1, output_padding=1, bias=True)
        self.bn = nn.BatchNorm2d(noutput, eps=_BN_EPS)


class Encoder(_Holder):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.initial_block = DownsamplerBlock(in_channels, 16)
        blocks = [DownsamplerBlock(16, 64)] + [non_bottleneck_1d(64, 0.03, 1) for _ in range(5)]
        blocks.append(DownsamplerBlock(64, 128))
        blocks += [non_bottleneck_1d(128, 0.3, d) for d in (2, 4, 8, 16, 2, 4, 8, 16)]
        self.layers = nn.ModuleList(blocks)
        # only used by the reference's only_encode=True path; never trained (ERFNet.py:84,92-93)
        self.output_conv = nn.Conv2d(128, num_classes, 1, stride=1, padding=0, bias=True)

    def forward(self, input, predict=False):
        """ERFNet.py:86-95: initial_block, the 15 layers, and with ``predict`` the 1x1 ``output_conv``."""
        output = self._run_range(input)
        if predict:
            output = _PointwiseFn.apply(output.permute(0, 2, 3, 1).contiguous(), self.output_conv.weight, self.output_conv.bias)
        return output


class Decoder(_Holder):
    def __init__(self, num_classes, pretrain):
        super().__init__()
        self.pretrain = pretrain
        self.layers = nn.ModuleList([
            UpsamplerBlock(128, 64), non_bottleneck_1d(64, 0, 1), non_bottleneck_1d(64, 0, 1),
            UpsamplerBlock(64, 16), non_bottleneck_1d(16, 0, 1), non_bottleneck_1d(16, 0, 1)])
        self.output_conv = nn.ConvTranspose2d(16, num_classes, 2, stride=2, padding=0, output_padding=0, bias=True)
        if pretrain:
            self.output_conv2 = nn.ConvTranspose2d(16, num_classes + 1, 2, stride=2, padding=0, output_padding=0,
                                                   bias=True)

    def forward(self, input, flag):
        """ERFNet.py:129-142: the six layers + ``output_conv`` (``output_conv2`` when pretrain and not flag).  The BP tree's
        decoder returns ``(output, output_seg)`` with ``output_seg`` = its input (BP/Networks/ERFNet.py:143-163)."""
        head = 1 if (self.pretrain and not flag) else 0
        output = self._run_range(input, head)
        if self._owner().three_outputs:
            return output, input
        return output


class _Plan:
    """Host-side engine plan for one input shape (owns the C object)."""

    def __init__(self, N, H, W, cin, cout, n_heads):
        lib = _lib.load()
        self.handle = lib.lf_erfnet_plan_create(N, H, W, cin, cout, n_heads)
        if not self.handle:
            raise _lib.LaneFitLibraryError("lf_erfnet_plan_create: " + lib.lf_last_error().decode())
        self.handle = ctypes.c_void_p(self.handle)
        self.n_params = lib.lf_erfnet_num_params(self.handle)
        self.n_bn = lib.lf_erfnet_num_bn(self.handle)
        self.n_drop = lib.lf_erfnet_num_dropout(self.handle)
        self.drop_floats = lib.lf_erfnet_dropmask_floats(self.handle)
        self.drop_off = [lib.lf_erfnet_dropmask_offset(self.handle, i) for i in range(self.n_drop)]
        self.drop_ch = [lib.lf_erfnet_dropmask_channels(self.handle, i) for i in range(self.n_drop)]
        self.enc_off = lib.lf_erfnet_encoder_offset(self.handle)
        self.shape = (N, H, W)

    def workspace_bytes(self, precision="fp32"):
        """Bytes of the whole-network workspace in a precision mode (bf16 tensors need larger partial-row regions): independent
        of the mode the shared plan was last set to."""
        return _lib.load().lf_erfnet_workspace_bytes_for(self.handle, _PRECISIONS[precision])

    def __del__(self):
        try:
            _lib.load().lf_erfnet_plan_destroy(self.handle)
        except Exception:
            pass


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class _PtrCache:
    """ctypes pointer arrays of tensor lists that rarely move (parameters, BatchNorm buffers, the gradient views of the flat
    buffer the caching allocator hands back at the same address every step): rebuilt only when an address changed."""

    def __init__(self):
        self._store = {}

    def get(self, tag, tensors):
        key = tuple(0 if t is None else t.data_ptr() for t in tensors)
        hit = self._store.get(tag)
        if hit is None or hit[0] != key:
            arr = (ctypes.c_void_p * len(key))(*[k or None for k in key])
            hit = (key, arr)
            self._store[tag] = hit
        return hit[1]


# lf_erfnet_set_precision modes (include/lanefit.h)
_PRECISIONS = {"fp32": 0, "bf16": 2, "fp32x9": 3}      # (1 and 4 were bf16_mfma / fp32x6: removed in round 6, no BASELINE config used them)


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, plan, x, head, training, dropmask, *params):
        lib = _lib.load()
        N, H, W = plan.shape
        dev = x.device
        ctx.precision = _PRECISIONS[net.precision]
        _lib.check(lib.lf_erfnet_set_precision(plan.handle, ctx.precision), "lf_erfnet_set_precision")
        ctx.ws_bytes = lib.lf_erfnet_workspace_bytes_for(plan.handle, ctx.precision)      # (the bf16 weight gradient has larger partial-row regions)
        ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=dev)
        # head = -1: encoder only (only_encode=True): no decoder launches, no logits
        logits = (torch.empty(N, net.out_channels + head, H, W, dtype=torch.float32, device=dev) if head >= 0 else
                  torch.empty(0, dtype=torch.float32, device=dev))
        params = [p.detach() for p in params]
        for p in params:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        host = net._ptrs.get("params", params)
        devarr = net._device_ptr_table(params)
        running = net._ptrs.get("running", net._running_buffers())
        _lib.check(lib.lf_erfnet_forward(plan.handle, _lib.ptr(x), host, _lib.ptr(devarr), running,
                                         _lib.ptr(dropmask), int(training), head, _lib.ptr(logits) if head >= 0 else None,
                                         _lib.ptr(ws), ctx.ws_bytes, _lib.stream()), "lf_erfnet_forward")
        ctx.set_materialize_grads(False)
        if head < 0:
            ctx.mark_non_differentiable(logits)
        if net.export_encoder_output and ctx.precision == 2:
            # bf16 tensors: the in-place view is bf16 and takes no gradient (the --clas heads are an fp32 path)
            nenc = N * (H // 8) * (W // 8) * 128
            enc = ws[4 * plan.enc_off: 4 * plan.enc_off + 2 * nenc].view(torch.bfloat16).view(N, H // 8, W // 8, 128)
            ctx.mark_non_differentiable(enc)
        elif net.export_encoder_output:
            # the encoder output is handed out IN PLACE: an NHWC view of the workspace (no copy, no transpose);
            # Net.forward permutes it to the reference's logical (N,128,H/8,W/8)
            nenc = N * (H // 8) * (W // 8) * 128
            enc = ws[4 * plan.enc_off: 4 * (plan.enc_off + nenc)].view(torch.float32).view(N, H // 8, W // 8, 128)
        else:
            enc = torch.empty(0, dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(enc)
        ctx.net, ctx.plan, ctx.head, ctx.ws, ctx.x, ctx.dropmask = net, plan, head, ws, x, dropmask
        ctx.params = params
        ctx.training = int(training)        # BatchNorm backward follows the mode the forward ran in
        return logits, enc

    @staticmethod
    def backward(ctx, glogits, genc):
        lib = _lib.load()
        N, H, W = ctx.plan.shape
        if ctx.head < 0:         # encoder-only forward: the incoming gradient is the encoder output's
            glogits = None
            if genc is None:
                genc = torch.zeros(N, H // 8, W // 8, 128, dtype=torch.float32, device=ctx.x.device)
        elif glogits is None:    # only the encoder output was used downstream
            glogits = torch.zeros(N, ctx.net.out_channels + ctx.head, H, W, dtype=torch.float32, device=ctx.x.device)
        if genc is not None:
            genc = genc.contiguous()
        plan, params = ctx.plan, ctx.params
        needs = ctx.needs_input_grad[6:]
        used = ctx.net._used_param_mask(ctx.head)
        # all parameter gradients are views of ONE flat buffer (module order): a data-parallel run all-reduces
        # it in place, with no flatten / copy-back kernels (dp.FlatGradAllReduce picks it up via flat_grad())
        want = [bool(n and u) for n, u in zip(needs, used)]
        # The flat buffer always spans ALL parameters (module order) plus dp.TAIL reducer slots, so its size never depends on
        # which parameters take part: unused ones (encoder.output_conv, the other head) leave zero-filled holes and keep
        # grad = None; the data-parallel bucket is this very buffer (dp.FlatGradAllReduce: fixed-size collective).
        from .dp import TAIL
        total = sum(p.numel() for p in params)
        flat = torch.empty(total + TAIL, dtype=torch.float32, device=params[0].device)
        grads, off, hole = [], 0, None
        for p, w in zip(params, want):
            if w:
                grads.append(flat[off: off + p.numel()].view(p.shape))
                if hole is not None:
                    flat[hole: off].zero_()
                    hole = None
            else:
                grads.append(None)
                hole = off if hole is None else hole
            off += p.numel()
        flat[(total if hole is None else hole):].zero_()          # trailing hole + the reducer's tail
        ctx.net._flat_grad = flat
        if glogits is not None:
            glogits = glogits.contiguous()
        _lib.check(lib.lf_erfnet_set_precision(plan.handle, ctx.precision), "lf_erfnet_set_precision")
        _lib.check(lib.lf_erfnet_backward(plan.handle, _lib.ptr(ctx.x), _lib.ptr(glogits), _lib.ptr(genc),
                                          ctx.net._ptrs.get("params", params), ctx.net._ptrs.get("grads", grads), _lib.ptr(ctx.dropmask), ctx.training,
                                          ctx.head, _lib.ptr(ctx.ws), ctx.ws_bytes, _lib.stream()), "lf_erfnet_backward")
        ctx.ws = None
        return (None, None, None, None, None, None) + tuple(grads)


class _PointwiseFn(torch.autograd.Function):
    """``encoder.output_conv`` (Conv2d(128, K, 1)) on the NHWC encoder output -> NCHW (lf_pointwise_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, enc_nhwc, weight, bias):
        lib = _lib.load()
        N, h, w, C = enc_nhwc.shape
        K = weight.shape[0]
        x = enc_nhwc.contiguous()
        wt = weight.detach().reshape(K, C).contiguous()
        y = torch.empty(N, K, h, w, dtype=torch.float32, device=x.device)
        _lib.check(lib.lf_pointwise_fwd(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(bias.detach().contiguous()), _lib.ptr(y), N, h, w, C, K,
                                        _lib.stream()), "lf_pointwise_fwd")
        ctx.save_for_backward(x, wt)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, wt = ctx.saved_tensors
        N, h, w, C = x.shape
        K = wt.shape[0]
        gy = gy.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(wt) if ctx.needs_input_grad[1] else None
        gb = torch.empty(K, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None     # independent of gw
        scratch = torch.empty(lib.lf_pointwise_scratch_floats(N, h, w, C, K), dtype=torch.float32, device=x.device)
        _lib.check(lib.lf_pointwise_bwd(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(wt), _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb), N, h, w,
                                        C, K, _lib.ptr(scratch), _lib.stream()), "lf_pointwise_bwd")
        return gx, (None if gw is None else gw.view(ctx.wshape)), gb


class _RangeFn(torch.autograd.Function):
    """Layers [first, last) of the plan (+ optional head) on an NCHW tensor: lf_erfnet_forward_range / _backward_range.

    The call owns a COMPACT workspace (the range's activations + the plan's globals, ``lf_erfnet_range_workspace_bytes``) until its
    backward: ``for l in net.encoder.layers: x = l(x)`` keeps the activations of one network, not a whole-network workspace per
    block.  With ``first == 0`` the input is the image and receives no gradient (``gx`` is None), as in the full pass."""

    @staticmethod
    def forward(ctx, net, plan, x, first, last, head, training, dropmask, *params):
        lib = _lib.load()
        N, H, W = plan.shape
        io = (ctypes.c_int * 6)()
        _lib.check(lib.lf_erfnet_layer_io(plan.handle, last - 1, io), "lf_erfnet_layer_io")
        oshape = (N, net.out_channels + head, H, W) if head >= 0 else (N, io[3], io[4], io[5])
        ctx.ws_bytes = lib.lf_erfnet_range_workspace_bytes(plan.handle, first, last)
        ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=x.device)
        y = torch.empty(oshape, dtype=torch.float32, device=x.device)
        params = [p.detach() for p in params]
        host = net._ptrs.get("params", params)
        devarr = net._device_ptr_table(params)
        ctx.precision = _PRECISIONS[net.precision]
        _lib.check(lib.lf_erfnet_set_precision(plan.handle, ctx.precision), "lf_erfnet_set_precision")
        running = net._ptrs.get("running", net._running_buffers())
        _lib.check(lib.lf_erfnet_forward_range(plan.handle, first, last, head, _lib.ptr(x), host, _lib.ptr(devarr), running,
                                               _lib.ptr(dropmask), int(training), _lib.ptr(y), _lib.ptr(ws), ctx.ws_bytes,
                                               _lib.stream()), "lf_erfnet_forward_range")
        ctx.set_materialize_grads(False)
        ctx.net, ctx.plan, ctx.cfg, ctx.ws, ctx.x, ctx.dropmask, ctx.params = net, plan, (first, last, head), ws, x, dropmask, params
        ctx.training = int(training)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        first, last, head = ctx.cfg
        if gy is None:
            return (None,) * (8 + len(ctx.params))
        gy = gy.contiguous().float()
        used = ctx.net._range_param_mask(first, last, head)
        needs = ctx.needs_input_grad[8:]
        grads = [torch.empty_like(p) if (n and u) else None for p, n, u in zip(ctx.params, needs, used)]
        gx = torch.empty_like(ctx.x) if (ctx.needs_input_grad[2] and first > 0) else None
        _lib.check(lib.lf_erfnet_set_precision(ctx.plan.handle, ctx.precision), "lf_erfnet_set_precision")
        _lib.check(lib.lf_erfnet_backward_range(ctx.plan.handle, first, last, head, _lib.ptr(ctx.x), _lib.ptr(gy),
                                                _ptr_array(ctx.params), _ptr_array(grads), _lib.ptr(ctx.dropmask), ctx.training,
                                                _lib.ptr(gx), _lib.ptr(ctx.ws), ctx.ws_bytes, _lib.stream()),
                   "lf_erfnet_backward_range")
        ctx.ws = None
        return (None, None, gx, None, None, None, None, None) + tuple(grads)


class Net(nn.Module):
    """``Net(layers=18, in_channels=1, out_channels=1, pretrained=False, pool=False)`` and
    ``forward(input, flag, only_encode=False) -> (encoder_output, decoder_output)``
    exactly as BEV/Networks/ERFNet.py:145-157.  ``three_outputs=True`` gives the BP variant's
    ``(encoder_output, decoder_output, output_seg)`` (BP/Networks/ERFNet.py:170-176), where ``output_seg`` is the encoder
    output itself (the decoder's second head is never built, :128-163).

    ``only_encode=True`` returns ``encoder.forward(input, predict=True)`` = ``output_conv`` (1x1, 128 -> num_classes) of the
    encoder output, (N, num_classes, H/8, W/8) (ERFNet.py:86-95,151-153); the engine stops after the encoder (head = -1:
    the decoder's BatchNorm running statistics and ``num_batches_tracked`` stay untouched, as in the reference), and the
    result is differentiable through the encoder like the ``--clas`` heads' input.

    Deviations: ``encoder_output`` is a channels-last view of the engine's workspace (logical shape (N,128,H/8,W/8) as in
    the reference, differentiable: the ``--clas`` heads train through it); no gradient is produced for the input image.
    """
    three_outputs = False

    def __init__(self, layers=18, in_channels=1, out_channels=1, pretrained=False, pool=False):
        super().__init__()
        self.encoder = Encoder(in_channels, out_channels)
        self.decoder = Decoder(out_channels, pretrained)
        self.in_channels, self.out_channels, self.pretrained = in_channels, out_channels, bool(pretrained)
        self._plans = {}
        self._ptr_cache = (None, None)
        self._ptrs = _PtrCache()
        self._flat_grad = None
        # precision mode: "fp32" (default, the parity path and the BASELINE headline); "bf16" (bf16 matrix cores AND bf16
        # activation / gradient tensors in HBM: BASELINE config 3 -- the reference has no such mode); "fp32x9" (fp32 tensors
        # and accumulation, the products of the 64- / 128-channel convs formed on the bf16 matrix cores from exact 3-way bf16
        # splits of both operands, all 9 partial products: fp32 accuracy, held to the fp32 tolerances by the tests).
        # Parameters, their gradients and the logits are fp32 in every mode.
        self.precision = "fp32"
        # encoder_output (N,128,H/8,W/8) is part of the return tuple (zero-copy view); wrappers that never read it
        # may switch it off
        self.export_encoder_output = True
        self._bind_blocks()

    def _blocks(self):
        """The plan's layers in module order -> the modules that own them."""
        return [self.encoder.initial_block] + list(self.encoder.layers) + list(self.decoder.layers)

    def _bind_blocks(self):
        import weakref
        ref = weakref.ref(self)
        blocks = self._blocks()
        for i, b in enumerate(blocks):
            b.__dict__["_lf_owner"], b._lf_range = ref, (i, i + 1)
        ne = 1 + len(self.encoder.layers)
        self.encoder.__dict__["_lf_owner"], self.encoder._lf_range = ref, (0, ne)
        self.decoder.__dict__["_lf_owner"], self.decoder._lf_range = ref, (ne, len(blocks))

    def __setstate__(self, state):
        super().__setstate__(state)
        self._bind_blocks()

    def _range_param_mask(self, first, last, head):
        """Which parameters (module order) a layer range reads: its blocks', plus the chosen head's."""
        own = set()
        for b in self._blocks()[first:last]:
            own.update(id(p) for p in b.parameters())
        if head >= 0:
            conv = self.decoder.output_conv2 if head == 1 else self.decoder.output_conv
            own.update(id(p) for p in conv.parameters())
        return [id(p) in own for p in self._ordered_params()]

    def _forward_range(self, x, first, last, head=-1):
        """Run layers [first, last) (+ head) on an NCHW tensor inside the plan of the whole network at the matching size."""
        if not x.is_cuda:
            raise _lib.LaneFitLibraryError("lanefit ERFNet needs its input on the MI355X; there is no CPU path")
        x = x.contiguous().float()
        N, C, h, w = x.shape
        s = _LAYER_IN_STRIDE[first]
        plan = self._plan(N, h * s, w * s)
        io = (ctypes.c_int * 6)()
        _lib.check(_lib.load().lf_erfnet_layer_io(plan.handle, first, io), "lf_erfnet_layer_io")
        if (C, h, w) != (io[0], io[1], io[2]):
            raise RuntimeError("ERFNet block expects an input of (N, %d, H, W), got %s" % (io[0], tuple(x.shape)))
        training = self._blocks()[first].training
        dropmask = self._make_dropmask(plan, x.device) if training else None
        y = _RangeFn.apply(self, plan, x, first, last, head, training, dropmask, *self._ordered_params())
        if training:
            bns = [m for b in self._blocks()[first:last] for m in b.modules() if isinstance(m, nn.BatchNorm2d)]
            torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
        return y

    # ---- bookkeeping -------------------------------------------------------------------
    def _ordered_params(self):
        ps = self.__dict__.get("_param_list")        # built once: nn.Parameter objects keep their identity through
        if ps is None:                                # .cuda() / .to() / load_state_dict (data is swapped in place)
            ps = [p for _, p in self.named_parameters()]
            self.__dict__["_param_list"] = ps
        return ps

    def _running_buffers(self):
        out = []
        for m in self._batchnorms():
            out += [m.running_mean, m.running_var]
        return out

    def _batchnorms(self):
        bns = self.__dict__.get("_bn_list")          # the module tree is fixed after construction
        if bns is None:
            bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            self.__dict__["_bn_list"] = bns
        return bns

    def _dropouts(self):
        return [m.dropout for m in self.modules() if isinstance(m, non_bottleneck_1d) and m.dropout.p != 0]

    def flat_grad(self):
        """The flat fp32 buffer whose views the last backward handed out as parameter gradients (or None)."""
        return self._flat_grad

    def _used_param_mask(self, head):
        names = [n for n, _ in self.named_parameters()]
        if head < 0:             # encoder only: no decoder parameter takes part (encoder.output_conv is a separate autograd node)
            return [n.startswith("encoder.") and not n.startswith("encoder.output_conv.") for n in names]
        unused_head = "decoder.output_conv." if head == 1 else "decoder.output_conv2."
        return [not (n.startswith("encoder.output_conv.") or n.startswith(unused_head)) for n in names]

    def _device_ptr_table(self, params):
        key = tuple(p.data_ptr() for p in params)
        if self._ptr_cache[0] != key:
            self._ptr_cache = (key, torch.tensor(key, dtype=torch.int64, device=params[0].device))
        return self._ptr_cache[1]

    def _plan(self, N, H, W):
        key = (N, H, W)
        if key not in self._plans:
            self._plans[key] = _Plan(N, H, W, self.in_channels, self.out_channels, 2 if self.pretrained else 1)
        return self._plans[key]

    def _make_dropmask(self, plan, device):
        """Dropout2d keep-masks, one (N,C) block per non_bottleneck_1d built with p > 0, drawn with torch's
        generator on the device: ONE uniform draw for all 13 blocks, thresholded by each block's current p
        (a block whose p was set to 0 keeps everything)."""
        if plan.n_drop == 0:
            return None
        built = [m for m in self.modules() if isinstance(m, non_bottleneck_1d) and m._built_with_dropout]
        ps = tuple(float(m.dropout.p) for m in built)
        cache = getattr(plan, "_drop_cache", None)
        if cache is None or cache[0] != ps or cache[1].device != device:
            N = plan.shape[0]
            pvec = torch.empty(plan.drop_floats, dtype=torch.float32)
            for p, off, ch in zip(ps, plan.drop_off, plan.drop_ch):
                pvec[off: off + N * ch] = p
            pvec = pvec.to(device)
            plan._drop_cache = cache = (ps, pvec, 1.0 / (1.0 - pvec), torch.zeros((), dtype=torch.float32, device=device))
        _, pvec, scale, zero = cache
        u = torch.rand(plan.drop_floats, dtype=torch.float32, device=device)
        return torch.where(u >= pvec, scale, zero)      # (three launches per step, not five; p = 1 gives 0, not 0 * inf)

    # ---- forward -----------------------------------------------------------------------
    def forward(self, input, flag, only_encode=False):
        if not input.is_cuda:
            raise _lib.LaneFitLibraryError("lanefit ERFNet needs its input on the MI355X; there is no CPU path")
        x = input.contiguous().float()
        N, C, H, W = x.shape
        assert C == self.in_channels, "expected %d input channels, got %d" % (self.in_channels, C)
        plan = self._plan(N, H, W)
        head = 0
        if self.pretrained and not flag:
            head = 1                                  # Decoder.forward: flag selects output_conv (ERFNet.py:134-141)
        dropmask = self._make_dropmask(plan, x.device) if self.training else None
        params = self._ordered_params()
        export = self.export_encoder_output
        if only_encode:
            if self.precision == "bf16":
                raise NotImplementedError("only_encode reads an fp32 encoder output: use precision 'fp32' or 'fp32x9'")
            self.export_encoder_output = True
            head = -1                                 # the engine stops after the encoder: decoder BN statistics untouched
        try:
            logits, enc = _BackboneFn.apply(self, plan, x, head, self.training, dropmask, *params)
        finally:
            self.export_encoder_output = export
        if only_encode:
            if self.training:
                torch._foreach_add_([m.num_batches_tracked for m in self.encoder.modules() if isinstance(m, nn.BatchNorm2d)], 1)
            return _PointwiseFn.apply(enc, self.encoder.output_conv.weight, self.encoder.output_conv.bias)
        if self.training:
            torch._foreach_add_([m.num_batches_tracked for m in self._batchnorms()], 1)
        if enc.dim() == 4:
            enc = enc.permute(0, 3, 1, 2)             # logical NCHW, channels-last memory
        if self.three_outputs:
            # BP/Networks/ERFNet.py:143-163: the decoder returns (output, output_seg) with output_seg = its INPUT (the encoder
            # output) unchanged, because do_segmentation is never set
            return enc, logits, (enc if enc.dim() == 4 else None)
        return enc, logits

