"""Functional CPU restatement of the reference ERFNet backbone (CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The reference backbone is plain ``torch.nn`` modules; this file restates its data
flow with ``torch.nn.functional`` calls on an explicit parameter dict so that it can
run in fp32 *and* fp64 on the GPU box (where /root/reference does not exist), with
externally supplied Dropout2d keep-masks, and so that autograd on it yields the
reference gradients.  ``tests/test_oracle_golden.py`` pins it against outputs of the
real reference modules (``oracle/gen_golden.py``).

Reference: BEV/Networks/ERFNet.py (= /root/reference/Birds_Eye_View_Loss/Networks/ERFNet.py);
BP/Networks/ERFNet.py differs only in the 3-tuple return (:170-176).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # ERFNet.py:17,33,39,102
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default


def layer_table():
    """[(prefix, kind, cin, cout, dropprob, dilation)] in module order.

    Encoder: ERFNet.py:63-84; Decoder: ERFNet.py:109-126.
    kind in {'down', 'nb1d', 'up'}.
    """
    t = [("encoder.initial_block", "down", None, 16, 0.0, 1),
         ("encoder.layers.0", "down", 16, 64, 0.0, 1)]
    for i in range(5):
        t.append(("encoder.layers.%d" % (1 + i), "nb1d", 64, 64, 0.03, 1))
    t.append(("encoder.layers.6", "down", 64, 128, 0.0, 1))
    for i, d in enumerate([2, 4, 8, 16, 2, 4, 8, 16]):
        t.append(("encoder.layers.%d" % (7 + i), "nb1d", 128, 128, 0.3, d))
    t += [("decoder.layers.0", "up", 128, 64, 0.0, 1),
          ("decoder.layers.1", "nb1d", 64, 64, 0.0, 1),
          ("decoder.layers.2", "nb1d", 64, 64, 0.0, 1),
          ("decoder.layers.3", "up", 64, 16, 0.0, 1),
          ("decoder.layers.4", "nb1d", 16, 16, 0.0, 1),
          ("decoder.layers.5", "nb1d", 16, 16, 0.0, 1)]
    return t


def param_spec(in_channels=3, out_channels=2, pretrained=False):
    """OrderedDict key -> shape, in the reference's ``state_dict()`` order.

    Includes BN buffers (running_mean, running_var, num_batches_tracked) and the
    never-trained ``encoder.output_conv`` (ERFNet.py:84).
    """
    spec = OrderedDict()

    def conv(p, co, ci, kh, kw):
        spec[p + ".weight"] = (co, ci, kh, kw)
        spec[p + ".bias"] = (co,)

    def bn(p, c):
        spec[p + ".weight"] = (c,)
        spec[p + ".bias"] = (c,)
        spec[p + ".running_mean"] = (c,)
        spec[p + ".running_var"] = (c,)
        spec[p + ".num_batches_tracked"] = ()

    enc_done = False
    for prefix, kind, cin, cout, _, _ in layer_table():
        if prefix.startswith("decoder") and not enc_done:
            conv("encoder.output_conv", out_channels, 128, 1, 1)
            enc_done = True
        if kind == "down":
            cin = in_channels if cin is None else cin
            conv(prefix + ".conv", cout - cin, cin, 3, 3)
            bn(prefix + ".bn", cout)
        elif kind == "nb1d":
            conv(prefix + ".conv3x1_1", cout, cout, 3, 1)
            conv(prefix + ".conv1x3_1", cout, cout, 1, 3)
            bn(prefix + ".bn1", cout)
            conv(prefix + ".conv3x1_2", cout, cout, 3, 1)
            conv(prefix + ".conv1x3_2", cout, cout, 1, 3)
            bn(prefix + ".bn2", cout)
        else:  # ConvTranspose2d weight is (Cin, Cout, kh, kw)
            spec[prefix + ".conv.weight"] = (cin, cout, 3, 3)
            spec[prefix + ".conv.bias"] = (cout,)
            bn(prefix + ".bn", cout)
    spec["decoder.output_conv.weight"] = (16, out_channels, 2, 2)
    spec["decoder.output_conv.bias"] = (out_channels,)
    if pretrained:
        spec["decoder.output_conv2.weight"] = (16, out_channels + 1, 2, 2)
        spec["decoder.output_conv2.bias"] = (out_channels + 1,)
    return spec


def make_params(seed=0, in_channels=3, out_channels=2, pretrained=False, bias_scale=0.05,
                dtype=torch.float32):
    """Deterministic synthetic parameters, independent of torch's module-init RNG order.

    Conv weights ~ N(0, 2/fan_in) (the kaiming rule the reference applies,
    BEV/Networks/utils.py:490-503), BN gamma ~ N(1, 0.02); biases and BN beta get small
    non-zero values (the reference zeroes them) so that bias handling is exercised.
    One ``torch.Generator`` per key (seed + index) => stable under spec edits.
    """
    out = OrderedDict()
    for idx, (k, shp) in enumerate(param_spec(in_channels, out_channels, pretrained).items()):
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = torch.zeros(shp, dtype=dtype)
        elif k.endswith("running_var"):
            out[k] = torch.ones(shp, dtype=dtype)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]   # torch's fan_in rule, also for ConvTranspose2d
            out[k] = (torch.randn(shp, generator=g, dtype=torch.float64) * (2.0 / fan_in) ** 0.5).to(dtype)
        elif ".bn" in k and k.endswith(".weight"):
            out[k] = (1.0 + 0.02 * torch.randn(shp, generator=g, dtype=torch.float64)).to(dtype)
        else:
            out[k] = (bias_scale * torch.randn(shp, generator=g, dtype=torch.float64)).to(dtype)
    return out


def _tap(taps, override, key, t):
    """Record tensor ``t`` under ``key``; with ``override`` replace its VALUE by override[key] while
    keeping the autograd graph (straight-through): lets a test evaluate the fp64 backward at exactly
    the forward state another implementation produced (identical ReLU masks / saved tensors)."""
    if override is not None and key in override:
        t = t + (override[key].to(t.dtype) - t).detach()
    if taps is not None:
        taps[key] = t
    return t


def _relu(z, override, key, bn_input_key=None):
    """ReLU.  In a straight-through evaluation (``override``) the DERIVATIVE mask is the one the other implementation's forward
    pass used, not sign(z) of this evaluation: an element whose pre-activation lies inside fp32 rounding of zero may be
    positive there and non-positive here, and that single element then carries its whole gradient into every parameter sum
    below it (measured: 1e-4..6e-4 of a decoder tensor's maximum where the arithmetic difference is 2e-6).
      * output saved under ``key`` (post-ReLU tensor): mask = saved > 0;
      * ``relu(bn1(t2))`` of non_bottleneck_1d is never stored: override[key] = (scale, shift), the folded fp32 vectors of that
        BatchNorm, and the mask is sign(fma(t2, scale, shift)) of the saved t2 = override[bn_input_key] -- exact in fp64 (the
        product of two fp32 numbers is exact in fp64 and the sign of a two-term fp64 sum is exact)."""
    if override is None or key not in override:
        return F.relu(z)
    if bn_input_key is None:
        mask = override[key] > 0
    else:
        sc, sh = override[key]
        mask = (override[bn_input_key].double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]) > 0
    return z * mask.to(z.dtype)


def _bn(x, P, prefix, training, stats_out):
    """nn.BatchNorm2d(eps=1e-3): batch stats in train mode, running stats in eval."""
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = x.numel() / x.shape[1]
            with torch.no_grad():
                rm = P[prefix + ".running_mean"].to(x.dtype)
                rv = P[prefix + ".running_var"].to(x.dtype)
                stats_out[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
                stats_out[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv \
                    + BN_MOMENTUM * var.detach() * n / (n - 1)
    else:
        mean = P[prefix + ".running_mean"].to(x.dtype)
        var = P[prefix + ".running_var"].to(x.dtype)
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def _down(x, P, p, training, stats_out, taps=None, override=None):
    """DownsamplerBlock.forward -- ERFNet.py:19-22."""
    y = torch.cat([F.conv2d(x, P[p + ".conv.weight"], P[p + ".conv.bias"], stride=2, padding=1),
                   F.max_pool2d(x, 2, stride=2)], 1)
    y = _tap(taps, override, p + "#0", y)
    return _relu(_bn(y, P, p + ".bn", training, stats_out), override, p)


def _nb1d(x, P, p, d, training, stats_out, keep, taps=None, override=None):
    """non_bottleneck_1d.forward -- ERFNet.py:44-60.  ``keep`` = (N,C) scaled keep-mask or None."""
    t1 = _relu(F.conv2d(x, P[p + ".conv3x1_1.weight"], P[p + ".conv3x1_1.bias"], padding=(1, 0)), override, p + "#0")
    t1 = _tap(taps, override, p + "#0", t1)
    t2 = F.conv2d(t1, P[p + ".conv1x3_1.weight"], P[p + ".conv1x3_1.bias"], padding=(0, 1))
    t2 = _tap(taps, override, p + "#1", t2)
    y = _relu(_bn(t2, P, p + ".bn1", training, stats_out), override, p + "#bn1", p + "#1")
    t3 = _relu(F.conv2d(y, P[p + ".conv3x1_2.weight"], P[p + ".conv3x1_2.bias"],
                        padding=(d, 0), dilation=(d, 1)), override, p + "#2")
    t3 = _tap(taps, override, p + "#2", t3)
    t4 = F.conv2d(t3, P[p + ".conv1x3_2.weight"], P[p + ".conv1x3_2.bias"], padding=(0, d), dilation=(1, d))
    t4 = _tap(taps, override, p + "#3", t4)
    y = _bn(t4, P, p + ".bn2", training, stats_out)
    if keep is not None:
        y = y * keep[:, :, None, None].to(y.dtype)
    return _relu(y + x, override, p)


def _up(x, P, p, training, stats_out, taps=None, override=None):
    """UpsamplerBlock.forward -- ERFNet.py:104-107."""
    y = F.conv_transpose2d(x, P[p + ".conv.weight"], P[p + ".conv.bias"], stride=2, padding=1,
                           output_padding=1)
    y = _tap(taps, override, p + "#0", y)
    return _relu(_bn(y, P, p + ".bn", training, stats_out), override, p)


def erfnet_forward(x, P, training=True, keep_masks=None, head="output_conv", stats_out=None,
                   taps=None, override=None):
    """Net.forward(input, flag) -- ERFNet.py:151-157 -> (encoder_output, decoder_output).

    ``keep_masks``: dict prefix -> (N,C) tensor holding 0 or 1/(1-p) per (sample, channel),
    i.e. what nn.Dropout2d draws (ERFNet.py:41,57-58); None => dropout disabled.
    ``head``: 'output_conv' or 'output_conv2' (Decoder.forward flag, :134-141).
    ``taps``: optional dict filled with every block output (key = prefix) and the tensors inside the
    block (key = prefix#slot: down/up pre-BN = #0; nb1d t1..t4 = #0..#3) for per-layer parity tests.
    ``override``: dict with the same keys -> values substituted straight-through (see ``_tap``), the ReLU derivative masks taken
    from the same state (see ``_relu``; optional key prefix#bn1 -> (scale, shift) of a non_bottleneck_1d's first BatchNorm).
    """
    enc = None
    y = x
    for prefix, kind, _, _, _, d in layer_table():
        if prefix == "decoder.layers.0":
            enc = y
        if kind == "down":
            y = _down(y, P, prefix, training, stats_out, taps, override)
        elif kind == "nb1d":
            keep = None if keep_masks is None else keep_masks.get(prefix)
            y = _nb1d(y, P, prefix, d, training, stats_out, keep, taps, override)
        else:
            y = _up(y, P, prefix, training, stats_out, taps, override)
        y = _tap(taps, override, prefix, y)
    dec = F.conv_transpose2d(y, P["decoder.%s.weight" % head], P["decoder.%s.bias" % head], stride=2)
    return enc, dec


def cast_params(P, dtype):
    return OrderedDict((k, v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in P.items())
