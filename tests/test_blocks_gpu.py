"""Block-level surface (Missing #5): the reference's sub-modules are callable on their own --
DownsamplerBlock / non_bottleneck_1d / UpsamplerBlock.forward(input), Encoder.forward(input, predict), Decoder.forward(input,
flag) (BEV/Networks/ERFNet.py:19-22,44-60,86-95,104-107,129-142) -- each running its layer range of the engine plan
(lf_erfnet_forward_range / lf_erfnet_backward_range).  Values and gradients (input and parameters) against the fp64 oracle's
block functions; Encoder and Decoder composed == the whole-network pass."""
import numpy as np
import pytest
import torch

from conftest import relerr
from oracle import erfnet_oracle, inputs

pytestmark = pytest.mark.gpu


def _build(out_channels=2, pretrained=False, three=False):
    from lanedetection_end2end_amd import erfnet
    cls = erfnet.Net
    if three:
        class cls(erfnet.Net):
            three_outputs = True
    net = cls(in_channels=3, out_channels=out_channels, pretrained=pretrained)
    P = erfnet_oracle.make_params(seed=9, out_channels=out_channels, pretrained=pretrained)
    net.load_state_dict(P)
    net = net.cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0
    return net, P


def _oracle_block(kind, x, Pd, prefix, d, training):
    if kind == "down":
        return erfnet_oracle._down(x, Pd, prefix, training, None)
    if kind == "up":
        return erfnet_oracle._up(x, Pd, prefix, training, None)
    return erfnet_oracle._nb1d(x, Pd, prefix, d, training, None, None)


@pytest.mark.parametrize("training", [True, False])
def test_every_block_kind_against_the_oracle(training):
    net, P = _build()
    net.train(training)
    table = erfnet_oracle.layer_table()
    # one block of every kind and channel count: stem, down 16->64, nb 64, down 64->128, nb 128 (dilated), up, nb 64 (dec), up, nb 16
    picks = [0, 1, 3, 7, 10, 16, 17, 19, 21]
    mods = [net.encoder.initial_block] + list(net.encoder.layers) + list(net.decoder.layers)
    stride = [1, 2] + [4] * 5 + [4] + [8] * 8 + [8, 4, 4, 4, 2, 2]
    N, H, W = 2, 64, 128
    worst = 0.0
    for li in picks:
        prefix, kind, cin, cout, _, d = table[li]
        cin = 3 if cin is None else cin
        h, w = H // stride[li], W // stride[li]
        rng = np.random.default_rng(100 + li)
        x = torch.from_numpy(rng.random((N, cin, h, w), dtype=np.float32) + (0 if li == 0 else rng.standard_normal((N, cin, h, w)).astype(np.float32) * 0.5))
        xg = x.cuda().requires_grad_(li > 0)
        y = mods[li](xg)
        gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        for p in net.parameters():
            p.grad = None
        (y * gy.cuda()).sum().backward()
        Pd = erfnet_oracle.cast_params(P, torch.float64)
        own = [k for k in Pd if k.startswith(prefix + ".") and Pd[k].is_floating_point() and "running" not in k]
        for k in own:
            Pd[k].requires_grad_(True)
        xd = x.double().requires_grad_(True)
        yd = _oracle_block(kind, xd, Pd, prefix, d, training)
        (yd * gy.double()).sum().backward()
        e_y = relerr(y.detach().cpu(), yd.detach())
        e_x = relerr(xg.grad.cpu(), xd.grad) if li > 0 else 0.0
        named = dict(net.named_parameters())
        gmax = max(float(Pd[k].grad.abs().max()) for k in own)
        e_p = 0.0
        for k in own:
            scale = float(Pd[k].grad.abs().max())
            if scale < 1e-6 * gmax:
                continue                      # conv biases in front of a train-mode BatchNorm: analytically zero
            e_p = max(e_p, float((named[k].grad.cpu().double() - Pd[k].grad).abs().max()) / scale)
        others = [k for k, p in named.items() if not k.startswith(prefix + ".") and p.grad is not None]
        print("%-26s %-5s train=%d: out %.1e  d/d input %.1e  parameter gradients %.1e" % (prefix, kind, training, e_y, e_x, e_p))
        assert not others, others             # only the block's own parameters receive gradients
        assert e_y < 1e-5 and e_x < 2e-4 and e_p < 2e-4, (prefix, e_y, e_x, e_p)
        worst = max(worst, e_y, e_x, e_p)
    print("worst %.1e" % worst)


def test_encoder_and_decoder_compose_to_the_network():
    net, P = _build(pretrained=True)
    N, H, W = 2, 64, 128
    x = torch.from_numpy(inputs.images(N, H, W, seed=51)).cuda()
    gy = torch.from_numpy(np.random.default_rng(5).standard_normal((N, 2, H, W)).astype(np.float32)).cuda()
    for training in (False, True):
        net.train(training)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        enc_f, dec_f = net(x, True)
        for p in net.parameters():
            p.grad = None
        (dec_f * gy).sum().backward()
        g_full = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        sd1 = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_state_dict(sd0)                       # same running statistics for the block-wise pass
        enc = net.encoder(x)
        dec = net.decoder(enc, True)
        assert torch.equal(enc, enc_f.contiguous()) and torch.equal(dec, dec_f)
        for p in net.parameters():
            p.grad = None
        (dec * gy).sum().backward()
        worst = 0.0
        gmax = max(float(v.abs().max()) for v in g_full.values())
        for k, p in net.named_parameters():
            if k in g_full:
                scale = float(g_full[k].abs().max())
                if training and k.endswith((".conv.bias", "conv1x3_1.bias", "conv1x3_2.bias")):
                    continue                  # biases in front of a train-mode BatchNorm: analytically zero, both are rounding noise
                if scale < 1e-6 * gmax:
                    continue
                e = float((p.grad - g_full[k]).abs().max()) / scale
                if e > 1e-4:
                    print("   MISMATCH %s: %.2e (scale %.2e)" % (k, e, scale))
                worst = max(worst, e)
            else:
                assert p.grad is None, k
        print("train=%d: encoder(x) / decoder(enc) == net(x) bit for bit; parameter gradients agree to %.1e" % (training, worst))
        assert worst < 2e-5
        for k, v in net.state_dict().items():         # the same running statistics were written
            assert torch.equal(v, sd1[k]), k
    # the other head, the predict branch, and the BP tree's two-tuple
    net.eval()
    with torch.no_grad():
        enc = net.encoder(x)
        d2 = net.decoder(enc, False)
        assert d2.shape == (N, 3, H, W) and torch.equal(d2, net(x, False)[1])
        pred = net.encoder(x, predict=True)
        assert torch.equal(pred, net(x, True, only_encode=True))
    bp, _ = _build(three=True)
    bp.eval()
    with torch.no_grad():
        out, seg = bp.decoder(bp.encoder(x), True)
        assert torch.equal(out, bp(x, True)[1]) and seg.shape == (N, 128, H // 8, W // 8)
    with pytest.raises(RuntimeError):
        net.decoder.layers[1](torch.zeros(N, 16, 8, 8, device="cuda"))      # wrong channel count for that block
