"""diagnostic (round 6): block-level train-mode test of encoder.layers.9 over several input seeds -- a ReLU tie or a real bug?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from oracle import erfnet_oracle
from test_blocks_gpu import _build, _oracle_block
from conftest import relerr
net, P = _build()
net.train(True)
table = erfnet_oracle.layer_table()
mods = [net.encoder.initial_block] + list(net.encoder.layers) + list(net.decoder.layers)
for li in (10, 12, 17):
    prefix, kind, cin, cout, _, d = table[li]
    for seed in range(100 + li, 100 + li + 6):
        N, h, w = 2, (8 if cin == 128 else 16), (16 if cin == 128 else 32)
        rng = np.random.default_rng(seed)
        x = torch.from_numpy(rng.random((N, cin, h, w), dtype=np.float32) + rng.standard_normal((N, cin, h, w)).astype(np.float32) * 0.5)
        xg = x.cuda().requires_grad_(True)
        y = mods[li](xg)
        gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        for p in net.parameters(): p.grad = None
        (y * gy.cuda()).sum().backward()
        Pd = erfnet_oracle.cast_params(P, torch.float64)
        xd = x.double().requires_grad_(True)
        taps = {}
        yd = erfnet_oracle._nb1d(xd, Pd, prefix, d, True, None, None, taps=taps)
        (yd * gy.double()).sum().backward()
        # how close to zero does the oracle's nearest pre-activation sit?
        near = min(float(t.detach().abs()[t.detach().abs() > 0].min()) for t in taps.values())
        print(prefix, "seed", seed, "out %.1e  d/d input %.1e   smallest nonzero |tap| in the oracle %.1e" % (relerr(y.detach().cpu(), yd.detach()), relerr(xg.grad.cpu(), xd.grad), near))
