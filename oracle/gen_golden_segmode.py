"""Golden vectors for the non-end-to-end ("segmentation") forward paths of the two LSQ_layer.Net classes, from the REAL reference
(authoring container only): arg-max of the logits -> per-lane maps valued k at class k -> row mask -> (BP only) the "prevent
singular matrix" overwrite of absent lanes with map [0,0] -> weighted least squares.
BEV/Networks/LSQ_layer.py:302-308,316,324-325; BP/Networks/LSQ_layer.py:279-293,298,308-314.

The backbone is replaced by a stub returning FIXED logits (the paths under test start at the logits; the backbone has its own
goldens), so the vectors are independent of fp32 noise in the network.  The fit runs in fp64 (the reference modules cast with
the same handful of ``.to(float64)`` calls gen_golden.py uses): in fp32 the reference's own bmm / inverse chain returns
coefficients that are rounding noise for these 0/k-valued maps in pixel coordinates (b = -1.5625 exactly, c off by 7 %:
cond(Z) ~ 1e9), which would pin nothing.  Writes tests/golden/segmode.npz.

    python -m oracle.gen_golden_segmode
"""
import os

import numpy as np
import torch

from . import inputs, ref_shims

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "segmode.npz")


def seg_logits(N, C, H, W, seed):
    """Logits whose arg-max paints lane-like bands: class k wins inside a slanted band, background elsewhere."""
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((N, C, H, W)).astype(np.float32) * 0.1
    ys = np.arange(H)[:, None] / H
    xs = np.arange(W)[None, :] / W
    for n in range(N):
        for k in range(1, C):
            centre = 0.15 + 0.7 * k / C + 0.25 * (ys - 0.6) * (1 if k % 2 else -1) + 0.02 * n
            z[n, k] += 3.0 * (np.abs(xs - centre) < 0.03)
    return z


class _Stub(torch.nn.Module):
    def __init__(self, logits, three):
        super().__init__()
        self.logits, self.three = logits, three

    def forward(self, x, flag):
        return (None, self.logits, None) if self.three else (None, self.logits)


def main():
    assert ref_shims.available(), "needs /root/reference"
    out = {}
    N, R = 2, 64
    x = torch.zeros(N, 3, R, 2 * R)
    # ---- BEV, 2 lanes
    ref = ref_shims.load("bev")
    model = ref.LSQ_layer.Net(ref_shims.default_args("bev", batch_size=N, resize=R, end_to_end=False))
    z = torch.from_numpy(seg_logits(N, 3, R, 2 * R, seed=21))
    model.net = _Stub(z.double(), False)
    model.M = model.M.double()
    model.project_layer.base_grid = model.project_layer.base_grid.double()
    model.ls_layer.tensor_ones = model.ls_layer.tensor_ones.double()
    model.ls_layer.reg_ls = model.ls_layer.reg_ls.double()
    b0, b1, b2, b3, masked, M, output, line, horizon = model(x, False)
    assert b2 is None and b3 is None
    out["bev_logits"] = z.numpy()
    out["bev_beta"] = np.stack([b0.detach().numpy(), b1.detach().numpy()], 1)[..., 0]
    out["bev_masked"] = masked.detach().numpy()
    # ---- BP, 4 lanes, lanes (0,2) and (1,3) flagged absent
    ref = ref_shims.load("bp")
    model = ref.LSQ_layer.Net(ref_shims.default_args("bp", batch_size=N, resize=R, nclasses=4, end_to_end=False, mask_percentage=0.2))
    z = torch.from_numpy(seg_logits(N, 5, R, 2 * R, seed=22))
    model.net = _Stub(z.double(), True)
    model.grid = model.grid.double()
    model.ls_layer.tensor_ones = model.ls_layer.tensor_ones.double()
    model.ls_layer.reg_ls = model.ls_layer.reg_ls.double()
    # (an integer gt_line: with torch >= 2 `repeat(gt_line.sum().item(), 1, 1)` rejects the float count a float tensor yields)
    gt_line = torch.zeros(N, 4, dtype=torch.long)
    gt_line[0, 2] = 1
    gt_line[1, 3] = 1
    res = model(x, gt_line, False)
    out["bp_logits"] = z.numpy()
    out["bp_gt_line"] = gt_line.numpy()
    out["bp_beta"] = np.stack([b.detach().numpy() for b in res[:4]], 1)[..., 0]
    out["bp_masked"] = res[4].detach().numpy()
    # no lane flagged: the overwrite is skipped
    res = model(x, torch.zeros(N, 4, dtype=torch.long), False)
    out["bp_beta_noflag"] = np.stack([b.detach().numpy() for b in res[:4]], 1)[..., 0]
    np.savez_compressed(OUT, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
