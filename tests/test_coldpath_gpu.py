"""The cold-path kernels of round 4 (csrc/lf_linear.hip) against torch's own ops in fp64: the nn.Linear tails of the --clas
heads (lf_linear_fwd / lf_linear_bwd, with and without the fused ReLU, vector and scalar K) and the segmentation-mode fit input
(lf_seg_maps) against the reference's statement sequence (BEV/Networks/LSQ_layer.py:302-308,316; BP :279-293,298,308-311).
Their use inside the heads / Net is covered by tests/test_clas_gpu.py and
tests/test_backbone_gpu.py::test_segmentation_mode_fit_vs_reference_goldens (goldens from the real reference)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,K,O,relu", [(5, 128, 4, False), (32, 2048, 256, False), (9, 32768, 128, True), (3, 10, 3, True),
                                        (64, 128, 3, False)])
def test_linear_matches_torch(N, K, O, relu):
    from lanedetection_end2end_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + K)
    x = torch.randn(N, K, generator=g)
    w = torch.randn(O, K, generator=g) / K ** 0.5
    b = torch.randn(O, generator=g)
    gy = torch.randn(N, O, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    yd = F.linear(xd, wd, bd)
    if relu:
        yd = F.relu(yd)
    (yd * gy.double()).sum().backward()
    xc, wc, bc = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xc, wc, bc, relu=relu)
    (y * gy.cuda()).sum().backward()
    rel = lambda a, r: float((a.detach().cpu().double() - r.detach()).abs().max() / r.detach().abs().max().clamp_min(1e-30))
    errs = (rel(y, yd), rel(xc.grad, xd.grad), rel(wc.grad, wd.grad), rel(bc.grad, bd.grad))
    print("linear N=%d K=%d O=%d relu=%s: y %.1e gx %.1e gw %.1e gb %.1e" % ((N, K, O, relu) + errs))
    assert max(errs) < 2e-6
    # no bias, no input gradient
    y2 = ops.linear(x.cuda(), wc, None, relu=relu)
    assert rel(y2, F.relu(F.linear(x.double(), w.double())) if relu else F.linear(x.double(), w.double())) < 2e-6
    # deterministic
    assert torch.equal(ops.linear(xc, wc, bc, relu=relu), y)


# (2, 2, ..) / (4, 4, ..): C == L -- a net built end_to_end=True (out_channels == lanes, not pretrained) called with end_to_end=False:
# the reference returns maps whose last lane is all zero (lf_seg_maps used to refuse L >= C)
@pytest.mark.parametrize("C,L,zero_rows,flagged", [(3, 2, 13, False), (5, 4, 20, True), (5, 4, 0, True), (3, 2, 64, False), (2, 2, 13, False), (4, 4, 20, True)])
def test_seg_maps_matches_reference_statements(C, L, zero_rows, flagged):
    from lanedetection_end2end_amd import ops
    N, H, W = 3, 64, 96
    g = torch.Generator().manual_seed(C * 10 + L)
    logits = torch.randn(N, C, H, W, generator=g)
    logits[0, :, 40, :7] = 1.5                       # ties: torch.max takes the first maximum
    gt_line = torch.zeros(N, L)
    if flagged:
        gt_line[1, 2] = 1
        gt_line[0, 0] = 1                            # map [0,0] flagged itself
        gt_line[2, L - 1] = 1
    # the reference's statements
    _, act = torch.max(logits, 1)
    act = act.float()
    ref = torch.stack([act * (act == k).float() for k in range(1, L + 1)], 1)
    ref = ref.index_fill(2, torch.arange(zero_rows), 0)
    if gt_line.sum() != 0:
        mask = gt_line[:, :, None, None].bool().expand_as(ref)
        ref[mask] = ref[0, 0].unsqueeze(0).repeat(int(gt_line.sum().item()), 1, 1).view(-1)
    got = ops.seg_maps(logits.cuda(), gt_line.cuda() if flagged else None, zero_rows, L)
    assert torch.equal(got.cpu(), ref)
    if not flagged:
        assert torch.equal(ops.seg_maps(logits.cuda(), gt_line.cuda(), zero_rows, L).cpu(), ref)
        # a flag tensor that cannot be expanded is ignored while it is all zeros (the reference never reaches expand_as) ...
        assert torch.equal(ops.seg_maps(logits.cuda(), torch.zeros(N, L + 2).cuda(), zero_rows, L).cpu(), ref)
        bad = torch.zeros(N, L + 2)
        bad[0, 0] = 1
        with pytest.raises(RuntimeError):            # ... and fails like expand_as once it is not
            ops.seg_maps(logits.cuda(), bad.cuda(), zero_rows, L)
