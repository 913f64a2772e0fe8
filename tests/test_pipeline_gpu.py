"""GPU: the on-device input pipeline (lf_pipeline_image / lf_pipeline_label) vs the oracle and vs golden vectors
from the real Pillow -- bit-exact (uint8 / int64 / the fp32 bits of v/255)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import pipeline_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_pipe():
    return np.load(os.path.join(GOLDEN, "pipeline.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def frames():
    return {s: po.synthetic_frame(s) for s in (0, 1)}


@pytest.mark.parametrize("R", [256, 320, 64, 512])
@pytest.mark.parametrize("tree,ncls", [("bev", 2), ("bp", 2), ("bp", 4)])
def test_pipeline_vs_pillow_golden(golden_pipe, frames, R, tree, ncls):
    from lanedetection_end2end_amd.pipeline import InputPipeline
    pipe = InputPipeline(R, tree=tree, nclasses=ncls)
    fr = torch.from_numpy(np.stack([frames[0][0], frames[1][0], frames[0][0], frames[1][0]])).cuda()
    lb = torch.from_numpy(np.stack([frames[0][1], frames[1][1], frames[0][1], frames[1][1]])).cuda()
    flip = torch.tensor([False, False, True, True])
    image, gt, horizon = pipe(fr, lb, flip)
    assert image.shape == (4, 3, R, 2 * R) and image.dtype == torch.float32 and gt.shape == (4, 1, R, 2 * R) and gt.dtype == torch.int64
    image, gt = image.cpu().numpy(), gt.cpu().numpy()
    for i, (seed, fl) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        tag = "s%d_R%d_f%d" % (seed, R, fl)
        u8 = np.rint(image[i] * 255).astype(np.uint8).transpose(1, 2, 0)
        assert (po.checksum(u8) == golden_pipe["img_sum_" + tag]).all(), tag
        assert np.array_equal(image[i][:, ::8, ::8], golden_pipe["img_sample_" + tag]), tag
        assert np.array_equal(image[i], po.preprocess_image(frames[seed][0], R, fl)), tag     # full tensor, fp32 bits
        ltag = "%s%d_%s" % (tree, ncls, tag)
        assert (po.checksum(gt[i]) == golden_pipe["gt_sum_" + ltag]).all(), ltag
        assert np.array_equal(gt[i], po.preprocess_label(frames[seed][1], R, fl, tree, ncls)), ltag
        if tree == "bev":
            assert np.array_equal(horizon[i].cpu().numpy(), po.bev_horizon(gt[i]))
    assert (horizon is None) == (tree == "bp")


def test_pipeline_other_frame_sizes_and_edge_cases():
    """Non-TuSimple frame sizes (odd widths, up-scaling), an unlabelled image (horizon = all ones), no labels."""
    from lanedetection_end2end_amd.pipeline import InputPipeline
    rng = np.random.default_rng(5)
    for (H, W, crop, R) in ((97, 333, 64, 16), (50, 120, 50, 40), (720, 1280, 640, 48)):
        pipe = InputPipeline(R, frame_hw=(H, W), crop=crop)
        fr = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
        lb = rng.integers(0, 5, (3, H, W), dtype=np.uint8)
        lb[1] = 0
        image, gt, horizon = pipe(torch.from_numpy(fr).cuda(), torch.from_numpy(lb).cuda(), torch.tensor([True, False, False]))
        for i, fl in enumerate((1, 0, 0)):
            ref = po.resize_bilinear_u8(fr[i][H - crop:], R, 2 * R)
            ref = ref[:, ::-1] if fl else ref
            assert np.array_equal(image[i].cpu().numpy(), ref.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
            g = po.resize_nearest_u8(lb[i][H - crop:], R, 2 * R).copy()
            g[(g == 3) | (g == 4)] = 0
            if fl:
                g = g[:, ::-1].copy()
                one, two = g == 1, g == 2
                g[one], g[two] = 2, 1
            assert np.array_equal(gt[i, 0].cpu().numpy(), g)
        assert horizon[1].min().item() == 1.0
        image2, gt2, hz2 = pipe(torch.from_numpy(fr).cuda())
        assert gt2 is None and hz2 is None and torch.equal(image2[1:], image[1:])


def test_pipeline_feeds_the_network():
    """uint8 frames -> pipeline -> BEV Net forward/backward: the device-resident path end to end."""
    from argparse import Namespace
    from lanedetection_end2end_amd.bev.Networks.LSQ_layer import Net
    from lanedetection_end2end_amd.pipeline import InputPipeline
    R, N = 64, 2
    args = Namespace(batch_size=N, nclasses=2, resize=R, end_to_end=True, mod="erfnet", layers=18, channels_in=3,
                     pretrained=False, pool=True, activation_layer="square", no_cuda=False, order=2, reg_ls=1.0,
                     use_cholesky=False, mask_percentage=0.3, clas=False, no_mapping=False, loss_policy="area",
                     weight_seg=30, weight_funct="none")
    torch.manual_seed(0)
    model = Net(args).cuda().train()
    fr = torch.from_numpy(np.stack([po.synthetic_frame(s)[0] for s in (0, 1)])).cuda()
    image, _, _ = InputPipeline(R)(fr)
    out = model(image, True)
    (out[0].sum() + out[1].sum()).backward()
    assert torch.isfinite(out[0]).all() and torch.isfinite(model.net.encoder.initial_block.conv.weight.grad).all()


def test_pipeline_indexed_pool_equals_gather_then_pipeline(frames):
    """``index``: the batch is gathered from a resident pool INSIDE the kernels (lf_pipeline_image_indexed /
    lf_pipeline_label_indexed) -- bit-identical to gathering the frames first (what a cached dataset's SubsetRandomSampler batch
    needs, BEV/Dataloader/Load_Data_new.py:305-320), repeated and out-of-order indices included."""
    from lanedetection_end2end_amd.pipeline import InputPipeline
    rng = np.random.default_rng(11)
    pool_f = torch.from_numpy(np.stack([frames[0][0], frames[1][0]] + [rng.integers(0, 256, frames[0][0].shape, dtype=np.uint8) for _ in range(3)])).cuda()
    pool_l = torch.from_numpy(np.stack([frames[0][1], frames[1][1]] + [rng.integers(0, 5, frames[0][1].shape, dtype=np.uint8) for _ in range(3)])).cuda()
    sel = torch.tensor([4, 0, 0, 3, 1, 2], device="cuda")
    flip = torch.tensor([True, False, True, False, False, True])
    for tree, ncls, R in (("bev", 2, 64), ("bp", 4, 256)):
        pipe = InputPipeline(R, tree=tree, nclasses=ncls)
        a = pipe(pool_f, pool_l, flip, index=sel)
        b = pipe(pool_f.index_select(0, sel), pool_l.index_select(0, sel), flip)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert (a[2] is None and b[2] is None) or torch.equal(a[2], b[2])
        assert a[0].shape == (6, 3, R, 2 * R)


def test_pipeline_index_outside_the_pool_raises(frames):
    """An index outside the resident pool -- stale, negative, >= pool -- must not read foreign memory: the kernels count it and read
    pool entry 0 instead, the host raises IndexError (as the reference's tensor indexing does) at the next call or at flush(),
    without a host sync in the step itself."""
    from lanedetection_end2end_amd.pipeline import InputPipeline
    pool_f = torch.from_numpy(np.stack([frames[0][0], frames[1][0]])).cuda()
    pool_l = torch.from_numpy(np.stack([frames[0][1], frames[1][1]])).cuda()
    pipe = InputPipeline(64, tree="bev", nclasses=2)
    good = pipe(pool_f, pool_l, None, index=torch.tensor([1, 0], device="cuda"))
    pipe.flush()                                                     # nothing pending
    out = pipe(pool_f, pool_l, None, index=torch.tensor([1, 2, -1, 0], device="cuda"))
    torch.cuda.synchronize()
    assert torch.isfinite(out[0]).all()
    assert torch.equal(out[0][1], good[0][1]) and torch.equal(out[0][2], good[0][1])      # the bad entries read pool entry 0
    assert torch.equal(out[0][0], good[0][0]) and torch.equal(out[1][3], good[1][1])
    with pytest.raises(IndexError, match="2 index value"):
        pipe(pool_f, pool_l, None, index=torch.tensor([0, 1], device="cuda"))
    ok = pipe(pool_f, pool_l, None, index=torch.tensor([0, 1], device="cuda"))                  # the counter was reset
    pipe.flush()
    assert torch.equal(ok[0][0], good[0][1])
