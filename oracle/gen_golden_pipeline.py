"""Generate tests/golden/pipeline.npz with the REAL Pillow (the library the reference's loader calls through
torchvision.transforms.functional) on seeded synthetic frames:
    python -m oracle.gen_golden_pipeline
Replays the pixel statements of LaneDataset.__getitem__ (BEV/Dataloader/Load_Data_new.py:77-101,
BP/Dataloader/Load_Data_new.py:126-173) with PIL calls: F.crop -> Image.crop, F.resize -> Image.resize,
F.hflip -> Image.transpose(FLIP_LEFT_RIGHT), ToTensor -> uint8 HWC -> CHW float32 .div(255) (torch).
Full tensors for R = 256, checksums + strided samples for the other sizes.
"""
import os
import sys

import numpy as np
import PIL
import torch
from PIL import Image

from . import pipeline_oracle as po

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEEDS = (0, 1)
SIZES = (256, 320, 64, 512)


def loader_image(frame, R, flip):
    image = Image.fromarray(frame, mode="RGB")
    w, h = image.size
    image = image.crop((0, h - 640, w, h))
    image = image.resize((2 * R, R), Image.BILINEAR)
    if flip:
        image = image.transpose(Image.FLIP_LEFT_RIGHT)
    u8 = np.asarray(image)
    t = torch.from_numpy(u8.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)      # ToTensor
    return u8, t.numpy()


def loader_label(label, R, flip, tree, nclasses):
    gt = Image.fromarray(label, mode="P")
    w, h = gt.size
    gt = gt.crop((0, h - 640, w, h)).resize((2 * R, R), Image.NEAREST)
    gt = np.array(gt)
    idx3, idx4 = np.isin(gt, 3), np.isin(gt, 4)
    if tree == "bev" or nclasses < 3:
        gt[idx3] = 0
        gt[idx4] = 0
    if flip:
        gt = np.flip(gt, axis=1)
        idx1, idx2 = np.isin(gt, 1), np.isin(gt, 2)
        gt[idx1] = 2
        gt[idx2] = 1
        if tree == "bp":
            gt[idx3] = 4
            gt[idx4] = 3
    gt = Image.fromarray(np.ascontiguousarray(gt))
    t = torch.from_numpy(np.asarray(gt).copy())[None].to(torch.float32).div(255)                  # ToTensor
    return (t * 255).long().numpy()


def main():
    out = {"pillow_version": np.array(PIL.__version__)}
    for seed in SEEDS:
        frame, label = po.synthetic_frame(seed)
        for R in SIZES:
            for flip in (0, 1):
                u8, f32 = loader_image(frame, R, flip)
                tag = "s%d_R%d_f%d" % (seed, R, flip)
                out["img_sum_" + tag] = po.checksum(u8)
                out["img_sample_" + tag] = f32[:, ::8, ::8].copy()
                if R == 256 and seed == 0 and flip == 0:
                    out["img_u8_" + tag] = u8
                for tree, ncls in (("bev", 2), ("bp", 2), ("bp", 4)):
                    g = loader_label(label, R, flip, tree, ncls)
                    ltag = "%s%d_%s" % (tree, ncls, tag)
                    out["gt_sum_" + ltag] = po.checksum(g)
                    if R == 256:
                        out["gt_" + ltag] = g.astype(np.uint8)
    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    out["lut"] = (v * 255).long().numpy()
    path = os.path.join(OUT, "pipeline.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
