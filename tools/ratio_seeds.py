#!/usr/bin/env python
"""|hip - cpu64| / |cpu32 - cpu64| over seeds (the statistic of tests/test_baseline_configs_gpu.py::
test_bev_distance_ratio_over_seeds) for SEVERAL builds of the library on one box: the CPU legs are computed once
(--oracle, cached as .npz), then one subprocess per library evaluates the HIP path (--hip).

    python tools/ratio_seeds.py --oracle /tmp/ratio --seeds 6
    python tools/ratio_seeds.py --hip /tmp/ratio --seeds 6        (uses the in-tree liblanefit_hip.so)
"""
import argparse
import os
import sys
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import e2e_oracle, erfnet_oracle, inputs  # noqa: E402

N, R = 8, 256


def case(seed):
    P = erfnet_oracle.make_params(seed=40 + seed, out_channels=2)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=500 + seed))
    gt = inputs.bev_gt_params(N, seed=600 + seed)
    return P, x, gt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle")
    ap.add_argument("--hip")
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--precision", default="fp32")
    a = ap.parse_args()
    if a.oracle:
        os.makedirs(a.oracle, exist_ok=True)
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        for seed in range(a.seeds):
            P, x, gt = case(seed)
            o32, o64 = (e2e_oracle.bev_step(x, P, gt, dt, R) for dt in (torch.float32, torch.float64))
            np.savez(os.path.join(a.oracle, "seed%d.npz" % seed), b32=o32["beta"], b64=o64["beta"], l32=o32["logits"],
                     l64=o64["logits"], d32=o32["dlogits"], d64=o64["dlogits"])
        return
    from lanedetection_end2end_amd.bev.Loss_crit import Area_Loss
    from lanedetection_end2end_amd.bev.Networks.LSQ_layer import Net
    args = Namespace(batch_size=N, nclasses=2, resize=R, end_to_end=True, mod="erfnet", layers=18, channels_in=3,
                     pretrained=False, pool=True, activation_layer="square", no_cuda=False, order=2, reg_ls=0.0,
                     use_cholesky=False, mask_percentage=0.3, clas=False, no_mapping=False, loss_policy="area", weight_seg=30,
                     weight_funct="none")
    model = None
    ratios = {"beta": [], "beta_rms": [], "logits": [], "dlogits": []}
    rms = lambda u, v: float(np.sqrt(np.mean((np.asarray(u, dtype=np.float64) - v) ** 2)))
    for seed in range(a.seeds):
        P, x, gt = case(seed)
        o = np.load(os.path.join(a.hip, "seed%d.npz" % seed))
        if model is None:
            model = Net(args)
            model.net.load_state_dict(P)
            model = model.cuda()
            for m in model.modules():
                if isinstance(m, torch.nn.Dropout2d):
                    m.p = 0
            model.net.precision = a.precision
            model.train()
        else:
            model.net.load_state_dict(P)
        crit = Area_Loss(2, "none")
        gtc = torch.from_numpy(gt).cuda()
        model.zero_grad(set_to_none=True)
        b0, b1, _, _, _, _, output, _, _ = model(x.cuda(), True)
        output.retain_grad()
        (crit(b0, gtc[:, 0]) + crit(b1, gtc[:, 1])).backward()
        beta = torch.stack([b0, b1], 1)[..., 0].detach().cpu().numpy()
        ratios["beta"].append(np.abs(beta - o["b64"]).max() / max(np.abs(o["b32"] - o["b64"]).max(), 1e-30))
        ratios["beta_rms"].append(rms(beta, o["b64"]) / max(rms(o["b32"], o["b64"]), 1e-30))
        ratios["logits"].append(rms(output.detach().cpu().numpy(), o["l64"]) / rms(o["l32"], o["l64"]))
        ratios["dlogits"].append(rms(output.grad.cpu().numpy(), o["d64"]) / rms(o["d32"], o["d64"]))
    for k, v in ratios.items():
        v = np.array(v)
        print("|hip - cpu64| / |cpu32 - cpu64|  %-8s  per seed %s   median %.2f  max %.2f" % (k, np.round(v, 2), np.median(v), v.max()))


if __name__ == "__main__":
    main()
