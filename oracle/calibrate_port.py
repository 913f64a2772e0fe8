"""Calibration of bench.py's cpu_baseline against the REAL reference: times, on the same inputs and thread count, in the
authoring container (the GPU box has no /root/reference),

  reference   the unmodified BEV LSQ_layer.Net + Area_Loss from /root/reference (through oracle/ref_shims.py)
  aten        oracle/vendor_baseline.bev_step on device "cpu": the reference's exact ATen call sequence (F.conv2d / F.batch_norm /
              F.dropout2d / bmm + inverse; the same oneDNN / LAPACK kernels the reference's nn.Modules dispatch to) -- what
              bench.py times as cpu_baseline on the GPU box
  port        oracle/e2e_oracle.bev_step (fp32 backbone, fp64 numpy fit + loss with analytic backward): bench.py's secondary figure

and writes profiles/cpu_port_calibration.json; bench.py quotes the ratios next to its numbers.

    python -m oracle.calibrate_port [--batch 4] [--steps 5]

TEST INFRASTRUCTURE ONLY.  Reference step = BEV/main.py:213-223,264-265: model(x, True) -> Area_Loss per lane ->
zero_grad -> backward, train mode (Dropout2d left ON: its cost is part of the reference's step).
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import e2e_oracle, erfnet_oracle, fit_oracle, inputs, ref_shims, vendor_baseline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3, help="the three legs are timed in turn this many times (clock drift)")
    a = ap.parse_args()
    assert ref_shims.available(), "needs /root/reference"
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    N, R = a.batch, 256
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=61))
    gt = inputs.bev_gt_params(N, seed=62)
    P = erfnet_oracle.make_params(seed=4, out_channels=2)

    ref = ref_shims.load("bev")
    model = ref.LSQ_layer.Net(ref_shims.default_args("bev", batch_size=N))
    sd = model.net.state_dict()
    model.net.load_state_dict({k: P[k] for k in sd})
    model.train()
    crit = ref.Loss_crit.Area_Loss(2, "none")
    gtt = torch.from_numpy(gt)

    def ref_step():
        b0, b1, _, _, _, _, _, _, _ = model(x, True)
        loss = crit(b0, gtt[:, 0]) + crit(b1, gtt[:, 1])
        model.zero_grad()
        loss.backward()
        return float(loss)

    Pa = vendor_baseline.trainable_params(4, "cpu")
    grid = vendor_baseline.bev_grid(R, "cpu")
    zr = fit_oracle.zero_rows_of(R, 0.3)

    def aten_step():
        return float(vendor_baseline.bev_step(x, Pa, gtt, grid, zr)[0])

    def port_step():
        return e2e_oracle.bev_step(x, P, gt, torch.float32, R)["loss"]

    legs = (("reference", ref_step), ("aten", aten_step), ("port", port_step))
    ts = {name: [] for name, _ in legs}
    for name, fn in legs:
        fn()                                           # warm-up: allocator, thread pool, oneDNN primitive cache
    for _ in range(a.rounds):
        for name, fn in legs:
            for _ in range(a.steps):
                t0 = time.perf_counter()
                fn()
                ts[name].append(time.perf_counter() - t0)
    out = {name: {"images_per_sec": N / float(np.median(t)), "median_s": float(np.median(t)), "steps": len(t)} for name, t in ts.items()}
    out["ratio_port_over_reference"] = out["port"]["images_per_sec"] / out["reference"]["images_per_sec"]
    out["ratio_aten_over_reference"] = out["aten"]["images_per_sec"] / out["reference"]["images_per_sec"]
    out["batch"], out["threads"] = N, threads
    out["cpu"] = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    out["note"] = ("reference = unmodified BEV LSQ_layer.Net + Area_Loss from /root/reference (cv2 stub, masked_select cast), "
                   "train mode with its Dropout2d; aten = oracle/vendor_baseline.bev_step on the CPU (the reference's "
                   "torch.nn.functional / bmm / inverse calls, Dropout2d on); port = oracle/e2e_oracle.bev_step fp32 (functional "
                   "torch backbone, fp64 numpy fit + loss with analytic backward, no dropout); legs timed in turn, medians")
    with open(os.path.join(ROOT, "profiles", "cpu_port_calibration.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
