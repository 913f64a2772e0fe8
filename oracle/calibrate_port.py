"""Port-vs-reference calibration of bench.py's cpu_baseline ("kind": "port"): times the REAL reference modules
(/root/reference, through oracle/ref_shims.py) and the port (oracle/e2e_oracle.bev_step, fp32 backbone) on the same
inputs, same thread count, in the authoring container (the GPU box has no /root/reference).  Writes
profiles/cpu_port_calibration.json; bench.py quotes the ratio next to the port's number.

    python -m oracle.calibrate_port [--batch 4] [--steps 5]

TEST INFRASTRUCTURE ONLY.  Reference step = BEV/main.py:213-223,264-265: model(x, True) -> Area_Loss per lane ->
zero_grad -> backward, train mode (dropout left ON in the reference: its cost is part of the reference's step).
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import e2e_oracle, erfnet_oracle, inputs, ref_shims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
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

    def port_step():
        return e2e_oracle.bev_step(x, P, gt, torch.float32, R)["loss"]

    out = {}
    for name, fn in (("reference", ref_step), ("port", port_step)):
        fn()
        ts = []
        for _ in range(a.steps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        out[name] = {"images_per_sec": N / float(np.median(ts)), "median_s": float(np.median(ts)), "steps": a.steps}
    out["ratio_port_over_reference"] = out["port"]["images_per_sec"] / out["reference"]["images_per_sec"]
    out["batch"], out["threads"] = N, threads
    out["cpu"] = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    out["note"] = ("reference = unmodified BEV LSQ_layer.Net + Area_Loss from /root/reference (cv2 stub, masked_select cast), "
                   "train mode with its Dropout2d; port = oracle/e2e_oracle.bev_step fp32 (functional torch backbone, fp64 numpy "
                   "fit + loss with analytic backward, no dropout)")
    with open(os.path.join(ROOT, "profiles", "cpu_port_calibration.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
