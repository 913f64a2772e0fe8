#!/usr/bin/env python
"""How long does the vendor-stack baseline leg of bench.py take on a fresh box (MIOpen compiles / tunes its kernels on first use)?
    python tools/vendor_time.py [0|1]      # cudnn.benchmark off / on
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fit_oracle, inputs, vendor_baseline  # noqa: E402

bench = len(sys.argv) > 1 and sys.argv[1] == "1"
torch.backends.cudnn.benchmark = bench
B, R = 32, 256
P = vendor_baseline.trainable_params(4, "cuda")
x = torch.from_numpy(inputs.images(B, R, 2 * R, seed=100)).cuda()
gt = torch.from_numpy(inputs.bev_gt_params(B, seed=200)).cuda()
grid = vendor_baseline.bev_grid(R, "cuda")
zr = fit_oracle.zero_rows_of(R, 0.3)
t0 = time.perf_counter()
vendor_baseline.bev_step(x, P, gt, grid, zr)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(4):
    vendor_baseline.bev_step(x, P, gt, grid, zr)
torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(20):
    vendor_baseline.bev_step(x, P, gt, grid, zr)
torch.cuda.synchronize()
t3 = time.perf_counter()
print("cudnn.benchmark=%s: first step %.1f s, next four %.1f s, steady %.2f ms/step = %.1f images/s"
      % (bench, t1 - t0, t2 - t1, 1e3 * (t3 - t2) / 20, B * 20 / (t3 - t2)))
