#!/bin/bash
# after a kernel change: kernel-level timings + conv parity tests + the headline bench
mkdir -p gpurun_out
python tools/kbench.py --iters 300 > gpurun_out/r2_kbench6.txt 2>&1
tail -12 gpurun_out/r2_kbench6.txt
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q > gpurun_out/r2_pytest6.txt 2>&1
tail -3 gpurun_out/r2_pytest6.txt
python bench.py --no-cpu-baseline > gpurun_out/r2_bench6.json 2> gpurun_out/r2_bench6.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench6.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["families"], d.get("fp32_split_x9", {}).get("value"))
PY
