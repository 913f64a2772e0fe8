#!/bin/bash
# bf16 / split / lean kernels after the buffer-addressing change: parity tests + the bench lines of every precision mode
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q > gpurun_out/r2_pytest8.txt 2>&1
tail -3 gpurun_out/r2_pytest8.txt
python bench.py --no-cpu-baseline > gpurun_out/r2_bench_modes_fp32.json 2>/dev/null
python bench.py --no-cpu-baseline --precision bf16 > gpurun_out/r2_bench_modes_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --workload bp --precision bf16 > gpurun_out/r2_bench_modes_bp_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --workload bp > gpurun_out/r2_bench_modes_bp_fp32.json 2>/dev/null
python bench.py --no-cpu-baseline --workload seg > gpurun_out/r2_bench_modes_seg_fp32.json 2>/dev/null
python bench.py --no-cpu-baseline --workload epoch > gpurun_out/r2_bench_modes_epoch.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2_bench_modes_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(f.split("bench_modes_")[1], d["value"], d["ms_per_step"], {k: v["tflops"] for k, v in r.get("families", {}).items()}, d.get("fp32_split_x9", {}).get("value"), d.get("roofline_hbm"), d.get("loss_first_last"))
    except Exception as e:
        print(f, "ERR", e)
PY
