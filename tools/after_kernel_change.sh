#!/bin/bash
# (every step under its own `timeout`: a crashed rocprofv3 child once sat on the box until gpurun's limit, 20 GPU-minutes)
# after a kernel change: kernel-level timings + conv parity tests + the headline bench (+ kernel stats)
#   gpurun --timeout 900 -- 'bash tools/after_kernel_change.sh <tag> [pytest-args]'
TAG=${1:-x}
shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/kbench.py --iters 300 > gpurun_out/${TAG}_kbench.txt 2>&1
tail -14 gpurun_out/${TAG}_kbench.txt
timeout 900 python -m pytest ${@:-tests/test_backbone_gpu.py} -m gpu -x -q > gpurun_out/${TAG}_pytest.txt 2>&1
tail -5 gpurun_out/${TAG}_pytest.txt
timeout 240 python bench.py --no-cpu-baseline --no-vendor-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["families"], d.get("fp32_split_x9", {}).get("value"))
PY
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o bench -- python bench.py --steps 10 --warmup 3 --min-seconds 1 --no-cpu-baseline --no-vendor-baseline > gpurun_out/${TAG}_trace.json 2> gpurun_out/${TAG}_trace.err
DB=$(find gpurun_out/prof_${TAG} -name '*_results.db' | head -1)
python profiles/summarize_rocpd.py "$DB" > gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_${TAG}
head -40 gpurun_out/${TAG}_kernel_stats.txt
