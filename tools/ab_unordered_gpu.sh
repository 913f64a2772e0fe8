#!/bin/bash
# A/B on one box: the in-tree library (data gradients launched without the in-order barrier beside their weight gradient) against
# tools/ab/liblanefit_r3head.so (all launches in order), then the backbone parity tests on the new library.
set -u
O=gpurun_out/r3o; mkdir -p $O
L=lanedetection_end2end_amd/liblanefit_hip.so
cp $L /tmp/new.so
B="python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline"
timeout 150 $B > $O/bench_new1.json 2> $O/err_new1.txt
cp tools/ab/liblanefit_r3head.so $L; timeout 150 $B > $O/bench_head.json 2> $O/err_head.txt
cp /tmp/new.so $L; timeout 150 $B > $O/bench_new2.json 2> $O/err_new2.txt
timeout 400 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x > $O/pytest_backbone.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
for f in new1 head new2; do python - $O/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("fp32_split_x9",{}).get("value"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
cat $O/rc.txt; tail -3 $O/pytest_backbone.txt
