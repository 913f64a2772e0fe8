#!/bin/bash
# A/B/A on one box: in-tree library (bn1-backward apply dispatched barrier-less beside its block's conv3x1_2 weight gradient) against
# tools/ab/liblanefit_r3head2.so; if the new library is faster, the backward-order tests and the full bench line follow.
set -u
O=gpurun_out/r3r; mkdir -p $O
L=lanedetection_end2end_amd/liblanefit_hip.so
cp $L /tmp/new.so
B="python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline"
timeout 150 $B > $O/bench_new1.json 2> $O/err1.txt
cp tools/ab/liblanefit_r3head2.so $L; timeout 150 $B > $O/bench_head.json 2> $O/err2.txt
cp /tmp/new.so $L; timeout 150 $B > $O/bench_new2.json 2> $O/err3.txt
python - <<'PY' > $O/ab.txt
import json
v={}
for f in ("new1","head","new2"):
    try: v[f]=json.loads(open("gpurun_out/r3r/bench_%s.json"%f).read().strip().splitlines()[-1])["ms_per_step"]
    except Exception as e: v[f]=None
print(v)
ok = all(v.values()) and (v["new1"]+v["new2"])/2 < v["head"]*0.997
print("FASTER" if ok else "NOT_FASTER")
PY
cat $O/ab.txt
if grep -q "^FASTER" $O/ab.txt; then
  timeout 200 python -m pytest tests/test_backbone_gpu.py tests/test_clas_gpu.py tests/test_main_loop_gpu.py tests/test_dp_gpu.py -m gpu -q -x > $O/pytest_a.txt 2>&1; echo "pytest_a rc=$?" > $O/rc.txt
  timeout 260 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -q -x -k "not seeds and not bf16" > $O/pytest_b.txt 2>&1; echo "pytest_b rc=$?" >> $O/rc.txt
  timeout 200 python bench.py > $O/r3_bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
  cat $O/rc.txt; tail -2 $O/pytest_a.txt; tail -2 $O/pytest_b.txt
fi
