#!/bin/bash
# TIMING-ONLY ablation (results of the ablation library are wrong by construction: its weight gradients read gradient buffers that
# have been overwritten): all batched weight gradients of a backward pass launched at its END, barrier-less back to back, against
# the in-tree library.  Answers: what would taking the weight gradients off the data-gradient chain buy?
set -u
O=gpurun_out/r3q; mkdir -p $O
L=lanedetection_end2end_amd/liblanefit_hip.so
cp $L /tmp/head.so
B="python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline"
timeout 150 $B > $O/bench_head1.json 2> $O/err1.txt
cp tools/ab/liblanefit_wgrad_end_ablation.so $L; timeout 150 $B > $O/bench_ablation.json 2> $O/err2.txt
cp /tmp/head.so $L; timeout 150 $B > $O/bench_head2.json 2> $O/err3.txt
for f in head1 ablation head2; do python - $O/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -2 $O/err2.txt
