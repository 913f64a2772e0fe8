#!/bin/bash
# round 4, GPU call 1: the two-accumulator tap-GEMM against the round-3 library on ONE box -- accuracy (ratio over seeds),
# kernel-level and whole-step timing, conv parity tests.   gpurun --timeout 1500 -- 'bash tools/r4_call1_gpu.sh'
set -u
O=gpurun_out/r4a; mkdir -p $O
L=lanedetection_end2end_amd/liblanefit_hip.so
cp $L /tmp/head.so
export TMPDIR=/tmp
nproc > $O/nproc.txt
# CPU legs of the ratio statistic run beside the (timing-insensitive) parity tests
( timeout 900 python tools/ratio_seeds.py --oracle /tmp/ratio --seeds 6 > $O/oracle.log 2>&1; echo done > /tmp/ratio.done ) &
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q > $O/pytest_backbone.txt 2>&1
tail -5 $O/pytest_backbone.txt
while [ ! -f /tmp/ratio.done ]; do sleep 2; done
tail -3 $O/oracle.log
for v in r3final r4dual r4seg64; do
  cp tools/ab/liblanefit_$v.so $L
  echo "== $v" | tee -a $O/ratio.txt
  timeout 300 python tools/ratio_seeds.py --hip /tmp/ratio --seeds 6 2>&1 | tail -3 | tee -a $O/ratio.txt
done
cp /tmp/head.so $L
timeout 300 python tools/ab_conv.py tools/ab/liblanefit_r3final.so tools/ab/liblanefit_r4dual.so tools/ab/liblanefit_r4seg64.so > $O/ab_conv.txt 2>&1
tail -30 $O/ab_conv.txt
B="python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline"
for v in r3final r4dual r4seg64 r3final r4dual; do
  cp tools/ab/liblanefit_$v.so $L
  timeout 150 $B 2> $O/err_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['families'])" | tee -a $O/ab_bench.txt
done
cp /tmp/head.so $L
