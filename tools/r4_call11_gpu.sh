#!/bin/bash
# round 4: 64-product segments with the even / odd tiles flushing in turn (no branch, half the adds per pair) vs the shipped
# 32-product form: accuracy (six seeds) and speed on one box
O=gpurun_out/r4g; mkdir -p $O
L=lanedetection_end2end_amd/liblanefit_hip.so
cp $L /tmp/head.so
export TMPDIR=/tmp
( timeout 900 python tools/ratio_seeds.py --oracle /tmp/ratio --seeds 6 > $O/oracle.log 2>&1; echo done > /tmp/ratio.done ) &
sleep 100        # the CPU legs use every core: keep them away from the timing below
while [ ! -f /tmp/ratio.done ]; do sleep 2; done
timeout 300 python tools/ab_conv.py tools/ab/liblanefit_r3final.so tools/ab/liblanefit_r4head.so tools/ab/liblanefit_r4seg64h.so > $O/ab_conv.txt 2>&1
tail -12 $O/ab_conv.txt
B="python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline"
for v in r4head r4seg64h r4head r4seg64h; do
  cp tools/ab/liblanefit_$v.so $L
  timeout 150 $B 2> $O/err_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['families'])" | tee -a $O/ab_bench.txt
done
for v in r4head r4seg64h; do
  cp tools/ab/liblanefit_$v.so $L
  echo "== $v" | tee -a $O/ratio.txt
  timeout 300 python tools/ratio_seeds.py --hip /tmp/ratio --seeds 6 2>&1 | tail -4 | tee -a $O/ratio.txt
done
cp /tmp/head.so $L
