#!/bin/bash
# round 6, GPU call 1: GPU suite, the new bench line, A/B of the accumulator flush (two v_pk_add_f32 vs four v_add_f32 per tile)
cd "$(dirname "$0")/.."
O=gpurun_out/r6_run1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
L=lanedetection_end2end_amd/liblanefit_hip.so
cp $L $O/base.so
for rep in 1 2 3; do
  for v in base flush_scalar; do
    if [ $v = base ]; then cp $O/base.so $L; else cp tools/ab/liblanefit_flush_scalar.so $L; fi
    timeout 300 python bench.py --no-extras --min-seconds 4 > $O/ab_${v}_$rep.json 2>> $O/ab.err
    python -c "import json; d=json.load(open('$O/ab_${v}_$rep.json')); print('$v', $rep, d['value'], d['ms_per_step'], d['roofline']['families'])"
  done
done
cp $O/base.so $L
timeout 300 python tools/ab_conv.py $O/base.so tools/ab/liblanefit_flush_scalar.so > $O/ab_conv.txt 2>&1; tail -30 $O/ab_conv.txt
rm -f $O/base.so
