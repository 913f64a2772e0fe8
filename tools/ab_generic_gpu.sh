set -u
O=gpurun_out/${OUT:-r3s}; mkdir -p $O
L=lanedetection_end2end_amd/liblanefit_hip.so
cp $L /tmp/head.so
B="python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline"
for v in ${VARIANTS:-head}; do
  if [ $v = head ]; then cp /tmp/head.so $L; else cp tools/ab/liblanefit_$v.so $L; fi
  timeout 150 $B 2> $O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['families'])"
done
cp /tmp/head.so $L
