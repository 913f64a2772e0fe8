#!/bin/bash
# round 6, GPU call 5: paired-job fp32 weight gradient -- kernel tests, A/B, headline
cd "$(dirname "$0")/.."
O=gpurun_out/r6_run5
mkdir -p $O
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_baseline_configs_gpu.py tests/test_blocks_gpu.py tests/test_bf16_kernels_gpu.py tests/test_lean_gpu.py -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/wgrad_traffic.py --iters 200 > $O/wgrad_traffic.txt 2>&1; cat $O/wgrad_traffic.txt | cut -c1-420
for rep in 1 2; do
  for pair in 1 0; do
    timeout 300 python -c "
import sys, runpy
sys.path.insert(0, '.')
from lanedetection_end2end_amd import _lib
_lib.load().lf_debug_set_wgrad_pair($pair)
sys.argv = ['bench.py', '--no-extras', '--min-seconds', '3']
runpy.run_path('bench.py', run_name='__main__')" > $O/bench_bev_pair${pair}_$rep.json 2> $O/bench_bev.err
    python -c "import json; d=json.load(open('$O/bench_bev_pair${pair}_$rep.json')); print('bev fp32 pair=$pair', d['value'], d['ms_per_step'], d['roofline']['families'])"
  done
done
