#!/bin/bash
# round 4, GPU call 5: LDS-staged bf16 tap-GEMM: A/B vs the streaming kernel, bf16 parity tests, config-3 bench in bf16
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/bf16_ab.py > $O/bf16_ab.txt 2>&1; tail -12 $O/bf16_ab.txt
timeout 600 python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -k "bf16" > $O/pytest_bf16.txt 2>&1; tail -8 $O/pytest_bf16.txt
timeout 300 python bench.py --workload bp --precision bf16 --steps 10 --warmup 3 --min-seconds 2 > $O/bench_bp_bf16.json 2> $O/bench_bp_bf16.err
python -c "import json; d=json.loads(open('$O/bench_bp_bf16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families'], d.get('roofline_hbm'))"
timeout 300 python bench.py --precision bf16 --steps 10 --warmup 3 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline > $O/bench_bev_bf16.json 2> $O/bench_bev_bf16.err
python -c "import json; d=json.loads(open('$O/bench_bev_bf16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families'], d.get('roofline_hbm'))"
