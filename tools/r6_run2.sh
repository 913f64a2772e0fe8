#!/bin/bash
# round 6, GPU call: the wave-private 64-channel bf16 kernel -- race screens, A/B against the whole-line kernel, config-3 bench;
# the straight-through bf16 block tests; the reseeded fp32 block test
cd "$(dirname "$0")/.."
O=gpurun_out/r6_run2
mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_kernels_gpu.py tests/test_blocks_gpu.py -q -s 2>&1 | grep -v "^$" > $O/pytest_kernels.txt; grep -E "^bf16 |passed|failed|Error|assert|worst" $O/pytest_kernels.txt | tail -40
timeout 600 python tools/bf16_ab.py --iters 100 > $O/bf16_ab.txt 2>&1; cat $O/bf16_ab.txt | cut -c1-330
timeout 300 python bench.py --workload bp --precision bf16 --no-extras --min-seconds 3 > $O/bench_bp16.json 2> $O/bench_bp16.err
python -c "import json; d=json.load(open('$O/bench_bp16.json')); print('bp bf16', d['value'], d['ms_per_step'], d['roofline']['families'])"; tail -3 $O/bench_bp16.err
