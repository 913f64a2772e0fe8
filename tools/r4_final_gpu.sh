#!/bin/bash
# round 4: full GPU suite, profile evidence (profiles/collect.sh r4), the bench lines of every workload.  Run after
# tools/stamp_head.sh on a clean tree:   gpurun --timeout 2400 -- 'bash tools/r4_final_gpu.sh'
O=gpurun_out/r4z; mkdir -p $O
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_all.txt 2>&1; tail -25 $O/pytest_all.txt
bash profiles/collect.sh r4 > $O/collect.log 2>&1; tail -25 $O/collect.log
cp profiles/r4_* profiles/traffic.json $O/ 2>/dev/null
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
for W in "bp fp32" "bp bf16" "seg fp32" "bev bf16" "bev fp32x9"; do
  set -- $W
  timeout 300 python bench.py --workload $1 --precision $2 --no-cpu-baseline --no-vendor-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python -c "import json; d=json.loads(open('$O/bench_$1_$2.json').read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['families'], d.get('roofline_hbm'))"
done
timeout 300 python bench.py --workload epoch > $O/bench_epoch.json 2> $O/bench_epoch.err; tail -c 400 $O/bench_epoch.json
LF_BENCH_SINGLE_DEVICE=1 LF_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 10 --warmup 3 --min-seconds 1 --no-cpu-baseline --no-vendor-baseline > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err; tail -c 600 $O/bench_2ranks_one_device.json
