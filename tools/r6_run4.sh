#!/bin/bash
# round 6, GPU call 4: channel-major BatchNorm partial rows -- full GPU suite, both step traces, benches
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r6_run4
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for W in "bev fp32" "bp bf16"; do
  set -- $W
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$1 -o bench -- python bench.py --workload $1 --precision $2 --steps 8 --warmup 3 --min-seconds 0 --no-extras > $O/trace_$1.json 2> $O/trace_$1.err
  DB=$(find $O/trace_$1 -name '*_results.db' | head -1)
  [ -n "$DB" ] && python profiles/summarize_rocpd.py "$DB" > $O/$1_kernel_stats.txt && grep -E "kernel time|finalize|bn_|wv_kernel|reduce" $O/$1_kernel_stats.txt | cut -c1-150
  rm -rf $O/trace_$1
done
timeout 300 python bench.py --workload bp --precision bf16 --no-extras --min-seconds 3 > $O/bench_bp16.json 2> $O/bench_bp16.err
python -c "import json; d=json.load(open('$O/bench_bp16.json')); print('bp bf16', d['value'], d['ms_per_step'], d['roofline']['families'])"
timeout 300 python bench.py --no-extras --min-seconds 3 > $O/bench_bev.json 2> $O/bench_bev.err
python -c "import json; d=json.load(open('$O/bench_bev.json')); print('bev fp32', d['value'], d['ms_per_step'], d['roofline']['families'])"
