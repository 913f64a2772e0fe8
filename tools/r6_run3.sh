#!/bin/bash
# round 6, GPU call 3: full GPU suite on the current tree, config-3 kernel trace, wgrad traffic experiment, bf16 A/B, headline check
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r6_run3
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 300 python tools/wgrad_traffic.py --iters 200 > $O/wgrad_traffic.txt 2>&1; cat $O/wgrad_traffic.txt | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_bp16 -o bench -- python bench.py --workload bp --precision bf16 --steps 8 --warmup 3 --min-seconds 0 --no-extras > $O/trace_bp16.json 2> $O/trace_bp16.err
DB=$(find $O/trace_bp16 -name '*_results.db' | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py "$DB" > $O/bp_bf16_kernel_stats.txt && head -45 $O/bp_bf16_kernel_stats.txt | cut -c1-150
rm -rf $O/trace_bp16
timeout 300 python bench.py --workload bp --precision bf16 --no-extras --min-seconds 3 > $O/bench_bp16.json 2> $O/bench_bp16.err
python -c "import json; d=json.load(open('$O/bench_bp16.json')); print('bp bf16', d['value'], d['ms_per_step'], d['roofline']['families'])"
timeout 300 python bench.py --no-extras --min-seconds 3 > $O/bench_bev.json 2> $O/bench_bev.err
python -c "import json; d=json.load(open('$O/bench_bev.json')); print('bev fp32', d['value'], d['ms_per_step'], d['roofline']['families'])"
