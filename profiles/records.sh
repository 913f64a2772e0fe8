#!/bin/bash
# The round's bench records and the GPU test run, on a 1-GPU MI355X box (through gpurun from the repo root, AFTER profiles/collect.sh's
# outputs have been copied into profiles/ and committed: bench.py reports whether profiles/traffic*.json were measured on these sources):
#
#     gpurun --timeout 2400 -- 'bash profiles/records.sh r5'
#
# Writes gpurun_out/rec/<tag>_bench*.json (one JSON line each: the driver's contract) and <tag>_pytest_gpu.txt; copy them into profiles/.
TAG=${1:-r5}
cd "$(dirname "$0")/.."
R=gpurun_out/rec
mkdir -p $R
python bench.py > $R/${TAG}_bench.json 2> $R/bench.err                                                  # the headline (all extras)
python bench.py --workload bp --precision bf16 > $R/${TAG}_bench_bp_320x640_b64_bf16.json 2> $R/bp16.err   # BASELINE config 3: geometry and dtype
python bench.py --workload bp > $R/${TAG}_bench_bp_320x640_b64.json 2> $R/bp.err
python bench.py --precision bf16 > $R/${TAG}_bench_bev_bf16.json 2> $R/bev16.err
python bench.py --workload seg > $R/${TAG}_bench_seg_512x1024_b16.json 2> $R/seg.err                      # config 5, one GPU's shard
python bench.py --workload epoch > $R/${TAG}_bench_epoch.json 2> $R/epoch.err                             # config 4
for f in $R/${TAG}_bench*.json; do
  python -c "import json; d=json.load(open('$f')); r=d.get('roofline') or {}; print('$f', d['value'], d['unit'], d['ms_per_step'], 'traffic on these sources:', (r.get('traffic_step') or {}).get('measured_on_these_sources'), 'parity ok:', (d.get('parity') or {}).get('ok'))"
done
# the six-seed distance-to-fp64 ratios of the fp32 and fp32x9 modes (printed by the test; kept as evidence)
( echo "# python -m pytest tests/test_baseline_configs_gpu.py -q -s -k ratio  (|hip - cpu64| / |cpu32 - cpu64| over six seeds, 8 x 3 x 256 x 512)"; python -m pytest tests/test_baseline_configs_gpu.py -q -s -k ratio 2>&1 | grep -E "^\[|passed|failed" ) > $R/${TAG}_ratio_seeds.txt
python -m pytest tests -m gpu -q > $R/pytest_full.txt 2>&1
( echo "# python -m pytest tests -m gpu -q  (commit $(cut -d' ' -f1 .git_head 2>/dev/null))"; grep -E "passed|failed|error" $R/pytest_full.txt | tail -3 ) > $R/${TAG}_pytest_gpu.txt
cat $R/${TAG}_pytest_gpu.txt
