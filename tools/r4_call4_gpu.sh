#!/bin/bash
# round 4, GPU call 4: new cold-path kernels, transitive full-size tests with their printed numbers, the seeds ratio, epoch bench
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_coldpath_gpu.py tests/test_clas_gpu.py tests/test_main_loop_gpu.py "tests/test_backbone_gpu.py::test_segmentation_mode_fit_vs_reference_goldens" -m gpu -q -s > $O/pytest_cold.txt 2>&1
tail -25 $O/pytest_cold.txt
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -q -s -k "transitive or seeds" > $O/pytest_transitive.txt 2>&1
grep -n "BatchNorm statistics\|vs .* chunks\|first chunk\|per seed\|passed\|failed\|Error" $O/pytest_transitive.txt | head -40
timeout 200 python bench.py --workload epoch > $O/bench_epoch.json 2> $O/bench_epoch.err; tail -c 900 $O/bench_epoch.json
timeout 200 python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-vendor-baseline > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families'], d.get('fp32_split_x9',{}).get('value'))"
