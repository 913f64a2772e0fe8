#!/bin/bash
set -u
O=gpurun_out/r3u; mkdir -p $O
timeout 200 python -m pytest tests/test_backbone_gpu.py tests/test_clas_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python bench.py > $O/r3_bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -2 $O/pytest.txt; python -c "
import json; d=json.loads(open('gpurun_out/r3u/r3_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['miopen_baseline']['value'], d['parity'].get('ok'), d['fp32_split_x9']['value'])"
