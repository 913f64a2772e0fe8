#!/bin/bash
set -u
O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 400 python bench.py > $O/r3_bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/pytest.txt; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3p/r3_bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["miopen_baseline"]["value"], d["parity"].get("ok"))
PY
