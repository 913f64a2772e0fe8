mkdir -p gpurun_out
python -m pytest tests/test_lean_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/lean_test.txt
python -m pytest tests/test_backbone_gpu.py -x -q -k "golden_and_grads or full_size or eval_mode or ragged or kernel_parity_every" 2>&1 | tail -8 >> gpurun_out/lean_test.txt
python -m pytest tests/test_blocks_gpu.py -x -q 2>&1 | tail -4 >> gpurun_out/lean_test.txt
python tools/lean_ab.py > gpurun_out/lean_ab.txt 2>&1
for m in 1 0 3 2 1 0; do
python - $m > gpurun_out/lean_bench_$m.json 2>gpurun_out/lean_bench_$m.err <<'PY'
import sys, runpy
from lanedetection_end2end_amd import _lib
_lib.load().lf_debug_set_lean_p(int(sys.argv[1]))
sys.argv = ["bench.py", "--no-extras"]
runpy.run_path("bench.py", run_name="__main__")
PY
echo "mode $m: $(python -c "import json,sys; d=json.load(open('gpurun_out/lean_bench_$m.json')); print(d['value'], d['ms_per_step'])")" >> gpurun_out/lean_modes.txt
done
cat gpurun_out/lean_test.txt; cat gpurun_out/lean_ab.txt | tail -20; cat gpurun_out/lean_modes.txt
