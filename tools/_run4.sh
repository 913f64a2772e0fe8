cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r2_pytest_gpu2.txt 2>&1
tail -8 gpurun_out/r2_pytest_gpu2.txt
grep -n "hip-cpu64\|C2 BEV\|C3 BP\|C5 seg\|parameter-gradient\|eval-mode backward\|FAILED\|Error" gpurun_out/r2_pytest_gpu2.txt | head -80
python bench.py --no-cpu-baseline > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err
python -c "
import json; d=json.load(open('gpurun_out/r2_bench3.json')); print(d['value'], d['ms_per_step'], d['timed_blocks_ms_per_step'], d['roofline']['families'], d.get('fp32_split_x9'))"
