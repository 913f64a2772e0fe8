cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r2_pytest_gpu3.txt 2>&1
tail -8 gpurun_out/r2_pytest_gpu3.txt
grep -n "FAILED\|Error" gpurun_out/r2_pytest_gpu3.txt | head -20
python bench.py > gpurun_out/r2_bench4.json 2> gpurun_out/r2_bench4.err
python -c "
import json; d=json.load(open('gpurun_out/r2_bench4.json')); print(d['value'], d['ms_per_step'], d['timed_blocks_ms_per_step'], d['roofline']['families'], d.get('fp32_split_x9'), d['cpu_baseline'], d['parity'])"
