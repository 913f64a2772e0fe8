mkdir -p gpurun_out/rec
python bench.py > gpurun_out/rec/r5_bench.json 2> gpurun_out/rec/r5_bench.err
python bench.py --workload bp --precision bf16 > gpurun_out/rec/r5_bench_bp_320x640_b64_bf16.json 2> gpurun_out/rec/bp16.err
python bench.py --workload bp > gpurun_out/rec/r5_bench_bp_320x640_b64.json 2> gpurun_out/rec/bp.err
python bench.py --precision bf16 > gpurun_out/rec/r5_bench_bev_bf16.json 2> gpurun_out/rec/bev16.err
python bench.py --workload seg > gpurun_out/rec/r5_bench_seg_512x1024_b16.json 2> gpurun_out/rec/seg.err
python bench.py --workload epoch > gpurun_out/rec/r5_bench_epoch.json 2> gpurun_out/rec/epoch.err
for f in gpurun_out/rec/*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('traffic_source',{}).get('measured_on_these_sources'), (d.get('parity') or {}).get('ok'))"; done
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/rec/r5_pytest_gpu.txt; cat gpurun_out/rec/r5_pytest_gpu.txt
