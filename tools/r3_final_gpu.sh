set -u
mkdir -p gpurun_out/r3m
timeout 420 python -m pytest tests/test_clas_gpu.py tests/test_baseline_configs_gpu.py tests/test_main_loop_gpu.py -m gpu -q -s > gpurun_out/r3m/pytest.txt 2>&1; echo "pytest rc=$?" > gpurun_out/r3m/rc.txt
timeout 600 bash profiles/collect.sh r3 > gpurun_out/r3m/collect.log 2>&1; echo "collect rc=$?" >> gpurun_out/r3m/rc.txt
timeout 400 python bench.py > gpurun_out/r3m/r3_bench.json 2> gpurun_out/r3m/bench.err; echo "bench rc=$?" >> gpurun_out/r3m/rc.txt
LF_BENCH_SINGLE_DEVICE=1 LF_BENCH_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 3 --min-seconds 1 --no-cpu-baseline --no-vendor-baseline > gpurun_out/r3m/r3_bench_2ranks_one_device.json 2> gpurun_out/r3m/bench2.err; echo "bench2 rc=$?" >> gpurun_out/r3m/rc.txt
for cfg in "bp:fp32:r3_bench_bp_320x640_b64" "bp:bf16:r3_bench_bp_320x640_b64_bf16" "seg:fp32:r3_bench_seg_512x1024_b16" "bev:bf16:r3_bench_bev_bf16" "bev:fp32x9:r3_bench_bev_fp32x9"; do
  IFS=: read w p name <<< "$cfg"
  timeout 150 python bench.py --workload $w --precision $p --min-seconds 2 --no-cpu-baseline --no-vendor-baseline > gpurun_out/r3m/$name.json 2>> gpurun_out/r3m/bench_cfg.err; echo "$name rc=$?" >> gpurun_out/r3m/rc.txt
done
timeout 200 python bench.py --workload epoch --no-cpu-baseline --no-vendor-baseline > gpurun_out/r3m/r3_bench_epoch.json 2>> gpurun_out/r3m/bench_cfg.err; echo "epoch rc=$?" >> gpurun_out/r3m/rc.txt
cat gpurun_out/r3m/rc.txt; tail -5 gpurun_out/r3m/pytest.txt
