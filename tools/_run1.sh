set -x
cd /root/repo
python tools/kbench.py --variants 0 2 --iters 20 > gpurun_out/r2_kbench1.txt 2>&1
tail -12 gpurun_out/r2_kbench1.txt
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q > gpurun_out/r2_pytest_backbone1.txt 2>&1
tail -15 gpurun_out/r2_pytest_backbone1.txt
python bench.py --no-cpu-baseline > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
cat gpurun_out/r2_bench1.json | head -c 3000
