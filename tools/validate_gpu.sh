#!/bin/bash
# full validation: gpu suite, profile collection, final bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_final.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest_final.txt
tail -4 gpurun_out/r2_pytest_final.txt
bash profiles/collect.sh r2 > gpurun_out/r2_collect.log 2>&1
tail -5 gpurun_out/r2_collect.log
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
tail -c 1500 gpurun_out/r2_bench_final.json
