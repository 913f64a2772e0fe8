#!/bin/bash
# GPU validation pass: the whole gpu suite, then the bf16 bench lines and the epoch workload.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_pytest_gpu4.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest_gpu4.txt
tail -5 gpurun_out/r2_pytest_gpu4.txt
grep -n "loss: first\|seg-mode\|FAILED\|Error" gpurun_out/r2_pytest_gpu4.txt | head -40
