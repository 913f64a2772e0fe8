cd /root/repo
python tools/kbench.py --variants 0 2 --iters 20 > gpurun_out/r2_kbench2.txt 2>&1
grep "^C=" gpurun_out/r2_kbench2.txt
python tools/kbench.py --phases > gpurun_out/r2_phases2.txt 2>&1
cat gpurun_out/r2_phases2.txt
