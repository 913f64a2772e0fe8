cd /root/repo
python tools/kbench.py --phases > gpurun_out/r2_phases4.txt 2>&1
cat gpurun_out/r2_phases4.txt
