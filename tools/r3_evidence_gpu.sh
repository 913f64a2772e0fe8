#!/bin/bash
# round-3 evidence pass (run through gpurun from the repo root): smoke(), per-wave phase stamps of the tap-GEMM and the weight
# gradient at HEAD, wait / issue counters of the two dominant kernels.  Every step under its own timeout.
set -u
export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
timeout 240 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" > $O/rc.txt
timeout 200 python tools/kbench.py --phases > $O/phases_fwd.txt 2>&1; echo "phases rc=$?" >> $O/rc.txt
timeout 200 python tools/kbench.py --phases --wgrad > $O/phases_wgrad.txt 2>&1; echo "phases wgrad rc=$?" >> $O/rc.txt
for shp in "128 32 64 1 16" "64 64 128 1 1"; do
  tag=$(echo $shp | cut -d' ' -f1)
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_$tag -o p -- python tools/kbench.py --iters 3 --one $shp > $O/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?" >> $O/rc.txt
  DB=$(find $O/pmc_$tag -name '*_results.db' | head -1)
  echo "## kbench --one $shp" >> $O/r3_pmc_tapgemm.txt
  python profiles/summarize_pmc.py "$DB" 3 >> $O/r3_pmc_tapgemm.txt 2>&1
  rm -rf $O/pmc_$tag
done
cat $O/rc.txt; tail -3 $O/smoke.txt
