#!/bin/bash
# PMC comparison of two library builds on the 64-channel conv launches (tuning aid): tools/ab_pmc.sh <tag> <libA> <libB> counters...
TAG=$1; A=$2; B=$3; shift 3
export TMPDIR=/tmp
mkdir -p gpurun_out
for L in $A $B; do
  n=$(basename $L .so)
  for C in "$@"; do
    rocprofv3 --pmc $C --kernel-trace -d gpurun_out/pmc_${TAG}_${n}_${C// /_} -o p -- python tools/ab_conv.py --c64 $L > /dev/null 2>&1
    DB=$(find gpurun_out/pmc_${TAG}_${n}_${C// /_} -name '*_results.db' | head -1)
    echo "== $n $C"
    python profiles/summarize_pmc.py "$DB" 3 | grep -v "^#" | cut -c1-220
    rm -rf gpurun_out/pmc_${TAG}_${n}_${C// /_}
  done
done
