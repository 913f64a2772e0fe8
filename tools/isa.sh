#!/bin/bash
# tools/isa.sh <file.hip> <kernel-name-substring>: compile one source for gfx950 and print resource usage + main-loop mix
set -e
src=/root/repo/lanedetection_end2end_amd/csrc/$1
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -c $src -o /tmp/isa_tmp.o -save-temps=obj 2>&1 | grep -v "^$" | head -20
