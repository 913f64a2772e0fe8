#!/bin/bash
# tools/isa_kernel.sh <asm.s> <mangled-name-substring> [top-n]: extract one kernel's ISA from a -save-temps assembly file
# (-> /tmp/isa_kernel.s) and print its instruction mix (top mnemonics) -- for tuning loops without a GPU
asm=$1; sub=$2
awk -v sub_="$sub" '$0 ~ "^_Z.*" sub_ ".*:" && !f {f=1} f{print} f && /^\.Lfunc_end/{exit}' "$asm" > /tmp/isa_kernel.s
wc -l /tmp/isa_kernel.s
grep -oE "^\s+(v|s|buffer|ds|global|scratch)_[a-z0-9_]+" /tmp/isa_kernel.s | sort | uniq -c | sort -rn | head -${3:-30}
