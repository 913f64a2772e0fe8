#!/bin/bash
# Writes .git_head (untracked; travels to the GPU box, where there is no .git): the commit the working tree is at, whether the
# tree is clean, and a digest of the sources the profiles depend on -- profiles/collect.sh refuses to write evidence for a
# dirty or mismatching tree.     bash tools/stamp_head.sh && gpurun ... 'bash profiles/collect.sh r4'
cd "$(dirname "$0")/.."
[ -x tools/l2_stream ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/l2_stream.hip -o tools/l2_stream
[ -x tools/ticket_tail ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result tools/ticket_tail.hip -o tools/ticket_tail
[ -x tools/write_policy ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result tools/write_policy.hip -o tools/write_policy
HEAD=$(git rev-parse HEAD)
DIRTY=$(git status --porcelain --untracked-files=no | wc -l)
DIGEST=$(cat lanedetection_end2end_amd/csrc/*.hip lanedetection_end2end_amd/csrc/*.h bench.py | sha256sum | cut -c1-16)
echo "$HEAD clean=$([ "$DIRTY" = 0 ] && echo yes || echo no) digest=$DIGEST" > .git_head
cat .git_head
