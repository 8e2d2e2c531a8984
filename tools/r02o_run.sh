#!/bin/bash
# Round-2 GPU session O: two consecutive steps' segment schedule; CU-mask partitioning microbenchmark.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 200 python tools/seg_timeline.py --steps 40 2>&1 | tail -34 | tee $O/r02o_seg_timeline.txt
timeout 300 python tools/mb_cumask.py 2>&1 | tail -12 | tee $O/r02o_cumask.txt
