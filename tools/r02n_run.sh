#!/bin/bash
# Round-2 GPU session N2: segment schedule + the gap between consecutive steps.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 200 python tools/seg_timeline.py --steps 40 2>&1 | tail -30 | tee $O/r02n_seg_timeline.txt
TG_OVERLAP_PARTS=0 TG_SEGMENTS=force timeout 200 python tools/seg_timeline.py --steps 40 2>&1 | tail -12 | tee $O/r02n_seg_timeline_serial.txt
