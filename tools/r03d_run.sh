#!/bin/bash
# Round-3 session D: per-dispatch timeline of ONE serial TecoGAN step (TG_OVERLAP=0): isolated durations of every launch in program order
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
TG_OVERLAP=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_d -- $B --steps 3 --warmup 2 > $O/r03d_prof.log 2>&1
python $R/tools/timeline.py /tmp/prof_d $O/r03d_timeline_serial.csv --last 3400
