#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o inf -- python $R/tools/bench_infer.py > $O/r03l_prof_inf.log 2>&1
db=$(find /tmp/prof_inf -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r03l_infer1080p_bf16_kernel_stats.txt
head -24 $O/r03l_infer1080p_bf16_kernel_stats.txt | cut -c1-150
