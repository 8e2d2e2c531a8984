#!/bin/bash
# Round-3 last GPU call: rocprofv3 kernel stats of the 1080p inference workload with the row-band warp kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 70 rocprofv3 --kernel-trace --stats -d $O/prof_v_inf -o inf -- python $R/tools/bench_infer.py > $O/prof_v_inf.log 2>&1
db=$(find $O/prof_v_inf -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r03_infer1080p_bf16_kernel_stats.txt; rm -rf $O/prof_v_inf
grep -n "warp_s2d\|total kernel" $O/r03_infer1080p_bf16_kernel_stats.txt | cut -c1-160
