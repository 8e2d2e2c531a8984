#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_frvsr10 -o frvsr -- python $R/bench.py --no-cpu-baseline > $O/prof_frvsr10.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_teco7 -o teco -- python $R/bench.py --config tecogan --steps 10 --no-cpu-baseline > $O/prof_teco7.log 2>&1
for n in frvsr10:frvsr teco7:teco; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r01i_${f}_kernel_stats.txt; done
head -45 $O/r01i_teco_kernel_stats.txt
