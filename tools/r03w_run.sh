#!/bin/bash
# Round-3 GPU session W (final validation after sessions Q-U): full GPU suite, the default bench line (with the bf16
# gradient / trajectory sub-records), rocprofv3 kernel stats of the three workloads, segment timeline.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=6 ) > $O/r03w_pytest_gpu.log 2>&1; grep -E "passed|failed" $O/r03w_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/r03w_pytest_gpu.log | cut -c1-200
( time timeout 420 python bench.py ) > $O/r03w_bench.json 2> $O/r03w_bench.err; cut -c1-400 $O/r03w_bench.json; tail -4 $O/r03w_bench.err
cd /tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_w_teco -o teco -- $B --steps 20 --warmup 3 > $O/prof_w_teco.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_w_frvsr -o frvsr -- $B --steps 40 --warmup 3 --config frvsr > $O/prof_w_frvsr.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_w_inf -o inf -- python $R/tools/bench_infer.py > $O/prof_w_inf.log 2>&1
for n in teco:tecogan frvsr:frvsr inf:infer1080p; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_w_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r03w_${f}_bf16_kernel_stats.txt; rm -rf $O/prof_w_$d; done
head -8 $O/r03w_tecogan_bf16_kernel_stats.txt | cut -c1-160
cd $R; timeout 120 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -18 > $O/r03w_seg_timeline.txt; cat $O/r03w_seg_timeline.txt
