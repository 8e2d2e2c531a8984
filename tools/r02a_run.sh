#!/bin/bash
# Round-2 GPU session A: full GPU suite (new BASELINE-size parity tests), default bench line, rocprofv3 kernel stats of
# the headline workload, overlap experiment.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q -s --durations=15 ) > $O/r02a_pytest_gpu.log 2>&1; tail -25 $O/r02a_pytest_gpu.log
( time timeout 600 python bench.py ) > $O/r02a_bench.json 2> $O/r02a_bench.err; cut -c1-600 $O/r02a_bench.json; tail -5 $O/r02a_bench.err
for th in 16 8; do TG_C3_MAXTH=$th timeout 120 python tools/mb_overlap.py 2>&1 | tail -1 | tee -a $O/r02a_overlap.txt; done
MB_BIG=24 TG_C3_MAXTH=8 timeout 120 python tools/mb_overlap.py 2>&1 | tail -1 | tee -a $O/r02a_overlap.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_teco_a -o teco -- python $R/bench.py --steps 20 --no-sub --no-roofline --no-cpu-baseline > $O/prof_teco_a.log 2>&1
db=$(find $O/prof_teco_a -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r02a_tecogan_bf16_kernel_stats.txt; head -12 $O/r02a_tecogan_bf16_kernel_stats.txt
rm -rf $O/prof_teco_a
