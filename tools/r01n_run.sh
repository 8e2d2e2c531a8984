#!/bin/bash
# End-of-round GPU session (after the epilogue changes): full GPU suite, bench lines, rocprofv3 kernel stats.
# (PMC traffic passes: tools/r01m_run.sh; the memory behaviour of the roofline kernel did not change since.)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 200 python -m pytest tests -m gpu -x -q > $O/r01n_pytest_gpu.log 2>&1; tail -1 $O/r01n_pytest_gpu.log
timeout 120 python bench.py > $O/r01n_bench_frvsr_bf16.json 2> $O/r01n_bench_frvsr.err; cut -c1-200 $O/r01n_bench_frvsr_bf16.json
timeout 120 python bench.py --config tecogan --steps 20 --cpu-seconds 10 > $O/r01n_bench_tecogan_bf16.json 2> $O/r01n_bench_teco.err; cut -c1-200 $O/r01n_bench_tecogan_bf16.json
timeout 60 python tools/bench_infer.py 2>/dev/null | tail -1 | tee $O/r01n_bench_infer.json
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats -d $O/prof_frvsr13 -o frvsr -- python $R/bench.py --no-cpu-baseline > $O/prof_frvsr13.log 2>&1
timeout 100 rocprofv3 --kernel-trace --stats -d $O/prof_teco10 -o teco -- python $R/bench.py --config tecogan --steps 10 --no-cpu-baseline > $O/prof_teco10.log 2>&1
timeout 100 rocprofv3 --kernel-trace --stats -d $O/prof_inf7 -o inf -- python $R/tools/bench_infer.py > $O/prof_inf7.log 2>&1
for n in frvsr13:frvsr teco10:tecogan inf7:infer1080p; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r01n_${f}_bf16_kernel_stats.txt; done
head -6 $O/r01n_frvsr_bf16_kernel_stats.txt
