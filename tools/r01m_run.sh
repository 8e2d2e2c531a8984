#!/bin/bash
# End-of-round GPU session: full GPU suite, smoke, PMC traffic passes, bench lines, rocprofv3 kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r01m_pytest_gpu.log 2>&1; tail -3 $O/r01m_pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch3 -- python $R/tools/pmc_conv.py > $O/pmc_fetch3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write3 -- python $R/tools/pmc_conv.py > $O/pmc_write3.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_fetch3 $O/pmc_write3 --json $R/profiles/pmc_traffic.json > $O/r01m_pmc_summary.txt 2>&1
cp $R/profiles/pmc_traffic.json $O/r01m_pmc_traffic.json; grep -i "conv3x3\|lincomb" $O/r01m_pmc_summary.txt
cd $R
timeout 600 python bench.py > $O/r01m_bench_frvsr_bf16.json 2> $O/r01m_bench_frvsr.err; cat $O/r01m_bench_frvsr_bf16.json
timeout 600 python bench.py --config tecogan --steps 20 > $O/r01m_bench_tecogan_bf16.json 2> $O/r01m_bench_teco.err; cut -c1-400 $O/r01m_bench_tecogan_bf16.json
timeout 600 python bench.py --dtype f32 --no-cpu-baseline > $O/r01m_bench_frvsr_f32.json 2>/dev/null; cut -c1-200 $O/r01m_bench_frvsr_f32.json
timeout 300 python tools/bench_infer.py 2>/dev/null | tail -2 | tee $O/r01m_bench_infer.json
TG_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 10 --warmup 3 2>/dev/null | cut -c1-220
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_frvsr12 -o frvsr -- python $R/bench.py --no-cpu-baseline > $O/prof_frvsr12.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_teco9 -o teco -- python $R/bench.py --config tecogan --steps 10 --no-cpu-baseline > $O/prof_teco9.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_inf6 -o inf -- python $R/tools/bench_infer.py > $O/prof_inf6.log 2>&1
for n in frvsr12:frvsr teco9:tecogan inf6:infer1080p; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r01m_${f}_bf16_kernel_stats.txt; done
head -8 $O/r01m_frvsr_bf16_kernel_stats.txt
