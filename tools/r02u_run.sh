#!/bin/bash
# Round-2 GPU session U (validation): full GPU suite, PMC passes, the default bench line, rocprofv3 kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
( time timeout 1300 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=6 ) > $O/r02u_pytest_gpu.log 2>&1; grep -E "passed|failed" $O/r02u_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/r02u_pytest_gpu.log | cut -c1-200
cd /tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
# PMC passes: one counter set per run, no trace domains beside --kernel-trace
P="$B --steps 2 --warmup 1 --no-graph"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_u_fetch -- $P > $O/pmc_u_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_u_write -- $P > $O/pmc_u_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_u_mfma -- $P > $O/pmc_u_mfma.log 2>&1
I="python $R/tools/bench_infer.py --frames 4 --warmup 2 --no-graph"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_u_ifetch -- $I > $O/pmc_u_ifetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_u_iwrite -- $I > $O/pmc_u_iwrite.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_u_imfma -- $I > $O/pmc_u_imfma.log 2>&1
cd $R
python tools/pmc_summary.py --json $O/r02u_pmc_train.json $O/pmc_u_fetch $O/pmc_u_write $O/pmc_u_mfma > $O/r02u_pmc_train.txt 2>&1; head -14 $O/r02u_pmc_train.txt | cut -c1-200
python tools/pmc_summary.py --json $O/r02u_pmc_infer.json $O/pmc_u_ifetch $O/pmc_u_iwrite $O/pmc_u_imfma > $O/r02u_pmc_infer.txt 2>&1; head -8 $O/r02u_pmc_infer.txt | cut -c1-200
cp $O/r02u_pmc_train.json $O/r02u_pmc_infer.json $R/profiles/ 2>/dev/null
rm -rf $O/pmc_u_*
( time timeout 700 python bench.py ) > $O/r02u_bench.json 2> $O/r02u_bench.err; cut -c1-600 $O/r02u_bench.json; tail -3 $O/r02u_bench.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_u_teco -o teco -- $B --steps 20 --warmup 3 > $O/prof_u_teco.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_u_frvsr -o frvsr -- $B --steps 40 --warmup 3 --config frvsr > $O/prof_u_frvsr.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_u_inf -o inf -- python $R/tools/bench_infer.py > $O/prof_u_inf.log 2>&1
for n in teco:tecogan frvsr:frvsr inf:infer1080p; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_u_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r02u_${f}_bf16_kernel_stats.txt; rm -rf $O/prof_u_$d; done
head -8 $O/r02u_tecogan_bf16_kernel_stats.txt | cut -c1-160
