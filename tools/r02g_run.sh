#!/bin/bash
# Round-2 GPU session G: full GPU suite, the default bench line, rocprofv3 kernel stats of the three workloads, PMC passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=6 ) > $O/r02g_pytest_gpu.log 2>&1; tail -14 $O/r02g_pytest_gpu.log | cut -c1-300
( time timeout 700 python bench.py ) > $O/r02g_bench.json 2> $O/r02g_bench.err; cut -c1-400 $O/r02g_bench.json; tail -3 $O/r02g_bench.err
cd /tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_g_teco -o teco -- $B --steps 20 --warmup 3 > $O/prof_g_teco.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_g_frvsr -o frvsr -- $B --steps 40 --warmup 3 --config frvsr > $O/prof_g_frvsr.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_g_inf -o inf -- python $R/tools/bench_infer.py > $O/prof_g_inf.log 2>&1
for n in teco:tecogan frvsr:frvsr inf:infer1080p; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_g_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r02g_${f}_bf16_kernel_stats.txt; rm -rf $O/prof_g_$d; done
head -8 $O/r02g_tecogan_bf16_kernel_stats.txt | cut -c1-160
# PMC passes: one counter set per run, no trace domains beside --kernel-trace
P="$B --steps 2 --warmup 1 --no-graph"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_g_fetch -- $P > $O/pmc_g_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_g_write -- $P > $O/pmc_g_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_g_mfma -- $P > $O/pmc_g_mfma.log 2>&1
I="python $R/tools/bench_infer.py --frames 4 --warmup 2 --no-graph"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_g_ifetch -- $I > $O/pmc_g_ifetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_g_iwrite -- $I > $O/pmc_g_iwrite.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_g_imfma -- $I > $O/pmc_g_imfma.log 2>&1
cd $R
python tools/pmc_summary.py --json $O/r02g_pmc_train.json $O/pmc_g_fetch $O/pmc_g_write $O/pmc_g_mfma > $O/r02g_pmc_train.txt 2>&1; head -30 $O/r02g_pmc_train.txt | cut -c1-200
python tools/pmc_summary.py --json $O/r02g_pmc_infer.json $O/pmc_g_ifetch $O/pmc_g_iwrite $O/pmc_g_imfma > $O/r02g_pmc_infer.txt 2>&1; head -12 $O/r02g_pmc_infer.txt | cut -c1-200
du -sh $O/pmc_g_*; find $O/pmc_g_* -name "*.csv" | head; find $O/pmc_g_* -type f ! -name "*counter_collection*" -delete 2>/dev/null
