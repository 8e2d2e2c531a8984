#!/bin/bash
# Round-6 final validation on the end state: full GPU suite, the default bench line, rocprofv3 kernel stats, PMC passes, segment timeline.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
T=${1:-z}
[ -n "$SKIP_SUITE" ] || ( time timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=6 ) > $O/r06${T}_pytest_gpu.log 2>&1; grep -E "passed|failed" $O/r06${T}_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/r06${T}_pytest_gpu.log | cut -c1-200
( time timeout 700 python bench.py ) > $O/r06${T}_bench.json 2> $O/r06${T}_bench.err; cut -c1-400 $O/r06${T}_bench.json; tail -3 $O/r06${T}_bench.err
cd /tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_z_teco -o teco -- $B --steps 20 --warmup 3 > $O/prof_z_teco.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_z_frvsr -o frvsr -- $B --steps 40 --warmup 3 --config frvsr > $O/prof_z_frvsr.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_z_inf -o inf -- python $R/tools/bench_infer.py --window 16 --warmup 40 --frames 80 > $O/prof_z_inf.log 2>&1
for n in teco:tecogan frvsr:frvsr inf:infer1080p; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_z_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r06_${f}_bf16_kernel_stats.txt 60; rm -rf $O/prof_z_$d; done
head -12 $O/r06_tecogan_bf16_kernel_stats.txt | cut -c1-160
P="$B --steps 2 --warmup 1 --no-graph"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_z_fetch -- $P > $O/pmc_z_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_z_write -- $P > $O/pmc_z_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_z_mfma -- $P > $O/pmc_z_mfma.log 2>&1
I="python $R/tools/bench_infer.py --frames 4 --warmup 2 --no-graph --no-lookahead"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_z_ifetch -- $I > $O/pmc_z_ifetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_z_iwrite -- $I > $O/pmc_z_iwrite.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_z_imfma -- $I > $O/pmc_z_imfma.log 2>&1
cd $R
python tools/pmc_summary.py --json $O/r06_pmc_train.json $O/pmc_z_fetch $O/pmc_z_write $O/pmc_z_mfma > $O/r06_pmc_train.txt 2>&1; head -16 $O/r06_pmc_train.txt | cut -c1-200
python tools/pmc_summary.py --json $O/r06_pmc_infer.json $O/pmc_z_ifetch $O/pmc_z_iwrite $O/pmc_z_imfma > $O/r06_pmc_infer.txt 2>&1; head -8 $O/r06_pmc_infer.txt | cut -c1-200
rm -rf $O/pmc_z_*/
timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -24 > $O/r06${T}_seg_timeline.txt
cat $O/r06${T}_seg_timeline.txt
