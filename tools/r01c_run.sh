#!/bin/bash
# One GPU-box session: final bench lines, rocprofv3 kernel stats of the same commands, PMC traffic passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/r01c_bench_frvsr_bf16.json 2> $O/r01c_bench_frvsr.err
python $R/bench.py --config tecogan --steps 20 > $O/r01c_bench_tecogan_bf16.json 2> $O/r01c_bench_teco.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_frvsr8 -o frvsr -- python $R/bench.py --no-cpu-baseline > $O/prof_frvsr8.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_teco6 -o teco -- python $R/bench.py --config tecogan --steps 10 --no-cpu-baseline > $O/prof_teco6.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/pmc_conv.py > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/pmc_conv.py > $O/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/r01c_pmc_summary.txt 2>&1
for n in frvsr8:frvsr teco6:teco; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r01c_${f}_kernel_stats.txt; done
cat $O/r01c_bench_frvsr_bf16.json $O/r01c_bench_tecogan_bf16.json; cat $O/r01c_pmc_summary.txt | grep -i "conv3x3\|lincomb"; head -12 $O/r01c_frvsr_kernel_stats.txt
