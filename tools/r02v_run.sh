#!/bin/bash
# Round-2 GPU session V: bicubic epilogue with chunk-major global accesses; CPU-baseline sampling; final bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_infer_gpu.py -m gpu -q -s -k "bicubic or inference" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | cut -c1-300 | tee $O/r02v_pytest.txt
for v in "" "TG_NO_BICUBIC_QUAD=1"; do echo "== infer $v" | tee -a $O/r02v_ab.txt; env $v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r02v_ab.txt; done
timeout 100 python tools/microbench.py --only "bicubic" 2>&1 | tail -3 | tee -a $O/r02v_ab.txt
( time timeout 700 python bench.py ) > $O/r02v_bench.json 2> $O/r02v_bench.err; cut -c1-300 $O/r02v_bench.json; tail -3 $O/r02v_bench.err
cd /tmp; timeout 100 rocprofv3 --kernel-trace --stats -d $O/prof_v_inf -o inf -- python $R/tools/bench_infer.py > $O/prof_v_inf.log 2>&1
db=$(find $O/prof_v_inf -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r02v_infer1080p_bf16_kernel_stats.txt; rm -rf $O/prof_v_inf; grep -E "bicubic|warp_s2d" $O/r02v_infer1080p_bf16_kernel_stats.txt | cut -c1-120
