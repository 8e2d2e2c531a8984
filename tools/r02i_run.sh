#!/bin/bash
# Round-2 GPU session I: deconv ws kernel, inference A/B, FRVSR regression check.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "deconv3x3s2 or bicubic or weights_in_registers or packed" 2>&1 | tail -6 | cut -c1-250 | tee $O/r02i_pytest.txt
for v in "" "TG_NO_DECONV_WS=1" "TG_NO_BICUBIC_QUAD=1"; do echo "== infer $v" | tee -a $O/r02i_ab.txt; env $v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r02i_ab.txt; done
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline --config frvsr"
for v in "" "TG_C3_PRIO=0" "TG_NO_C3_PACK=1" "TG_C3_PRIO=0 TG_NO_C3_PACK=1"; do
  echo "== frvsr $v" | tee -a $O/r02i_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" | tee -a $O/r02i_ab.txt
done
cd /tmp; timeout 100 rocprofv3 --kernel-trace --stats -d $O/prof_i_inf -o inf -- python $R/tools/bench_infer.py > $O/prof_i_inf.log 2>&1
db=$(find $O/prof_i_inf -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r02i_infer1080p_bf16_kernel_stats.txt; rm -rf $O/prof_i_inf; head -14 $O/r02i_infer1080p_bf16_kernel_stats.txt | cut -c1-140
