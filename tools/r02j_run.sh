#!/bin/bash
# Round-2 GPU session J: bicubic quad fix, packed-tile threshold, VGG middle-frame overlap piece (bit 16) A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "bicubic or packed" 2>&1 | tail -4 | cut -c1-250 | tee $O/r02j_pytest.txt
echo "== infer" | tee $O/r02j_ab.txt; timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r02j_ab.txt
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
echo "== frvsr" | tee -a $O/r02j_ab.txt; timeout 120 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02j_ab.txt
for v in "TG_OVERLAP_PARTS=15" "TG_OVERLAP_PARTS=31" "TG_OVERLAP_PARTS=15" "TG_OVERLAP_PARTS=31" "TG_OVERLAP_PARTS=27" "TG_OVERLAP_PARTS=0"; do
  echo "== tecogan $v" | tee -a $O/r02j_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02j_ab.txt
done
TG_OVERLAP_PARTS=31 timeout 200 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "tecogan_step_fp32_parity or three_steps or no_pingpong" 2>&1 | tail -4 | cut -c1-250 | tee -a $O/r02j_pytest.txt
