#!/bin/bash
# Round-4 GPU session U: the BPTT cut (TG_BWD_CUT=k): early generator weight gradients (parts 2) / early FNet slice (1) / both (3)
# on the side stream beside the second half of the BPTT.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "bptt_cut" 2>&1 | tail -4 > $O/r04u_pytest.txt
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
{
for kp in "0 3" "9 2" "9 3" "12 2" "6 2" "0 3" "9 2" "4 2"; do set -- $kp; echo "== tecogan TG_BWD_CUT=$1 TG_BWD_CUT_PARTS=$2"; TG_BWD_CUT=$1 TG_BWD_CUT_PARTS=$2 timeout 300 $B 2>&1 | tail -1 | ms; done
for kp in "0 3" "4 2" "6 2" "4 3" "0 3" "2 2"; do set -- $kp; echo "== frvsr TG_BWD_CUT=$1 TG_BWD_CUT_PARTS=$2"; TG_BWD_CUT=$1 TG_BWD_CUT_PARTS=$2 timeout 300 $B --config frvsr 2>&1 | tail -1 | ms; done
echo "== timeline TG_BWD_CUT=9 parts 2"; TG_BWD_CUT=9 TG_BWD_CUT_PARTS=2 timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -22
} > $O/r04u_ab.txt 2>&1
cat $O/r04u_pytest.txt $O/r04u_ab.txt
