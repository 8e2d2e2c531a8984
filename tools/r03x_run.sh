#!/bin/bash
# Round-3 GPU session X: several early VGG chunks (TG_FWD_CUTS, an experiment switch removed after this session: it lost)
# against the one-cut schedule, alternating on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
{
for c in "" "6,12,15" "" "5,10,15" "7,13,16" "6,11,14,17" "" "6,12,15"; do
  echo "== TG_FWD_CUTS='$c'"; TG_FWD_CUTS=$c timeout 120 $B 2>/dev/null | ms
done
TG_FWD_CUTS=6,12,15 timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -20
} > $O/r03x_ab.txt 2>&1
cat $O/r03x_ab.txt
