#!/bin/bash
# Round-4 GPU session K: where do D's own-gradient passes (`down`) and the lookahead target pass go now that the main stream
# idles ~1.1 ms before the BPTT?  TG_OVERLAP_PARTS 111 (down on the side stream beside the BPTT) vs 103 (down on the main
# stream, in the gap), with and without the target lookahead.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for rep in 1 2; do for p in 111 103; do for l in 1 0; do
  echo "== tecogan TG_OVERLAP_PARTS=$p TG_TARGET_LOOKAHEAD=$l"; TG_OVERLAP_PARTS=$p TG_TARGET_LOOKAHEAD=$l timeout 120 $B 2>/dev/null | ms
done; done; done
echo "== timeline TG_OVERLAP_PARTS=103 lookahead 1"; TG_OVERLAP_PARTS=103 timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -20
echo "== timeline TG_OVERLAP_PARTS=103 lookahead 0"; TG_OVERLAP_PARTS=103 TG_TARGET_LOOKAHEAD=0 timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -20
} > $O/r04k_ab.txt 2>&1
cat $O/r04k_ab.txt
