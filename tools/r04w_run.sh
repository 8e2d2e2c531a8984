#!/bin/bash
# Round-4 GPU session W: re-tune of two schedule constants on the final kernels: where the one VGG cut goes (TG_VGG_CUTS) and
# whether the generator's weight gradients should still run beside FNet's backward pass (overlap bit 32) in either configuration.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
{
for c in 8 7 9 6 8 7; do echo "== tecogan TG_VGG_CUTS=$c"; TG_VGG_CUTS=$c timeout 300 $B 2>&1 | tail -1 | ms; done
for o in 103 71 103 71; do echo "== tecogan TG_OVERLAP_PARTS=$o"; TG_OVERLAP_PARTS=$o timeout 300 $B 2>&1 | tail -1 | ms; done
for o in 103 71 0 103 71 0; do echo "== frvsr TG_OVERLAP_PARTS=$o"; TG_OVERLAP_PARTS=$o timeout 300 $B --config frvsr 2>&1 | tail -1 | ms; done
echo "== frvsr timeline"; timeout 200 python tools/seg_timeline.py --steps 30 --config frvsr 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -12
} > $O/r04w_ab.txt 2>&1
cat $O/r04w_ab.txt
