#!/bin/bash
# Round-4 GPU session L: two residual blocks per launch (csrc/resblock2_lat.hip): bit identity, microbench, step A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 300 python -m pytest -q -x tests/test_kernels_gpu.py -k "two_residual_blocks or resblock_one_launch" 2>&1 | tail -6
timeout 300 python -m pytest -q tests/test_train_gpu.py -k "one_launch_residual or frvsr_step_bf16 or bf16_mode_error_at_baseline_config_C2 or frvsr_two_steps" 2>&1 | tail -4
echo "== microbench"; timeout 200 python tools/mb_resblock.py 2>&1 | grep "res block"
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1 0; do
  echo "== tecogan TG_RESBLOCK_PAIR=$m"; TG_RESBLOCK_PAIR=$m timeout 120 $B 2>/dev/null | ms
  echo "== frvsr TG_RESBLOCK_PAIR=$m"; TG_RESBLOCK_PAIR=$m timeout 120 $B --config frvsr 2>/dev/null | ms
done
echo "== timeline default"; timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -20
} > $O/r04l_ab.txt 2>&1
cat $O/r04l_ab.txt
