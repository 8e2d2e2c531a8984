#!/bin/bash
# Round-4 GPU session F: the forward recurrence in chunks, each chunk's VGG pass as soon as its frames exist (TG_VGG_CUTS),
# with and without the target lookahead; parity of the new schedule; timeline of the default.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 400 python -m pytest -q tests/test_train_gpu.py -k "lookahead or three_steps or tecogan_step_fp32_parity or no_pingpong or temporal_only" --deselect tests/test_train_gpu.py::test_tecogan_step_fp32_parity_at_baseline_config_C3 2>&1 | tail -4
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for c in "11" "5,10,15" "6,12" "4,8,12,16" "7,14" "5,10,15" "11" "6,11,15" "8,14"; do
  for l in 1 0; do echo "== tecogan TG_VGG_CUTS=$c TG_TARGET_LOOKAHEAD=$l"; TG_VGG_CUTS=$c TG_TARGET_LOOKAHEAD=$l timeout 120 $B 2>/dev/null | ms; done
done
echo "== timeline default"; timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -26
} > $O/r04f_ab.txt 2>&1
cat $O/r04f_ab.txt
