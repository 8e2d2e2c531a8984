#!/bin/bash
# Round-4 GPU session E: target lookahead (the next batch's VGG target features computed during this step's BPTT phase):
# parity test, step A/B against the in-step target pass, with / without the de-duplicated target pass; segment timeline.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 400 python -m pytest -q tests/test_train_gpu.py -k "lookahead or validation_pass or deduplicated or three_steps or tecogan_step_fp32_parity" --deselect tests/test_train_gpu.py::test_tecogan_step_fp32_parity_at_baseline_config_C3 2>&1 | tail -15
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1 0; do echo "== tecogan TG_TARGET_LOOKAHEAD=$m"; TG_TARGET_LOOKAHEAD=$m timeout 120 $B 2>/dev/null | ms; done
for m in 0 1; do echo "== tecogan TG_VGGT_DEDUP=$m (lookahead on)"; TG_VGGT_DEDUP=$m timeout 120 $B 2>/dev/null | ms; done
echo "== timeline (lookahead on)"; timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -20
echo "== timeline (lookahead off)"; TG_TARGET_LOOKAHEAD=0 timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -18
} > $O/r04e_ab.txt 2>&1
cat $O/r04e_ab.txt
