#!/bin/bash
# Round-4 GPU session I: forward HR tail of the training recurrence as latency-regime launches (csrc/hr_fwd_lat.hip): parity,
# microbench, step A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 300 python -m pytest -q tests/test_kernels_gpu.py -k "deconv_latency or hr_tail_training or pack_weights_frag" 2>&1 | tail -12
timeout 300 python -m pytest -q tests/test_train_gpu.py -k "one_launch_residual or frvsr_step_bf16 or bf16_mode_error_at_baseline_config_C2" 2>&1 | tail -4
echo "== microbench"; timeout 120 python tools/mb_hr_fwd.py 2>&1 | grep "forward HR"
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1 0; do
  echo "== tecogan TG_HR_FWD_LAT=$m"; TG_HR_FWD_LAT=$m timeout 120 $B 2>/dev/null | ms
  echo "== frvsr TG_HR_FWD_LAT=$m"; TG_HR_FWD_LAT=$m timeout 120 $B --config frvsr 2>/dev/null | ms
done
echo "== timeline default"; timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -20
} > $O/r04i_ab.txt 2>&1
cat $O/r04i_ab.txt
