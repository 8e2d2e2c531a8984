#!/bin/bash
# Round-4 GPU session J: the BPTT's conv_tran1 input gradient on the latency kernel; parity; step A/B of the round's latency kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 300 python -m pytest -q tests/test_kernels_gpu.py -k "deconv_latency or hr_tail" 2>&1 | tail -4
timeout 300 python -m pytest -q tests/test_train_gpu.py -k "one_launch_residual or frvsr_step_bf16 or bf16_mode_error_at_baseline_config_C2 or frvsr_two_steps" 2>&1 | tail -4
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1 0; do
  echo "== tecogan TG_HR_BWD_LAT=$m"; TG_HR_BWD_LAT=$m timeout 120 $B 2>/dev/null | ms
  echo "== frvsr TG_HR_BWD_LAT=$m"; TG_HR_BWD_LAT=$m timeout 120 $B --config frvsr 2>/dev/null | ms
done
echo "== all latency kernels off (round-3 chain)"; TG_RESBLOCK_LAT=0 timeout 120 $B 2>/dev/null | ms; TG_RESBLOCK_LAT=0 timeout 120 $B --config frvsr 2>/dev/null | ms
echo "== timeline default"; timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -20
} > $O/r04j_ab.txt 2>&1
cat $O/r04j_ab.txt
