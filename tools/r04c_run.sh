#!/bin/bash
# Round-4 GPU session C: fragment-order weights + unified weight-stream pipeline of the one-launch residual block: parity,
# prefetch-distance sweep with cycle stamps, step A/B; the failing host test of session B with its message.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 300 python -m pytest -q -x tests/test_kernels_gpu.py -k "resblock or pack_weights" 2>&1 | tail -3
echo "== trace / sweep"; timeout 300 python tools/trace_rb.py 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|amdgpu.ids"
echo "== microbench (product default)"; timeout 120 python tools/mb_resblock.py 2>&1 | grep "res block"
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1; do
  echo "== tecogan TG_RESBLOCK_LAT=$m"; TG_RESBLOCK_LAT=$m timeout 120 $B 2>/dev/null | ms
  echo "== frvsr TG_RESBLOCK_LAT=$m"; TG_RESBLOCK_LAT=$m timeout 120 $B --config frvsr 2>/dev/null | ms
done
timeout 400 python -m pytest -q tests/test_train_gpu.py -k "validation_pass or bench_gpus_2 or one_launch_residual or frvsr_step_bf16 or tecogan_three_steps" 2>&1 | tail -25
echo "== timeline default"; timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -17
} > $O/r04c_ab.txt 2>&1
cat $O/r04c_ab.txt
