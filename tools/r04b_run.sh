#!/bin/bash
# Round-4 GPU session B: load-schedule variants of the one-launch residual block (cycle stamps + chain timing), parity of the
# new default, step A/B; the bf16-vs-fp32 error table (where does the timed mode's gradient error come from?); new host tests.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 300 python -m pytest -q -x tests/test_kernels_gpu.py -k "resblock" 2>&1 | tail -3
echo "== trace / variants"; timeout 200 python tools/trace_rb.py 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|amdgpu.ids"
echo "== microbench (product default)"; timeout 120 python tools/mb_resblock.py 2>&1 | grep "res block"
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1; do
  echo "== tecogan TG_RESBLOCK_LAT=$m"; TG_RESBLOCK_LAT=$m timeout 120 $B 2>/dev/null | ms
  echo "== frvsr TG_RESBLOCK_LAT=$m"; TG_RESBLOCK_LAT=$m timeout 120 $B --config frvsr 2>/dev/null | ms
done
timeout 400 python -m pytest -q -x tests/test_train_gpu.py -k "validation_pass or bench_gpus_2 or one_launch_residual or deduplicated" 2>&1 | tail -4
} > $O/r04b_ab.txt 2>&1
{ timeout 400 python tools/bf16_error_table.py 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|amdgpu.ids"; } > $O/r04b_bf16_error_table.txt 2>&1
cat $O/r04b_ab.txt; cat $O/r04b_bf16_error_table.txt
