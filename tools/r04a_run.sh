#!/bin/bash
# Round-4 GPU session A: the one-launch residual block (csrc/resblock_lat.hip): parity (bit identity against the two-launch
# path), microbench, cycle trace, step A/B; then the three switches round 3 left unmeasured.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 300 python -m pytest -q -x tests/test_kernels_gpu.py -k "resblock" 2>&1 | tail -5
timeout 300 python -m pytest -q -x tests/test_train_gpu.py -k "one_launch_residual or frvsr_step_bf16 or bf16_mode_error or frvsr_two_steps or tecogan_three_steps" 2>&1 | tail -5
echo "== microbench"; timeout 120 python tools/mb_resblock.py 2>&1 | grep "res block"
echo "== trace"; timeout 60 python tools/trace_rb.py 2>&1 | grep -v "^ROCm\|^HIP\|^Host" | tail -12
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1 0; do
  echo "== tecogan TG_RESBLOCK_LAT=$m"; TG_RESBLOCK_LAT=$m timeout 120 $B 2>/dev/null | ms
  echo "== frvsr TG_RESBLOCK_LAT=$m"; TG_RESBLOCK_LAT=$m timeout 120 $B --config frvsr 2>/dev/null | ms
done
for d in 1 0 1; do echo "== tecogan TG_VGGT_DEDUP=$d"; TG_VGGT_DEDUP=$d timeout 120 $B 2>/dev/null | ms; done
for v in 239 111 239; do echo "== tecogan TG_OVERLAP_PARTS=$v"; TG_OVERLAP_PARTS=$v timeout 120 $B 2>/dev/null | ms; done
for m in 1 256 1; do echo "== tecogan TG_DECONV_WS_MIN_TILES=$m"; TG_DECONV_WS_MIN_TILES=$m timeout 120 $B 2>/dev/null | ms; done
echo "== deconv kernels at the training shapes"; timeout 60 python tools/mb_deconv.py 2>&1 | grep deconv; TG_DECONV_WS_MIN_TILES=1 timeout 60 python tools/mb_deconv.py 2>&1 | grep deconv
echo "== timeline default"; timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -17
} > $O/r04a_ab.txt 2>&1
cat $O/r04a_ab.txt
