#!/bin/bash
# Round-4 GPU session A (prepared at the end of round 3, after the GPU budget was spent):
#  1. the de-duplicated VGG target pass (TG_VGGT_DEDUP=1: 40 instead of 76 target images per configs[2] step): parity test,
#     then the step A/B, alternating on one box; enable by default if it wins;
#  2. TG_OVERLAP_PARTS=239 (bit 128: the late VGG pass starts beside the loss / D-fake-pass segment `fwd_c`): step A/B + timeline;
#  3. the training chain's transposed convs on deconv3x3s2_ws instead of conv_igemm (TG_DECONV_WS_MIN_TILES=1): isolated + step A/B
#     (a parity run of tests/test_kernels_gpu.py -k deconv under that setting first if it wins);
#  4. segment timelines of both settings (does `fwd_a` shorten when the side stream carries less beside it?).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 200 python -m pytest -q -x tests/test_train_gpu.py::test_tecogan_step_with_deduplicated_vgg_target_pass tests/test_train_gpu.py::test_tecogan_step_fp32_parity 2>&1 | tail -3
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for d in 0 1 0 1 0 1; do echo "== tecogan TG_VGGT_DEDUP=$d"; TG_VGGT_DEDUP=$d timeout 120 $B 2>/dev/null | ms; done
for v in 111 239 111 239; do echo "== tecogan TG_OVERLAP_PARTS=$v"; TG_OVERLAP_PARTS=$v timeout 120 $B 2>/dev/null | ms; done
echo "== timeline TG_OVERLAP_PARTS=239"; TG_OVERLAP_PARTS=239 timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -17
echo "== deconv kernels at the training shapes"; timeout 60 python tools/mb_deconv.py 2>&1 | grep deconv; TG_DECONV_WS_MIN_TILES=1 timeout 60 python tools/mb_deconv.py 2>&1 | grep deconv
for m in 256 1 256 1; do echo "== tecogan TG_DECONV_WS_MIN_TILES=$m"; TG_DECONV_WS_MIN_TILES=$m timeout 120 $B 2>/dev/null | ms; done
for d in 0 1; do echo "== timeline TG_VGGT_DEDUP=$d"; TG_VGGT_DEDUP=$d timeout 100 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL" | head -16; done
} > $O/r04a_ab.txt 2>&1
cat $O/r04a_ab.txt
