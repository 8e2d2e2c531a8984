#!/bin/bash
# Round-3 GPU session Y: row-band warp_s2d forward kernel and the merged scatter of warp_s2d backward: parity tests,
# isolated timings, step A/B of the scatter merge, inference frame.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "warp" 2>&1 | tail -5
timeout 200 python -m pytest -q -x tests/test_train_gpu.py::test_frvsr_step_fp32_parity tests/test_train_gpu.py::test_tecogan_step_fp32_parity tests/test_train_gpu.py::test_tecogan_no_pingpong_backward_flow_branch tests/test_train_gpu.py::test_frvsr_step_bf16_error_is_bounded "tests/test_infer_gpu.py::test_inference_fp32_parity" tests/test_infer_gpu.py::test_inference_bf16_bounded 2>&1 | tail -3
echo "== microbench"; timeout 100 python tools/mb_warp.py 2>&1 | grep warp_s2d
echo "== microbench merge off"; TG_WARP_BWD_MERGE=0 timeout 100 python tools/mb_warp.py 2>&1 | grep warp_s2d_bwd
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for m in 1 0 1 0; do
  echo "== tecogan TG_WARP_BWD_MERGE=$m"; TG_WARP_BWD_MERGE=$m timeout 120 $B 2>/dev/null | ms
  echo "== frvsr TG_WARP_BWD_MERGE=$m"; TG_WARP_BWD_MERGE=$m timeout 120 $B --config frvsr 2>/dev/null | ms
done
echo "== inference"; timeout 120 python tools/bench_infer.py 2>/dev/null | cut -c1-200
timeout 120 python tools/bench_infer.py 2>/dev/null | cut -c1-200
} > $O/r03y_ab.txt 2>&1
cat $O/r03y_ab.txt
