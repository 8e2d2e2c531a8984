#!/bin/bash
# Round-3 session U: FNet's flow gradient padded to 8 channels + its 14 weight gradients as one multi-geometry launch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "multi_geometry" 2>&1 | grep -E "passed|failed|Error|assert" | head -6 | tee $O/r03u_pytest.txt
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -k "test_frvsr_step_fp32_parity or test_tecogan_step_fp32_parity or test_frvsr_step_bf16_error_is_bounded or test_frvsr_two_steps_graph_replay" --deselect tests/test_train_gpu.py::test_frvsr_step_fp32_parity_at_baseline_config_C2 --deselect tests/test_train_gpu.py::test_tecogan_step_fp32_parity_at_baseline_config_C3 2>&1 | grep -E "passed|failed|Error|assert" | head -5 | tee -a $O/r03u_pytest.txt
for v in "TG_FNET_WGRAD_MULTI=0" "TG_FNET_WGRAD_MULTI=1" "MB_N=36 TG_FNET_WGRAD_MULTI=0" "MB_N=36 TG_FNET_WGRAD_MULTI=1"; do env $v timeout 60 python tools/mb_fnet.py 2>&1 | grep "^FNet" | tee -a $O/r03u_mb_fnet.txt; done
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in 0 1 0 1; do
  echo "== tecogan TG_FNET_WGRAD_MULTI=$v" | tee -a $O/r03u_ab.txt; TG_FNET_WGRAD_MULTI=$v timeout 120 $B --steps 120 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03u_ab.txt
  echo "== frvsr TG_FNET_WGRAD_MULTI=$v" | tee -a $O/r03u_ab.txt; TG_FNET_WGRAD_MULTI=$v timeout 120 $B --config frvsr --steps 300 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03u_ab.txt
done
