#!/bin/bash
# Round-3 session T: ordered-reduction parity mode (TG_DETERMINISTIC=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{ echo "== small config, default mode"; timeout 300 python tools/c3_repeat.py --config small --runs 4 2>&1 | grep -E "^mode|^run|IDENT|DIFF|Error"
  echo "== small config, TG_DETERMINISTIC=1"; TG_DETERMINISTIC=1 timeout 300 python tools/c3_repeat.py --config small --runs 4 2>&1 | grep -E "^mode|^run|IDENT|DIFF|Error"
  echo "== C3 (configs[2]), TG_DETERMINISTIC=1, 3 runs"; TG_DETERMINISTIC=1 timeout 600 python tools/c3_repeat.py --config c3 --runs 3 2>&1 | grep -E "^mode|^run|IDENT|DIFF|Error"
  echo "== C3 (configs[2]), default mode, 3 runs"; timeout 300 python tools/c3_repeat.py --config c3 --runs 3 2>&1 | grep -E "^mode|^run|IDENT|DIFF|Error"
} | tee $O/r03t_repeat.txt
TG_DETERMINISTIC=1 timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -k "test_tecogan_step_fp32_parity or test_frvsr_step_fp32_parity or test_deterministic" --deselect tests/test_train_gpu.py::test_frvsr_step_fp32_parity_at_baseline_config_C2 --deselect tests/test_train_gpu.py::test_tecogan_step_fp32_parity_at_baseline_config_C3 2>&1 | grep -E "passed|failed|Error|assert" | head -5 | tee $O/r03t_pytest.txt
