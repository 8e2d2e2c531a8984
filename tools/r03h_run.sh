#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2; do
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "baseline_config" > $O/r03h_pytest_full_$i.txt 2>&1; grep -E "passed|failed|Error|assert|^\[C|^    [0-9]" $O/r03h_pytest_full_$i.txt | head -30 | tee -a $O/r03h_pytest.txt
done
