#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "hr_tail" 2>&1 | grep -E "passed|failed|Error|assert" | head | tee $O/r03m_pytest.txt
for v in "TG_HR_TAIL=0" "TG_HR_TAIL=1"; do echo "== infer $v" | tee -a $O/r03m_ab.txt; env $v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r03m_ab.txt; done
timeout 100 python -m pytest tests/test_infer_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert" | head | tee -a $O/r03m_pytest.txt
