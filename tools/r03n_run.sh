#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "weights_in_registers or conv_forward or ws" 2>&1 | grep -E "passed|failed|Error|assert" | head -5 | tee $O/r03n_pytest.txt
for v in "TG_C3WS_ROT=0" "TG_C3WS_ROT=1" "TG_C3WS_ROT=0" "TG_C3WS_ROT=1"; do echo "== $v" | tee -a $O/r03n_ab.txt
env $v timeout 100 python tools/microbench.py --only "conv3x3 inf  [1,270,480,64" 2>&1 | grep conv3x3 | tee -a $O/r03n_ab.txt
env $v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r03n_ab.txt; done
