#!/bin/bash
# Round-2 GPU session Z: spread of the D weight-gradient error at C3 under different schedules; LDS-staged weight prologue of
# the weights-in-registers kernel (opt-in) parity + inference A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
TG_C3WS_WLDS=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "weights_in_registers" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 | cut -c1-300 | tee $O/r02z_pytest.txt
for v in "TG_C3WS_WLDS=1" "TG_C3WS_WLDS=0"; do echo "== infer $v" | tee -a $O/r02z_ab.txt; env $v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r02z_ab.txt; done
timeout 400 python tools/c3_repeat.py 2>&1 | grep -E "oracle|worst" | tee $O/r02z_c3_repeat.txt
