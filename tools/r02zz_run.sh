#!/bin/bash
# Round-2 GPU session ZZ: LDS-staged weight prologue on by default: kernel + inference + API parity, inference rate.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_infer_gpu.py tests/test_api_gpu.py -m gpu -q -s -k "weights_in_registers or inference or api or fused or deconv" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | cut -c1-300 | tee $O/r02zz_pytest.txt
timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee $O/r02zz_infer.txt
