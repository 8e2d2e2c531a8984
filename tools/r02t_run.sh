#!/bin/bash
# Round-2 GPU session T: rotated weight-panel fetch order in the wide-layer DMA kernel: parity, trace, microbench A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "wide_layer" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | cut -c1-300 | tee $O/r02t_pytest.txt
TG_C3DMA_PAIR=1 timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "wide_layer" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | cut -c1-300 | tee -a $O/r02t_pytest.txt
timeout 100 python tools/trace_dma.py 2>&1 | grep -E "==|stage  [2-4] " | tee $O/r02t_trace_dma.txt
for v in "" "TG_C3DMA_ROT=0" "TG_C3DMA_PAIR=1"; do echo "== microbench $v" | tee -a $O/r02t_microbench.txt; env $v timeout 200 python tools/microbench.py --only "conv3x3 wide" 2>&1 | tail -8 | tee -a $O/r02t_microbench.txt; env $v timeout 100 python tools/microbench.py --only "conv3x3 vgg " 2>&1 | tail -2 | tee -a $O/r02t_microbench.txt; done
