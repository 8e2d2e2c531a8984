#!/bin/bash
# Round-4 GPU session V: conv3x3_ws fragment-order launches with the next tile's DMA instructions between the MFMA groups (ILV)
# against the same library built with -DWS_NO_ILV: parity, kernel microbench, 1080p inference stream.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
OLD=$(python tools/build_variant.py conv3x3_ws.hip -DWS_NO_ILV | tail -1)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "fragment_order or weights_in_registers" -x 2>&1 | tail -3 > $O/r04v_pytest.txt
{
echo "== kernel (interleaved, then -DWS_NO_ILV)"
timeout 200 python tools/mb_ws.py 2>&1 | grep conv
TECOGAN_HIP_LIB=$OLD timeout 200 python tools/mb_ws.py 2>&1 | grep conv
echo "== 1080p inference stream (new, old, new, old)"
for v in "" $OLD "" $OLD; do TECOGAN_HIP_LIB=$v timeout 300 python tools/bench_infer.py 2>&1 | tail -1; done
} > $O/r04v_ab.txt 2>&1
cat $O/r04v_pytest.txt $O/r04v_ab.txt
