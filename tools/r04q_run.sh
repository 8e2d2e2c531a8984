#!/bin/bash
# Round-4 GPU session Q: conv3x3_ws with fragment-order weights (whole-line weight prologue) at one / two workgroups per CU: parity,
# kernel A/B, cycle trace, 1080p inference stream A/B (TG_C3WS_FRAG=0: round-3 path).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "fragment_order or weights_in_registers" -x 2>&1 | tail -4 > $O/r04q_pytest.txt
timeout 900 python -m pytest tests/test_infer_gpu.py -q -m gpu -x -k "bf16 or graph" 2>&1 | tail -3 >> $O/r04q_pytest.txt
{
echo "== kernel"
for c in 2 1; do TG_C3WS_FRAG_PER_CU=$c timeout 200 python tools/mb_ws.py 2>&1 | grep conv; done
echo "== 1080p inference stream (TG_C3WS_FRAG = 1, 0, 1, 0; then per-CU 1)"
for v in 1 0 1 0; do TG_C3WS_FRAG=$v timeout 300 python tools/bench_infer.py 2>&1 | tail -1; done
TG_C3WS_FRAG_PER_CU=1 timeout 300 python tools/bench_infer.py 2>&1 | tail -1
} > $O/r04q_ab.txt 2>&1
cat $O/r04q_pytest.txt $O/r04q_ab.txt
