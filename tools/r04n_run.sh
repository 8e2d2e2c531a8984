#!/bin/bash
# Round-4 GPU session N: the persistent launch of the fused HR tail for the inference frame: parity tests, microbench against the
# per-tile launch and csrc/hr_tail.hip, and the 1080p inference stream A/B (TG_HR_TAIL_LAT=0: round-3 tail).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "hr_tail" -x 2>&1 | tail -5 > $O/r04n_pytest.txt
timeout 900 python -m pytest tests/test_infer_gpu.py -q -m gpu -x -s 2>&1 | tail -12 >> $O/r04n_pytest.txt
{
echo "== HR tail microbench"
timeout 200 python tools/mb_infer_tail.py 2>&1 | grep "HR tail"
TG_HR_TAIL_PERSIST_MIN=1000000000 timeout 200 python tools/mb_infer_tail.py 2>&1 | grep "HR tail"
echo "== 1080p inference stream"
for v in 1 0 1 0; do echo "TG_HR_TAIL_LAT=$v"; TG_HR_TAIL_LAT=$v timeout 300 python tools/bench_infer.py 2>&1 | tail -2; done
} > $O/r04n_ab.txt 2>&1
cat $O/r04n_pytest.txt $O/r04n_ab.txt
