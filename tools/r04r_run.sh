#!/bin/bash
# Round-4 GPU session R: inference frame lookahead (next frame's FNet on a side stream) + weight stream first in the fragment-order
# prologue of conv3x3_ws: parity, kernel microbench, 1080p stream A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "fragment_order" -x 2>&1 | tail -3 > $O/r04r_pytest.txt
timeout 1200 python -m pytest tests/test_infer_gpu.py -q -m gpu -x -s 2>&1 | tail -12 >> $O/r04r_pytest.txt
{
echo "== kernel"
timeout 200 python tools/mb_ws.py 2>&1 | grep conv
echo "== 1080p inference stream: lookahead, none, lookahead, none"
for v in "" --no-lookahead "" --no-lookahead; do timeout 300 python tools/bench_infer.py $v 2>&1 | tail -1; done
echo "== 144x180 (calendar size), lookahead / none"
for v in "" --no-lookahead; do timeout 300 python tools/bench_infer.py --h 144 --w 180 --frames 200 $v 2>&1 | tail -1; done
} > $O/r04r_ab.txt 2>&1
cat $O/r04r_pytest.txt $O/r04r_ab.txt
