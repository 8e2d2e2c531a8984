#!/bin/bash
# Round-2 GPU session L: fixed test cases; per-dispatch timeline of the overlapped TecoGAN step (critical-path analysis).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "wide_layer or packed" 2>&1 | tail -4 | cut -c1-300 | tee $O/r02l_pytest.txt
cd /tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_l -- $B --steps 6 --warmup 3 > $O/tl_l.log 2>&1
python $R/tools/timeline.py $O/tl_l $O/r02l_timeline.csv --last 13000; rm -rf $O/tl_l
TG_OVERLAP_PARTS=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_l0 -- $B --steps 6 --warmup 3 > $O/tl_l0.log 2>&1
python $R/tools/timeline.py $O/tl_l0 $O/r02l_timeline_serial.csv --last 13000; rm -rf $O/tl_l0
