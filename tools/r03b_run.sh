#!/bin/bash
# Round-3 session B: transpose-read weight-gradient kernel in isolation (single layer and the grouped trunk launch)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in 0 1; do TG_WGRAD_TR=$v timeout 120 python tools/mb_wgrad.py 2>&1 | grep "^wgrad" | tee -a $O/r03b_mb_wgrad.txt; done
TG_SEG_STAMPS=1 timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl" | head -20 | tee $O/r03b_seg_timeline.txt
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "standin or captured" 2>&1 | tail -5 | tee $O/r03b_pytest.txt
