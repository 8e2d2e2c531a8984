#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
python tools/mb_wgrad.py 2>&1 | grep wgrad
for b in 256 1024; do TG_WGRAD_BLOCKS=$b python tools/mb_wgrad.py 2>&1 | grep wgrad; done
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-200
