#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 20 python -m pytest tests/test_kernels_gpu.py -x -q -k "grouped" 2>&1 | tail -2
TG_WGRAD_GROUPED=1 timeout 20 python bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | cut -c1-150 | sed "s/^/grouped /"
TG_WGRAD_GROUPED=0 timeout 20 python bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | cut -c1-150 | sed "s/^/single  /"
