#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad or conv_bwd" 2>&1 | tail -2
python tools/mb_wgrad.py 2>&1 | grep "wgrad" | sed "s/^/row3 /"
TG_NO_WGRAD_ROW3=1 python tools/mb_wgrad.py 2>&1 | grep "wgrad" | sed "s/^/tap  /"
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-160
TG_NO_WGRAD_ROW3=1 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-160 | sed "s/^/tap /"
python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-160
TG_NO_WGRAD_ROW3=1 python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-160 | sed "s/^/tap /"
