#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -2
python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-170
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-170
