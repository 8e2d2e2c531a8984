#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_infer_gpu.py -x -q 2>&1 | tail -2
for n in inf vgg1 vgg3; do python tools/mb_conv.py $n 2>&1 | grep force; done
python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-170
python tools/bench_infer.py 2>/dev/null | tail -1
