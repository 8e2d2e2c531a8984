#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k wgrad 2>&1 | tail -2
for a in 0 1 16 28; do for b in 256 512; do TG_WGRAD_ABL=$a TG_WGRAD_BLOCKS=$b python tools/mb_wgrad.py 2>&1 | grep "gen40" | sed "s/^/abl=$a /"; done; done
for b in 128 256 512 1024; do TG_WGRAD_BLOCKS=$b python tools/mb_wgrad.py 2>&1 | grep wgrad; done
