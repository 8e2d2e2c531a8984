#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k wgrad 2>&1 | tail -2
for pf in 4; do for b in 256; do TG_WGRAD_PF=$pf TG_WGRAD_BLOCKS=$b python tools/trace_wgrad.py 2>&1 | grep -v amdgpu.ids; done; done
for pf in 1 2 4; do for b in 128 256 512 1024; do TG_WGRAD_PF=$pf TG_WGRAD_BLOCKS=$b python tools/mb_wgrad.py 2>&1 | grep "wgrad" | sed "s/^/pf=$pf /"; done; done
