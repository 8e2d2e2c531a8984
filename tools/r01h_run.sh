#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for pf in 1 4; do for b in 256 512; do TG_WGRAD_PF=$pf TG_WGRAD_BLOCKS=$b python tools/trace_wgrad.py 2>&1 | grep -v amdgpu.ids; done; done
TG_WGRAD_PF=1 TG_WGRAD_BLOCKS=1024 python tools/trace_wgrad.py 40 128 128 64 64 2>&1 | grep -v amdgpu.ids | head -14
