#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
for pf in 1 4; do for b in 256 512 1024; do TG_WGRAD_PF=$pf TG_WGRAD_BLOCKS=$b python tools/mb_wgrad.py 2>&1 | grep wgrad | sed "s/^/pf=$pf /"; done; done
python tools/mb_conv.py c8 2>&1 | grep force
for f in 16,64 8,64 4,64; do TG_C3_FORCE=$f python tools/mb_conv.py c8 2>&1 | grep force; done
for a in 1 2 3 4 12; do TG_C3_FORCE=16,64 TG_C3_ABL=$a python tools/mb_conv.py c8 2>&1 | grep force | sed "s/^/abl=$a /"; done
TG_NO_CONV3X3=1 python tools/mb_conv.py c8 2>&1 | grep force | sed "s/^/generic /"
for n in gen inf vgg1 vgg3 fnet out; do python tools/mb_conv.py $n 2>&1 | grep force; done
python tools/microbench.py 2>&1 | tail -25
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-260
python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-260
python tools/bench_infer.py 2>&1 | tail -4
