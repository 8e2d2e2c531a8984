#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
python tools/mb_floor.py 2>&1 | grep -v amdgpu.ids
for n in c8 c8vgg; do python tools/mb_conv.py $n 2>&1 | grep force; TG_NO_C8=1 python tools/mb_conv.py $n 2>&1 | grep force | sed "s/^/no-c8 /"; done
python tools/mb_wgrad.py 2>&1 | grep wgrad
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
TG_NO_C8=1 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200 | sed "s/^/no-c8 /"
python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>/dev/null | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_frvsr9 -o frvsr -- python $R/bench.py --no-cpu-baseline > $O/prof_frvsr9.log 2>&1
db=$(find $O/prof_frvsr9 -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r01e_frvsr_kernel_stats.txt; head -30 $O/r01e_frvsr_kernel_stats.txt
