#!/bin/bash
# Round-4 GPU session O: conv3x3_ws halo pixel pitch 144 -> 160 bytes (conflict-free ds_read_b128 under the gfx950 lane grouping):
# parity, kernel A/B against the old layout (tools/_trace variant library built here), 1080p inference stream and training steps.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
OLD=$(python tools/build_variant.py conv3x3_ws.hip -DWS_PS=9 | tail -1)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "ws or conv or deconv or hr_tail" -x 2>&1 | tail -4 > $O/r04o_pytest.txt
{
echo "== kernel"
timeout 200 python tools/mb_ws.py 2>&1 | grep conv
TECOGAN_HIP_LIB=$OLD timeout 200 python tools/mb_ws.py 2>&1 | grep conv
echo "== 1080p inference stream (new, old, new, old)"
for v in "" $OLD "" $OLD; do TECOGAN_HIP_LIB=$v timeout 300 python tools/bench_infer.py 2>&1 | tail -1; done
echo "== TecoGAN / FRVSR steps (new, old, new, old)"
for v in "" $OLD "" $OLD; do TECOGAN_HIP_LIB=$v timeout 300 python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10 2>&1 | tail -1 | cut -c1-160; done
} > $O/r04o_ab.txt 2>&1
cat $O/r04o_pytest.txt $O/r04o_ab.txt
