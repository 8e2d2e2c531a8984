#!/bin/bash
# Round-4 GPU session AA: transposed-read weight-gradient kernel with swizzled 128-byte pixels in LDS (4-way conflict of
# ds_read_b64_tr_b16 removed): parity, kernel time, TecoGAN / FRVSR steps; against -DTR_NO_SWZ.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
OLD=$(python tools/build_variant.py conv_wgrad_tr.hip -DTR_NO_SWZ | tail -1)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wgrad" -x 2>&1 | tail -3 > $O/r04aa_pytest.txt
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
{
echo "== kernel (swizzled, then -DTR_NO_SWZ)"
timeout 300 python tools/mb_wgrad.py 2>&1 | grep "grouped\|out-conv"
TECOGAN_HIP_LIB=$OLD timeout 300 python tools/mb_wgrad.py 2>&1 | grep "grouped\|out-conv"
echo "== tecogan (new, old, new, old)"
for v in "" $OLD "" $OLD; do TECOGAN_HIP_LIB=$v timeout 300 $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
echo "== frvsr (new, old, new, old)"
for v in "" $OLD "" $OLD; do TECOGAN_HIP_LIB=$v timeout 300 $B --config frvsr 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
} > $O/r04aa_ab.txt 2>&1
cat $O/r04aa_pytest.txt $O/r04aa_ab.txt
