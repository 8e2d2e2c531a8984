#!/bin/bash
# Round-4 GPU session Y: packed DMA conv with 32-channel blocks (J = 1) where 64-channel blocks leave most of the chip idle:
# parity, conv5 kernel time, FNet passes, TecoGAN / FRVSR steps; TG_C3DMA_J1=0 is the old launch.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wide_layer_dma" -x 2>&1 | tail -3 > $O/r04y_pytest.txt
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
{
echo "== conv5 (J1 on, off)"
timeout 200 python tools/mb_conv5.py 2>&1 | grep conv5
TG_C3DMA_J1=0 timeout 200 python tools/mb_conv5.py 2>&1 | grep conv5
echo "== FNet N=72 (J1 on, off)"
MB_N=72 timeout 200 python tools/mb_fnet.py 2>&1 | grep "per pass"
TG_C3DMA_J1=0 MB_N=72 timeout 200 python tools/mb_fnet.py 2>&1 | grep "per pass"
echo "== tecogan (J1 = 1, 0, 1, 0)"
for v in 1 0 1 0; do TG_C3DMA_J1=$v timeout 300 $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
echo "== frvsr (J1 = 1, 0, 1, 0)"
for v in 1 0 1 0; do TG_C3DMA_J1=$v timeout 300 $B --config frvsr 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
} > $O/r04y_ab.txt 2>&1
cat $O/r04y_pytest.txt $O/r04y_ab.txt
