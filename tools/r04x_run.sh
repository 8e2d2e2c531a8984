#!/bin/bash
# Round-4 GPU session X: packed-tile DMA conv with the channel block as the fast workgroup index (an XCD owns a slice of the weight
# panel): parity, kernel time, HBM fetch bytes (rocprofv3 --pmc FETCH_SIZE, own pass), TecoGAN step; against -DDM_NO_XCD_REMAP.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
OLD=$(python tools/build_variant.py conv3x3_dma.hip -DDM_NO_XCD_REMAP | tail -1)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wide_layer_dma" -x 2>&1 | tail -3 > $O/r04x_pytest.txt
{
echo "== kernel (remapped, then old mapping)"
timeout 200 python tools/mb_conv5.py 2>&1 | grep conv5
TECOGAN_HIP_LIB=$OLD timeout 200 python tools/mb_conv5.py 2>&1 | grep conv5
echo "== tecogan step (new, old, new, old)"
for v in "" $OLD "" $OLD; do TECOGAN_HIP_LIB=$v timeout 300 python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
} > $O/r04x_ab.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_x_new -- python $R/tools/mb_conv5.py --pmc > $O/pmc_x_new.log 2>&1
TECOGAN_HIP_LIB=$OLD timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_x_old -- python $R/tools/mb_conv5.py --pmc > $O/pmc_x_old.log 2>&1
cd $R
echo "== FETCH_SIZE per launch, all image counts together (remapped / old)" >> $O/r04x_ab.txt
python tools/pmc_summary.py $O/pmc_x_new 2>&1 | grep dma >> $O/r04x_ab.txt
python tools/pmc_summary.py $O/pmc_x_old 2>&1 | grep dma >> $O/r04x_ab.txt
rm -rf $O/pmc_x_new $O/pmc_x_old
cat $O/r04x_pytest.txt $O/r04x_ab.txt
