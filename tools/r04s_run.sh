#!/bin/bash
# Round-4 GPU session S: (1) FNet's backward pass in two batch slices, the late one beside the BPTT (TG_FNET_BWD_SPLIT=k);
# (2) packed 4x4 / 8x8 DMA tiles for FNet's deepest levels at small batches (TG_C3DMA_MIN_WG_PACK: 16 -> 8 / 4).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "two_batch_slices" 2>&1 | tail -4 > $O/r04s_pytest.txt
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
{
for k in 0 9 12 6 0 9; do echo "== tecogan TG_FNET_BWD_SPLIT=$k"; TG_FNET_BWD_SPLIT=$k timeout 300 $B 2>&1 | tail -1 | ms; done
for k in 0 4 6; do echo "== frvsr TG_FNET_BWD_SPLIT=$k"; TG_FNET_BWD_SPLIT=$k timeout 300 $B --config frvsr 2>&1 | tail -1 | ms; done
for p in 16 8 4 16 4; do echo "== frvsr TG_C3DMA_MIN_WG_PACK=$p"; TG_C3DMA_MIN_WG_PACK=$p timeout 300 $B --config frvsr 2>&1 | tail -1 | ms; done
for p in 16 4; do echo "== tecogan TG_C3DMA_MIN_WG_PACK=$p"; TG_C3DMA_MIN_WG_PACK=$p timeout 300 $B 2>&1 | tail -1 | ms; done
echo "== timeline TG_FNET_BWD_SPLIT=9"; TG_FNET_BWD_SPLIT=9 timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -22
} > $O/r04s_ab.txt 2>&1
cat $O/r04s_pytest.txt $O/r04s_ab.txt
