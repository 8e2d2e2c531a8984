#!/bin/bash
# Round-4 GPU session D: (1) what would whole-line streams buy the wide-layer DMA conv?  timing-only hooks: weight panel
# contiguous per stage (TG_C3DMA_WTEST), halo addressed as channel-blocked activations (TG_C3DMA_HTEST); (2) the full GPU
# suite on the round's state; (3) the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
  echo "== wide-layer DMA conv, TG_C3DMA_WTEST=$1 TG_C3DMA_HTEST=$2 (timing only when != 0 0)"
  TG_C3DMA_WTEST=$1 TG_C3DMA_HTEST=$2 timeout 200 python tools/microbench.py --only "conv3x3 vgg " 2>&1 | grep "conv3x3"
  TG_C3DMA_WTEST=$1 TG_C3DMA_HTEST=$2 timeout 200 python tools/microbench.py --only "conv3x3 wide" 2>&1 | grep "conv3x3"
done
} > $O/r04d_dma_stream.txt 2>&1
cat $O/r04d_dma_stream.txt
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/r04d_pytest_gpu.log 2>&1; grep -E "passed|failed" $O/r04d_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR|^\[C3|^\[270" $O/r04d_pytest_gpu.log | cut -c1-300
( time timeout 400 python bench.py ) > $O/r04d_bench.json 2> $O/r04d_bench.err; cut -c1-400 $O/r04d_bench.json; tail -4 $O/r04d_bench.err
