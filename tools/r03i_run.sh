#!/bin/bash
# Round-3 session I: packed tiles (8x8 / 4x4 images) in the wide-layer DMA kernel: parity, microbench A/B, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "wide_layer" 2>&1 | grep -E "passed|failed|Error|assert" | head | tee $O/r03i_pytest.txt
for v in "TG_C3DMA_MIN_WG_PACK=100000" "TG_C3DMA_MIN_WG_PACK=16"; do echo "== microbench $v" | tee -a $O/r03i_microbench.txt
env $v timeout 200 python tools/microbench.py --only "vgg5" 2>&1 | grep conv3x3 | tee -a $O/r03i_microbench.txt
env $v timeout 200 python tools/microbench.py --only "conv3x3 fnet [72" 2>&1 | grep conv3x3 | tee -a $O/r03i_microbench.txt
done
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_C3DMA_MIN_WG_PACK=100000" "TG_C3DMA_MIN_WG_PACK=100" "TG_C3DMA_MIN_WG_PACK=16"; do
  echo "== tecogan $v" | tee -a $O/r03i_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03i_ab.txt
done
