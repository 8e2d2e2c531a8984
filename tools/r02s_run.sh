#!/bin/bash
# Round-2 GPU session S: two-tiles-per-stage variant of the wide-layer DMA kernel: parity, trace, microbench, step A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "wide_layer" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | cut -c1-300 | tee $O/r02s_pytest.txt
timeout 100 python tools/trace_dma.py 2>&1 | grep -E "==|stage  [2-5] " | tee $O/r02s_trace_dma.txt
for v in "" "TG_C3DMA_PAIR=0"; do echo "== microbench $v" | tee -a $O/r02s_microbench.txt; env $v timeout 200 python tools/microbench.py --only "conv3x3 wide" 2>&1 | tail -8 | tee -a $O/r02s_microbench.txt; env $v timeout 100 python tools/microbench.py --only "conv3x3 vgg " 2>&1 | tail -3 | tee -a $O/r02s_microbench.txt; done
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "" "TG_C3DMA_PAIR=0"; do
  echo "== tecogan $v" | tee -a $O/r02s_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02s_ab.txt
done
