#!/bin/bash
# Round-2 GPU session K: the wide-layer LDS-DMA conv kernel (conv3x3_dma.hip): parity, microbench A/B, step A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "wide_layer or packed" 2>&1 | tail -8 | cut -c1-300 | tee $O/r02k_pytest.txt
for v in "" "TG_NO_C3DMA=1"; do echo "== microbench $v" | tee -a $O/r02k_microbench.txt; env $v timeout 200 python tools/microbench.py --only "conv3x3 wide" 2>&1 | tail -9 | tee -a $O/r02k_microbench.txt; echo; env $v timeout 100 python tools/microbench.py --only "conv3x3 vgg " 2>&1 | tail -3 | tee -a $O/r02k_microbench.txt; done
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "" "TG_NO_C3DMA=1" "TG_OVERLAP_PARTS=0" "TG_OVERLAP_PARTS=0 TG_NO_C3DMA=1"; do
  echo "== tecogan $v" | tee -a $O/r02k_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02k_ab.txt
done
echo "== frvsr" | tee -a $O/r02k_ab.txt; timeout 120 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02k_ab.txt
