#!/bin/bash
# Round-3 GPU session A (prepared at the end of round 2): first hardware contact of the three kernels written after round 2's
# GPU budget was spent -- the transpose-read probe, their gated tests, and their A/Bs.  Each step is independent.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
[ -x tools/_trace/probe_tr ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probe_tr.hip -o tools/_trace/probe_tr
timeout 60 tools/_trace/probe_tr > $O/r03a_probe_tr.txt 2>&1; head -20 $O/r03a_probe_tr.txt
export TG_TEST_UNVALIDATED=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "wide_layer" 2>&1 | tee $O/r03a_dma3_full.txt | grep -E "passed|failed|Error|assert" | tail -4 | cut -c1-300 | tee $O/r03a_dma3_pytest.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "hr_tail" 2>&1 | tee $O/r03a_hrtail_full.txt | grep -E "passed|failed|Error|assert" | tail -4 | cut -c1-300 | tee $O/r03a_hrtail_pytest.txt
TG_WGRAD_TR=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "transpose_read" 2>&1 | tee $O/r03a_wgradtr_full.txt | grep -E "passed|failed|Error|assert" | tail -4 | cut -c1-300 | tee $O/r03a_wgradtr_pytest.txt
unset TG_TEST_UNVALIDATED
for v in "MB_CONV_FLAGS=0" "MB_CONV_FLAGS=2"; do echo "== microbench $v" | tee -a $O/r03a_microbench.txt; env $v timeout 200 python tools/microbench.py --only "conv3x3 wide" 2>&1 | tail -8 | tee -a $O/r03a_microbench.txt; env $v timeout 100 python tools/microbench.py --only "conv3x3 vgg " 2>&1 | tail -2 | tee -a $O/r03a_microbench.txt; done
for v in "TG_HR_TAIL=0" "TG_HR_TAIL=1"; do echo "== infer $v" | tee -a $O/r03a_ab.txt; env $v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | tee -a $O/r03a_ab.txt; done
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_WGRAD_TR=0" "TG_WGRAD_TR=1"; do
  echo "== tecogan $v" | tee -a $O/r03a_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03a_ab.txt
  echo "== frvsr $v" | tee -a $O/r03a_ab.txt; env $v timeout 120 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03a_ab.txt
done
