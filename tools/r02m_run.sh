#!/bin/bash
# Round-2 GPU session M: DMA rounds interleaved with the MFMA groups -- parity, stage trace, microbench; step timelines.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "wide_layer" 2>&1 | tail -3 | cut -c1-300 | tee $O/r02m_pytest.txt
timeout 100 python tools/trace_dma.py 2>&1 | tail -80 | tee $O/r02m_trace_dma.txt
timeout 200 python tools/microbench.py --only "conv3x3 wide" 2>&1 | tail -9 | tee $O/r02m_microbench.txt
timeout 100 python tools/microbench.py --only "conv3x3 vgg " 2>&1 | tail -3 | tee -a $O/r02m_microbench.txt
cd /tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_l -- $B --steps 6 --warmup 3 > $O/tl_l.log 2>&1
python $R/tools/timeline.py $O/tl_l $O/r02m_timeline.csv --last 13000; rm -rf $O/tl_l
TG_OVERLAP_PARTS=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl_l0 -- $B --steps 6 --warmup 3 > $O/tl_l0.log 2>&1
python $R/tools/timeline.py $O/tl_l0 $O/r02m_timeline_serial.csv --last 13000; rm -rf $O/tl_l0
