#!/bin/bash
# Round-4 GPU session M: do the latency-regime kernels (small tiles, weights re-streamed per tile, several workgroups per CU) also
# beat the throughput-regime kernels at INFERENCE resolution?  residual block at [1,270,480,64], HR tail at t1 = [1,540,960,64].
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "== residual block"; timeout 200 python tools/mb_resblock.py --big 2>&1 | grep "res block"
echo "== HR tail"; timeout 200 python tools/mb_infer_tail.py 2>&1 | grep "HR tail"
} > $O/r04m_ab.txt 2>&1
cat $O/r04m_ab.txt
