#!/bin/bash
# Round-2 GPU session E: segment-DAG engine (single-stream graphs on two streams): tests + overlap pieces A/B; stream /
# instruction priority for the chain.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for p in 0 1; do TG_C3_PRIO=$p timeout 120 python tools/mb_forktax.py 2>&1 | tail -2 | tee -a $O/r02e_forktax.txt; done
( time timeout 900 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=5 ) > $O/r02e_pytest_gpu.log 2>&1; tail -25 $O/r02e_pytest_gpu.log | cut -c1-300
B="python bench.py --steps 40 --warmup 3 --no-sub --no-roofline --no-cpu-baseline"
for parts in 0 15 1 4 5 7 8 31; do
  echo "== tecogan TG_OVERLAP_PARTS=$parts" | tee -a $O/r02e_ab.txt; TG_OVERLAP_PARTS=$parts timeout 120 $B 2>&1 | tail -1 | cut -c1-150 | tee -a $O/r02e_ab.txt
done
echo "== tecogan TG_OVERLAP_PARTS=15 TG_C3_PRIO=1" | tee -a $O/r02e_ab.txt; TG_C3_PRIO=1 TG_OVERLAP_PARTS=15 timeout 120 $B 2>&1 | tail -1 | cut -c1-150 | tee -a $O/r02e_ab.txt
for parts in 0 16; do
  echo "== frvsr TG_OVERLAP_PARTS=$parts" | tee -a $O/r02e_ab.txt; TG_OVERLAP_PARTS=$parts timeout 120 $B --config frvsr 2>&1 | tail -1 | cut -c1-150 | tee -a $O/r02e_ab.txt
done
