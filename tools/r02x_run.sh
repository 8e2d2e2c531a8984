#!/bin/bash
# Round-2 GPU session X: position of the single early/late VGG cut (TG_VGG_CUTS=<frame>), default 10 of 19.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "" "11" "12" "13" "14" ""; do
  echo "== tecogan TG_VGG_CUTS=$v" | tee -a $O/r02x_ab.txt; TG_VGG_CUTS=$v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02x_ab.txt
done
