#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_VGG_CUT=11" "TG_VGG_CUT=9" "TG_VGG_CUT=13" "TG_VGG_CUT=15" "TG_VGG_CUT=11 TG_WGRAD_TR=0"; do
  echo "== tecogan $v" | tee -a $O/r03k_ab.txt; env $v timeout 120 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03k_ab.txt
done
for v in "TG_WGRAD_TR=1" "TG_WGRAD_TR=0" "TG_WGRAD_TR=1" "TG_WGRAD_TR=0"; do
  echo "== frvsr $v" | tee -a $O/r03k_ab.txt; env $v timeout 120 $B --config frvsr --steps 300 2>&1 | tail -1 | python -c "$J" | tee -a $O/r03k_ab.txt
done
