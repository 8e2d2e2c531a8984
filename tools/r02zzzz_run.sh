#!/bin/bash
# Round-2 GPU session (last): LDS-staged weight prologue A/B on the training steps, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in "TG_C3WS_WLDS=1" "TG_C3WS_WLDS=0"; do
echo "== tecogan $v" | tee -a $O/r02zzzz_ab.txt; env $v timeout 60 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02zzzz_ab.txt
echo "== frvsr $v" | tee -a $O/r02zzzz_ab.txt; env $v timeout 60 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02zzzz_ab.txt
done
