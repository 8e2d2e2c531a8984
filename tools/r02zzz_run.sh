#!/bin/bash
# Round-2 GPU session ZZZ: last sanity of the training paths with the final defaults (bf16 tests, step rates).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "bf16 or frvsr_two_steps or three_steps" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 | cut -c1-300 | tee $O/r02zzz_pytest.txt
J="import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
B="python bench.py --steps 100 --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
echo "== tecogan" | tee -a $O/r02zzz_ab.txt; timeout 100 $B 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02zzz_ab.txt
echo "== frvsr" | tee -a $O/r02zzz_ab.txt; timeout 100 $B --config frvsr 2>&1 | tail -1 | python -c "$J" | tee -a $O/r02zzz_ab.txt
