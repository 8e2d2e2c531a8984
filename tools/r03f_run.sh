#!/bin/bash
# Round-3 session F: host-side changes (device-side dt_ratio, eval_losses, loaders, main.py validation prints) + the full default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_api_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "fading or standin or three_steps or gan_losses or l1 or api or loader or scene or frame" > $O/r03f_pytest_full.txt 2>&1; grep -E "passed|failed|Error|assert" $O/r03f_pytest_full.txt | head -10 | tee $O/r03f_pytest.txt
( time timeout 900 python bench.py ) > $O/r03f_bench.json 2> $O/r03f_bench.err; cut -c1-400 $O/r03f_bench.json; tail -4 $O/r03f_bench.err
