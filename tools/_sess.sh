cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "conv4x4s2" --tb=line 2>&1 | grep "AssertionError:\|passed\|failed\|Error" | cut -c1-300 ) > $O/r05f_pytest_k4.log 2>&1; cat $O/r05f_pytest_k4.log
