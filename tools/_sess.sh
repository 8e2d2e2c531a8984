cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -q -s -k "standin or world2 or exchange or one_rank or data_parallel or validation_pass" --tb=short 2>&1 | grep -v "^$" | cut -c1-300 | tail -25 ) > $O/r05o_pytest.log 2>&1; cat $O/r05o_pytest.log
