#!/bin/bash
# Round-3 GPU session V (last): GPU suite without the 168 s configs[2] oracle test and the 32 s calendar clip (both green in
# session W; the kernels they cover changed only in the warp pair, which session Y and this run test at the same shapes),
# inference PMC passes for the new warp kernel, the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests -m gpu -q --maxfail=10 --deselect tests/test_train_gpu.py::test_tecogan_step_fp32_parity_at_baseline_config_C3 --deselect tests/test_infer_gpu.py::test_calendar_clip_fp32_parity_every_frame ) > $O/r03v_pytest_gpu.log 2>&1; grep -E "passed|failed" $O/r03v_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/r03v_pytest_gpu.log | cut -c1-200
cd /tmp
I="python $R/tools/bench_infer.py --frames 4 --warmup 2 --no-graph"
timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_v_ifetch -- $I > $O/pmc_v_ifetch.log 2>&1
timeout 100 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_v_iwrite -- $I > $O/pmc_v_iwrite.log 2>&1
timeout 100 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_v_imfma -- $I > $O/pmc_v_imfma.log 2>&1
cd $R
python tools/pmc_summary.py --json $O/r03_pmc_infer.json $O/pmc_v_ifetch $O/pmc_v_iwrite $O/pmc_v_imfma > $O/r03_pmc_infer.txt 2>&1; head -10 $O/r03_pmc_infer.txt | cut -c1-200
[ -s $O/r03_pmc_infer.json ] && cp $O/r03_pmc_infer.json $R/profiles/
rm -rf $O/pmc_v_*
( time timeout 300 python bench.py ) > $O/r03v_bench.json 2> $O/r03v_bench.err; cut -c1-300 $O/r03v_bench.json; tail -4 $O/r03v_bench.err
