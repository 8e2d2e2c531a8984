R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=25 --durations=6 ) > $O/r05zz_pytest_gpu.log 2>&1; grep -E "passed|failed" $O/r05zz_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/r05zz_pytest_gpu.log | cut -c1-200
( time timeout 700 python bench.py ) > $O/r05zz_bench.json 2> $O/r05zz_bench.err; cut -c1-300 $O/r05zz_bench.json; tail -3 $O/r05zz_bench.err
cd /tmp
B="python $R/bench.py --no-sub --no-roofline --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_z_teco -o teco -- $B --steps 20 --warmup 3 > $O/prof_z_teco.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_z_inf -o inf -- python $R/tools/bench_infer.py > $O/prof_z_inf.log 2>&1
for n in teco:tecogan inf:infer1080p; do d=${n%%:*}; f=${n##*:}; db=$(find $O/prof_z_$d -name "*.db" | head -1); python $R/tools/prof_summary.py $db $O/r05_${f}_bf16_kernel_stats.txt 60; rm -rf $O/prof_z_$d; done
cd $R
timeout 200 python tools/seg_timeline.py --steps 30 2>&1 | grep -v "^ROCm\|^HIP\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" | head -24 > $O/r05zz_seg_timeline.txt
cat $O/r05zz_seg_timeline.txt | head -18
