cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1400 python tools/bf16_trajectory.py --steps 2000 --batches 8 --seeds 3 --out gpurun_out/r05_bf16_trajectory.txt 2>&1 | grep -v "^ROCm\|^HIP\|amdgpu.ids" | tail -30
