#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "
import bench, tempfile; t=tempfile.mkdtemp(dir='/tmp'); bench._write_scenes(t); print(t)" > /tmp/scn.txt 2>/dev/null
S=$(tail -1 /tmp/scn.txt)
for v in "--prefetch 2" "--prefetch 0" "--prefetch 2 --threads 1"; do
timeout 200 python tools/mb_mainloop.py $v --scenes $S 2>&1 | grep -E "synthetic|png|cache|inside" | sed "s/^/[$v] /" | tee -a $O/r03g_mainloop.txt
done
timeout 200 python tools/mb_mainloop.py 2>&1 | grep -E "synthetic|png|cache|inside" | tee -a $O/r03g_mainloop.txt
