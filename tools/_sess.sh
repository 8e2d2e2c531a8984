cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for s in 0 1 2 3; do TG_RT_STAGGER=$s timeout 100 python tools/_mb_rt.py 2>&1 | grep stagger; done
