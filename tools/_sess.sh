cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for kb in 0 88 0 88; do AB_TAG="k4_coexist_kb=$kb" TG_K4_COEXIST_KB=$kb timeout 120 python tools/_ab.py 2>&1 | grep "ms/step" | cut -c1-120; done
