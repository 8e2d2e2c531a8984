cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 0" "0 1" "1 1"; do set -- $cfg; AB_TAG="wr_prio=$1 dcap=$2" TG_WR_PRIO=$1 TG_AB_DCAP=$2 timeout 120 python tools/_ab.py 2>&1 | grep "ms/step" | cut -c1-200; done
