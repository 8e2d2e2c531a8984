#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cat > /tmp/p.py <<'PY'
import sys, json
tag = sys.argv[1]
line = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
if not line:
    print(tag, "FAILED"); sys.exit(0)
d = json.loads(line[-1])
print(tag, d["ms_per_step"], d["losses"]["vgg_all"], d["losses"]["All_loss_Gen"])
PY
python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>&1 | python /tmp/p.py base
for c in 64 128 192 256; do TG_OVERLAP_VGG=$c python bench.py --config tecogan --steps 20 --no-cpu-baseline 2>&1 | python /tmp/p.py cap$c; done
TG_OVERLAP_VGG=128 python bench.py --config tecogan --steps 5 --no-cpu-baseline 2>&1 | grep -i "error\|Traceback" | head -5
