J="import sys,json; d=json.loads(sys.stdin.readline()); print(d[\"ms_per_step\"])"
B="python bench.py --warmup 5 --no-sub --no-roofline --no-cpu-baseline"
for v in 24 96 24 96 24; do echo "== tecogan TG_C3DMA_MIN_WG=$v"; TG_C3DMA_MIN_WG=$v timeout 100 $B --steps 120 2>&1 | tail -1 | python -c "$J"; done
for v in 24 96 24 96; do echo "== infer TG_C3DMA_MIN_WG=$v"; TG_C3DMA_MIN_WG=$v timeout 100 python tools/bench_infer.py 2>&1 | tail -1 | cut -c1-86; done
