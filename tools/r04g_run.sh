#!/bin/bash
# Round-4 GPU session G: single forward cuts (VGG chunking) on one box; anatomy of a VGG pass (per-kernel table under rocprofv3,
# pass time against the image count); per-NODE cost of the serial step (kernel durations + gaps).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
{
B="python bench.py --no-sub --no-roofline --no-cpu-baseline --steps 150 --warmup 10"
ms() { grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for c in 11 8 6 9 11 7 13; do echo "== tecogan TG_VGG_CUTS=$c"; TG_VGG_CUTS=$c timeout 120 $B 2>/dev/null | ms; done
echo "== frvsr"; timeout 120 $B --config frvsr 2>/dev/null | ms
echo "== VGG pass against the image count"; timeout 200 python tools/mb_vgg.py 2>&1 | grep "VGG-19"
timeout 200 python tools/mb_vgg.py --flags 1 --n 44 2>&1 | grep "VGG-19"
timeout 200 python tools/mb_vgg.py --fwd-only --n 40 76 2>&1 | grep "VGG-19"
} > $O/r04g_ab.txt 2>&1
cat $O/r04g_ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_g_vgg -o vgg -- python $R/tools/mb_vgg.py --n 32 > $O/prof_g_vgg.log 2>&1
DB=$(ls $O/prof_g_vgg/*/*.db $O/prof_g_vgg/*.db 2>/dev/null | head -1); [ -n "$DB" ] && python $R/tools/prof_summary.py $DB $O/r04g_vgg32_kernel_stats.txt 60; head -45 $O/r04g_vgg32_kernel_stats.txt | cut -c1-170
TG_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_g_serial -- python $R/bench.py --no-sub --no-roofline --no-cpu-baseline --steps 6 --warmup 3 > $O/prof_g_serial.log 2>&1
python $R/tools/timeline.py $O/prof_g_serial $O/r04g_timeline_serial.csv --last 12000 && python $R/tools/node_costs.py $O/r04g_timeline_serial.csv --steps 4 --top 45 > $O/r04g_node_costs.txt 2>&1; cat $O/r04g_node_costs.txt | cut -c1-150
rm -rf $O/prof_g_vgg $O/prof_g_serial $O/r04g_timeline_serial.csv
