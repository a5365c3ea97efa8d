#!/bin/bash
# Per-kernel average durations of the Whisper-base decode step under rocprofv3 (run on the GPU box):  tools/whisper_step_kernels.sh OUTDIR
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --workload whisper --no-exact --no-cpu-baseline --steps 3 --warmup 1 > $out/line.json 2>/dev/null
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/prof
python3 - $out/kernel_stats.csv <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(f"{r['Name'][28:100]:72s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:8.2f} us {r['Percentage']:>6s} %")
P
