#!/bin/bash
# Per-kernel average durations of a bench.py workload's decode / forward under rocprofv3 (run on the GPU box):
#   tools/step_kernels.sh OUTDIR WORKLOAD [rows]
out=$1; wl=$2; rows=${3:-8}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --workload $wl --no-exact --no-cpu-baseline --steps 2 --warmup 1 > $out/line.json 2>/dev/null
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/prof
python3 - $out/kernel_stats.csv $rows <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2])]:
    print(f"{r['Name'][28:110]:82s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:8.2f} us {r['Percentage']:>6s} %")
P
