#!/bin/bash
# Per-GPU shares of BASELINE configs[4] / configs[3] (bench.py --workload c5 | c4): the bench line and the rocprofv3 kernel summary
# of the same command.   tools/r03_profile_c45.sh OUTDIR   (on the GPU box; summaries are then copied to profiles/r03)
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c5 c4; do
  steps=10; warm=3; [ $w == c4 ] && steps=3 && warm=1
  echo "bench line $w"
  timeout -k 5 500 python3 bench.py --workload $w --steps $steps --warmup $warm --no-cpu-baseline --no-exact > $out/bench_${w}_line.json 2> $out/bench_${w}.err || echo "bench $w failed"
  echo "kernel trace $w"
  timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -- python3 bench.py --workload $w --steps $steps --warmup $warm --no-cpu-baseline --no-exact > $out/kt_$w.log 2>&1 || echo "kernel trace $w failed"
  cp $(find $out/kt_$w -name "*kernel_stats.csv" | head -1) $out/bench_${w}_kernel_stats.csv 2>/dev/null
  grep "^{\"metric\"" $out/kt_$w.log | tail -1 > $out/bench_${w}_line_under_rocprof.json
done
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete; find $out -name "*agent_info.csv" -delete; find $out -name "*domain_stats.csv" -delete
find $out -type d -empty -delete
ls -la $out
