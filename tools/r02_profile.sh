#!/bin/bash
# Round-2 profiles of the default bench command: per-kernel durations (kernel trace) and HBM-side traffic (PMC, separate passes).
#   tools/r02_profile.sh OUTDIR      (run on the GPU box; summaries are copied to profiles/r02 afterwards)
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "kernel trace"; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/kt.log 2>&1 || echo "kernel trace failed"
echo "vit FETCH"; timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/vf -- python3 bench.py --workload vit --steps 3 --warmup 1 --no-cpu-baseline > $out/vf.log 2>&1 || echo "vf failed"
echo "vit WRITE"; timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/vw -- python3 bench.py --workload vit --steps 3 --warmup 1 --no-cpu-baseline > $out/vw.log 2>&1 || echo "vw failed"
# (the whole-decode-step traffic of the Whisper leg: raw TCC_EA0_RDREQ passes, tools/collect_step_traffic.py - the derived FETCH_SIZE hangs on that run)
python3 tools/collect_traffic.py $out/vf $out/vw linear_bf16 $out/vit_traffic.json
python3 tools/collect_traffic.py $out/vf $out/vw vit_tokens_kernel $out/vit_tokens_traffic.json
python3 tools/collect_traffic.py $out/vf $out/vw attn_head_hd64 $out/attention_traffic.json
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv 2>/dev/null
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
ls -la $out
