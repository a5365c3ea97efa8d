#!/bin/bash
# Everything the round's profiles/ directory holds, in one gpurun call:   tools/profile_round.sh TAG   (e.g. v6)
# kernel-trace summaries and the bench lines of both workloads, then the two PMC passes for the ViT GEMM traffic.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/vit_bench.json 2> $out/vit_bench.err; tail -c 600 $out/vit_bench.json
python3 bench.py --workload whisper > $out/whisper_bench.json 2> $out/whisper_bench.err; tail -c 400 $out/whisper_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/vit_trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/vit_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/whisper_trace -- python3 bench.py --workload whisper --steps 2 --warmup 1 --no-cpu-baseline > $out/whisper_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_w.log 2>&1
python3 tools/collect_traffic.py $out/f $out/w linear_bf16 $out/vit_traffic.json
# keep only the summaries (the raw traces are large)
find $out -name "*kernel_stats.csv" | head
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
du -sh $out
