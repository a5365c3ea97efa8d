#!/bin/bash
# rocprofv3 kernel-trace summary of one Wav2Vec2 bench run:   tools/w2v_profile.sh [base|large]
model=${1:-base}
out=$GRAFT_REPO_ROOT/gpurun_out/w2v_prof_$model
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/w2v_bench.py --model $model --steps 3 > $out/bench.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
head -30 $out/kernel_stats.csv | cut -c1-200
