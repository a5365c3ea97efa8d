#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the HBM-priced kernels of the two benchmarked paths:
# patch-embed (vit_tokens), LayerNorm, log-mel (stft_mel).     tools/hbm_kernels_traffic.sh OUTDIR
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/vf -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/vf.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/vw -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/vw.log 2>&1
# (counter collection over the graph-replaying Whisper bench aborts in rocprofv3 on this image: the front end runs alone)
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/wf -- python3 tools/logmel_one.py > $out/wf.log 2>&1
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/ww -- python3 tools/logmel_one.py > $out/ww.log 2>&1
python3 tools/collect_traffic.py $out/vf $out/vw vit_tokens_kernel $out/vit_tokens.json
python3 tools/collect_traffic.py $out/vf $out/vw layernorm_kernel $out/layernorm_vit.json
python3 tools/collect_traffic.py $out/wf $out/ww stft_mel_kernel $out/stft_mel.json
find $out -name "*counter_collection.csv" -delete
