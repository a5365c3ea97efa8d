#!/bin/bash
# Round-3 profiles, one file per LEG and schedule so that every figure of the bench line can be recomputed from a summary:
#   tools/r03_profile.sh OUTDIR      (run on the GPU box; the summaries are then copied to profiles/r03)
# rocprofv3 --kernel-trace --stats of `bench.py --workload vit` (default two-stream schedule, and PM_ENCODER_STREAMS=1) and of
# `--workload whisper`; HBM-side traffic of the ViT leg's kernels (PMC, separate passes); SQ / TCP / TCC counters of the fc2 GEMM;
# the per-shape GEMM table and the vendor yardstick from the SAME box; the bench line itself.
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() {  # name, then bench.py arguments (the program itself behind "--": no env / shell wrapper under rocprofv3)
  local name=$1; shift
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -- python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-exact > $out/kt_$name.log 2>&1 || echo "kernel trace $name failed"
  cp $(find $out/kt_$name -name "*kernel_stats.csv" | head -1) $out/bench_${name}_kernel_stats.csv 2>/dev/null
  grep "^{\"metric\"" $out/kt_$name.log | tail -1 > $out/bench_${name}_line_under_rocprof.json
}
echo "kernel traces"
kt vit_default --workload vit
export PM_ENCODER_STREAMS=1
kt vit_one_stream --workload vit
unset PM_ENCODER_STREAMS
kt whisper_default --workload whisper
echo "vit traffic"
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/vf -- python3 bench.py --workload vit --steps 3 --warmup 1 --no-cpu-baseline > $out/vf.log 2>&1 || echo "vf failed"
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/vw -- python3 bench.py --workload vit --steps 3 --warmup 1 --no-cpu-baseline > $out/vw.log 2>&1 || echo "vw failed"
python3 tools/collect_traffic.py $out/vf $out/vw linear_bf16 $out/vit_traffic.json
python3 tools/collect_traffic.py $out/vf $out/vw vit_tokens_kernel $out/vit_tokens_traffic.json
python3 tools/collect_traffic.py $out/vf $out/vw attn_head_hd64 $out/attention_traffic.json
echo "gemm counters (fc2 shape)"
bash tools/pmc_gemm.sh $out/pmc_fc2 50432 768 3072 > $out/gemm_fc2_pmc_counters.txt 2>&1 || echo "pmc fc2 failed"
bash tools/pmc_gemm.sh $out/pmc_qkv 50432 2304 768 > $out/gemm_qkv_pmc_counters.txt 2>&1 || echo "pmc qkv failed"
echo "per-shape tables"
for k in auto 6 7 2; do
  if [ $k == auto ]; then python3 tools/tile_check.py --time-only; else PM_GEMM_KERNEL=$k python3 tools/tile_check.py --time-only; fi
done > $out/gemm_kernels_steady_state.txt 2>&1
python3 tools/blas_reference_bench.py > $out/vendor_blas_steady_state.txt 2>&1
echo "whisper step traffic (eager decode: counter collection cannot sample graph replays)"
timeout -k 5 250 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/st_raw_f -- python3 bench.py --workload whisper --no-graph --steps 1 --warmup 0 --no-cpu-baseline --no-exact > $out/st_f.log 2>&1 || echo "step fetch pass failed"
timeout -k 5 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/st_w -- python3 bench.py --workload whisper --no-graph --steps 1 --warmup 0 --no-cpu-baseline --no-exact > $out/st_w.log 2>&1 || echo "step write pass failed"
python3 tools/collect_step_traffic.py $out/st_raw_f $out/st_w $out/whisper_step_traffic.json > /dev/null || echo "step traffic failed"
echo "bench line"
python3 bench.py --steps 20 --warmup 5 > $out/bench_line_default_run.json 2> $out/bench_line.err
python3 tools/exact_time.py > $out/exact_mode_cost.txt 2>&1
python3 tools/exact_encoder_profile.py >> $out/exact_mode_cost.txt 2>&1
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
find $out -type d -empty -delete
ls $out
