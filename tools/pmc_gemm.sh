#!/bin/bash
# PMC passes over one GEMM shape (separate passes, counters only).   tools/pmc_gemm.sh OUTDIR M N K
# At most TWO TCP counters per pass: round 1's four-counter TCP lines died in rocprofv3 with "error code 38: Request exceeds the
# capabilities of the hardware" (gpurun_out/pmc/g8192.p2.log) - the TCP block does not hold four of these at once.
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r line; do
  i=$((i+1))
  echo "pass $i: $line"
  timeout -k 5 120 rocprofv3 --pmc $line --output-format csv -d $out/p$i -- python3 tools/gemm_one.py "$@" > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out.p$i.log; }
done <<'EOC'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum
TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM
EOC
python3 - "$out" <<'EOP'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_bf16" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:45s} {sum(v)/len(v):18.1f}  (n={len(v)})")
EOP
