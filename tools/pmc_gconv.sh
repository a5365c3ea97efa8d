#!/bin/bash
# PMC passes over the grouped positional conv (separate passes, counters only).   tools/pmc_gconv.sh OUTDIR [d_model]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r line; do
  i=$((i+1))
  echo "pass $i: $line"
  timeout -k 5 120 rocprofv3 --pmc $line --output-format csv -d $out/p$i -- python3 tools/gconv_one.py "$@" > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out.p$i.log; }
done <<'EOC'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM
TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum
FETCH_SIZE
WRITE_SIZE
EOC
python3 - "$out" <<'EOP'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "grouped_conv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:45s} {sum(v)/len(v):18.1f}  (n={len(v)})")
EOP
