#!/bin/bash
# SQ counter passes over one attention shape.   tools/pmc_attn.sh OUTDIR B L H
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r line; do
  i=$((i+1))
  echo "pass $i: $line"
  timeout -k 5 120 rocprofv3 --pmc $line --output-format csv -d $out/p$i -- python3 tools/attn_one.py "$@" > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out.p$i.log; }
done <<'EOC'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM
EOC
python3 - "$out" <<'EOP'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"] or "attn_head" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} {sum(v)/len(v):18.1f}  (n={len(v)})")
EOP
