#!/bin/bash
# SQ counters of one GEMM variant on one dual shape: usage pmc_pp.sh N K variant [nt|nn]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pp
mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/ppp
  rocprofv3 --pmc $set --output-format csv -d /tmp/ppp -- python3 $R/tools/scratch/pk_one.py $1 $2 $3 ${4:-nt} 6 > /dev/null 2>&1
  f=$(find /tmp/ppp -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$1 $2 v$3" >> $O/pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "gemm" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(sys.argv[2], k[-30:], {n: round(sum(v[-3:]) / 3) for n, v in c.items()})
PY
done
tail -4 $O/pmc.txt
