#!/bin/bash
# WRITE_SIZE per launch of one dual GEMM shape for several variants
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2w
mkdir -p $O
: > $O/write.txt
for cfg in "3072 768 14" "3072 768 5" "768 3072 14"; do
  set -- $cfg
  for c in WRITE_SIZE; do
    rm -rf /tmp/pw
    rocprofv3 --pmc $c --output-format csv -d /tmp/pw -- python3 $R/tools/scratch/pk_one.py $1 $2 $3 nt 10 > /dev/null 2>&1
    f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$1 $2 v$3 $c" >> $O/write.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "gemm" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[2], k, len(v), "launches, KB per launch:", round(sum(v[-5:]) / 5))
PY
  done
done
cat $O/write.txt
