#!/bin/bash
# usage: prof_pk.sh <outdir> N K variant
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $GRAFT_REPO_ROOT/tools/scratch/pk_one.py "$@" > /dev/null 2>&1
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1); head -5 $f > $GRAFT_REPO_ROOT/$out/stats_$1_$2_$3.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d /tmp/prof_pmc -- python3 $GRAFT_REPO_ROOT/tools/scratch/pk_one.py "$@" > /dev/null 2>&1
f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/$out/pmc_$1_$2_$3.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "gemm" not in k: continue
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
rm -rf /tmp/prof_kt /tmp/prof_pmc
