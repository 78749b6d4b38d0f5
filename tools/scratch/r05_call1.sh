#!/bin/bash
# round 5, call 1: weight-gradient tile -> XCD mapping A/B (tests, probe, bench step)
set -e
O=gpurun_out/r5a; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "wgrad or ring_weight" > $O/t_wgrad.log 2>&1; tail -2 $O/t_wgrad.log
VLNI_TN_XCD=0 python tools/tn_xcd_probe.py > $O/probe_xcd0.log 2>&1; echo probe0 done
VLNI_TN_XCD=1 python tools/tn_xcd_probe.py > $O/probe_xcd1.log 2>&1; echo probe1 done
A="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity"
VLNI_TN_XCD=0 python bench.py $A > $O/bench_xcd0.json 2> $O/bench_xcd0.err; echo bench0 done
VLNI_TN_XCD=1 python bench.py $A > $O/bench_xcd1.json 2> $O/bench_xcd1.err; echo bench1 done
VLNI_TN_XCD=0 python bench.py $A > $O/bench_xcd0b.json 2> $O/bench_xcd0b.err; echo bench0b done
VLNI_TN_XCD=1 python bench.py $A > $O/bench_xcd1b.json 2> $O/bench_xcd1b.err; echo bench1b done
python - <<'PY'
import json
for n in ("xcd0","xcd1","xcd0b","xcd1b"):
    d=json.load(open(f"gpurun_out/r5a/bench_{n}.json")); f=d["roofline"].get("families",{})
    print(n, d["ms_per_step"], d["roofline"]["frac"], {k:(v.get("frac"), v.get("ms")) for k,v in f.items()} if isinstance(f,dict) else f)
PY
