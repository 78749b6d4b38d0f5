#!/bin/bash
# round 5, call 3: n-problem GEMM launches + the lockstep history step (tests, then bench A/B)
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "multi_problem or dual or variants_identical or timed_gemm or mixed_epilogue" > $O/t_ops.log 2>&1; tail -4 $O/t_ops.log
python -m pytest tests/test_tape_gpu.py tests/test_buckets_gpu.py -q -x > $O/t_tape.log 2>&1; tail -4 $O/t_tape.log
python -m pytest tests/test_hamt_gpu.py -q -x -k "reference_golden and (taped or graph)" > $O/t_hamt.log 2>&1; tail -4 $O/t_hamt.log
A="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity"
for i in 1 2; do
VLNI_LOCKSTEP_HISTORY=0 python bench.py $A > $O/bench_lock0_$i.json 2> $O/bench_lock0_$i.err; echo lock0 done
VLNI_LOCKSTEP_HISTORY=1 python bench.py $A > $O/bench_lock1_$i.json 2> $O/bench_lock1_$i.err; echo lock1 done
done
python - <<'PY'
import json
for n in ("lock0_1","lock1_1","lock0_2","lock1_2"):
    try:
        d=json.load(open(f"gpurun_out/r5c/bench_{n}.json")); f=d["roofline"].get("families",{})
        print(n, d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("launches"), {k:(v.get("frac"), v.get("ms")) for k,v in f.items()})
    except Exception as e: print(n, "failed", e)
PY
