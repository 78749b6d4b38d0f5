#!/bin/bash
# round 5, call 5: what would 3-problem attention / LayerNorm launches buy the lockstep step? (timing-only build of the step without them)
O=gpurun_out/r5e; mkdir -p $O
A="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2; do
VLNI_LOCKSTEP_HISTORY=0 python bench.py $A > $O/b_lock0_$i.json 2> $O/b_lock0_$i.err; echo lock0 done
VLNI_LOCKSTEP_HISTORY=1 python bench.py $A > $O/b_lock1_$i.json 2> $O/b_lock1_$i.err; echo lock1 done
VLNI_LOCKSTEP_HISTORY=1 VLNI_LOCKSTEP_FAKE=1 python bench.py $A > $O/b_fake_$i.json 2> $O/b_fake_$i.err; echo fake done
VLNI_LOCKSTEP_HISTORY=0 VLNI_OVERLAP_HISTORY=0 python bench.py $A > $O/b_serial_$i.json 2> $O/b_serial_$i.err; echo serial done
done
python - <<'PY'
import json
for n in ("lock0_1","lock1_1","fake_1","serial_1","lock0_2","lock1_2","fake_2","serial_2"):
    try:
        d=json.load(open(f"gpurun_out/r5e/b_{n}.json")); print(n, d["ms_per_step"])
    except Exception as e: print(n, "failed", e)
PY
