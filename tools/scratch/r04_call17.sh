#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c17; mkdir -p $O
VLNI_GEMM_BREAKDOWN=1 timeout -k 10 600 python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extras --no-parity > $O/bench.json 2> $O/bench.err
grep "gemm M=" $O/bench.err | cut -c18-140
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c17/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
