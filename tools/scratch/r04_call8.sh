#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c8; mkdir -p $O
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c8/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
VLNI_ATTN_BWD=chunked timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/bench_chunked.json 2> $O/bench_chunked.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c8/bench_chunked.json").read().strip().splitlines()[-1])
print("chunked attention bwd: ms/step", d["ms_per_step"])
PY
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v Warn $O/tests.log | tail -n 6
