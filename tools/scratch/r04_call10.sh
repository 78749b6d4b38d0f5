#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c10; mkdir -p $O
timeout -k 10 900 python3 bench.py --steps 10 --warmup 5 --quick-cpu --dump-tune $O/tune.pkl > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -n 3 $O/bench.err | cut -c1-300
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c10/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic_ratio", d["roofline"]["traffic_ratio"])
print("families", json.dumps(d["roofline"]["families"]))
print("duet", json.dumps(d["extras"]["duet_b32"])[:900])
print({k: v["ms_per_step"] for k, v in d["extras"].items()})
PY
timeout -k 10 600 python3 bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune $O/tune.pkl > $O/bench_ng.json 2> $O/bench_ng.err
echo "no-graph with loaded tune rc=$?"; grep "loaded" $O/bench_ng.err | cut -c1-200
