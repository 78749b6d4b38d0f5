#!/bin/bash
# round 4, call 1: stale-mask repro (4 combinations) + baseline bench line of the round-3 build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c1
mkdir -p $O
for a in "0 0" "0 1" "1 0" "1 1"; do
  timeout -k 10 240 python3 tools/stale_mask_repro.py $a > $O/repro_${a// /_}.log 2>&1
  echo "repro $a rc=$?"
  tail -n 2 $O/repro_${a// /_}.log
done
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c1/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
