#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c12; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hamt_gpu.py tests/test_tape_gpu.py tests/test_buckets_gpu.py tests/test_wrappers_gpu.py tests/test_dropout_gpu.py tests/test_fulldepth_gpu.py -q -m gpu -p no:cacheprovider -x > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v Warn $O/tests.log | grep "^FAILED\|^E  \|passed\|failed\|Error" | cut -c1-250 | head -20
for g in 1 0; do
VLNI_LANG_QKV_ONCE=$g timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity > $O/bench_$g.json 2> $O/bench_$g.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r4c12/bench_$g.json").read().strip().splitlines()[-1])
print("LANG_QKV_ONCE=$g ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "loss", d["config"]["loss"])
PY
done
