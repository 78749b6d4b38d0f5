#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c11; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "gemm" > $O/tests.log 2>&1
echo "gemm tests rc=$?"; grep -v Warn $O/tests.log | grep "^FAILED\|^E  \|passed\|failed" | cut -c1-250 | head -20
for g in 1 0; do
VLNI_GELU_STORE_GRAD=$g timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity > $O/bench_$g.json 2> $O/bench_$g.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r4c11/bench_$g.json").read().strip().splitlines()[-1])
print("GELU_STORE_GRAD=$g ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "loss", d["config"]["loss"])
PY
done
timeout -k 10 900 python3 -m pytest tests/test_fulldepth_gpu.py tests/test_tape_gpu.py tests/test_dropout_gpu.py -q -m gpu -p no:cacheprovider > $O/tests2.log 2>&1
echo "fulldepth/tape/dropout rc=$?"; grep -v Warn $O/tests2.log | grep "^FAILED\|^E  \|passed\|failed" | cut -c1-250 | head -20
