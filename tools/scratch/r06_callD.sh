#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d
mkdir -p $O; cd $R
T="timeout -k 10"
$T 900 python3 -m pytest tests/test_ops_gpu.py tests/test_wrappers_gpu.py tests/test_trainer_gpu.py -q -x -m gpu -k "wgrad or ring or table_rows or head_sizes or second_backward or store_mode or agent_backward or embed_combine or block_calls" > $O/tests.txt 2>&1 || (tail -40 $O/tests.txt; exit 1)
tail -5 $O/tests.txt
$T 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --dump-tune $O/tune.json > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json
python3 -c "
import json; t=json.load(open('$O/tune.json')); print('tn picks', sorted(t['tn']))"
