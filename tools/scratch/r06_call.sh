#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6j
mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_hamt_gpu.py tests/test_duet_gpu.py tests/test_wrappers_gpu.py tests/test_tape_gpu.py -q -x -m gpu -k "no_lang_ca or reverie or fix_local or wrapper or graphed or one_autograd or tape" > $O/tests.txt 2>&1; grep -v "^test_\|^  File" $O/tests.txt | tail -25
VLNI_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/bench_forced_rccl.json 2> $O/bench_forced_rccl.err; python3 -c "
import json; d=json.load(open('$O/bench_forced_rccl.json')); print(d['ms_per_step'], json.dumps(d['config']['rccl'])[:1200])" || tail -5 $O/bench_forced_rccl.err
