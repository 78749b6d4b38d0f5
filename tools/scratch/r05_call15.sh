#!/bin/bash
# round 5, call 15: block-level entry points (self-attention / FFN): equivalence tests, golden tests, drop-in eager timing A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5r; mkdir -p $O; cd $R
python -m pytest tests/test_ops_gpu.py -q -x -k "block_calls or blocks_against_torch" > $O/t_ops.log 2>&1; tail -3 $O/t_ops.log
python -m pytest tests/test_hamt_gpu.py tests/test_duet_gpu.py -q -x -k "reference_golden" > $O/t_gold.log 2>&1; tail -3 $O/t_gold.log
python -m pytest tests/test_tape_gpu.py tests/test_dropout_gpu.py tests/test_trainer_gpu.py -q -x > $O/t_tape.log 2>&1; tail -3 $O/t_tape.log
for b in 0 1; do
VLNI_BLOCK_CALLS=$b python tools/dropin_probe.py hamt > $O/hamt_$b.log 2>&1; head -3 $O/hamt_$b.log | tail -2
VLNI_BLOCK_CALLS=$b python tools/dropin_probe.py duet > $O/duet_$b.log 2>&1; head -3 $O/duet_$b.log | tail -2
done
