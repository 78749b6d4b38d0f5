#!/bin/bash
# round 5, call 2: golden tests over all drivers (HAMT + DUET), the lang-side cache test, the full-depth taped + replayed bf16 / f16 test
O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests/test_hamt_gpu.py -q -k "reference_golden or language_side_cache" > $O/t_hamt.log 2>&1; tail -30 $O/t_hamt.log
python -m pytest tests/test_duet_gpu.py -q -k "reference_golden" > $O/t_duet.log 2>&1; tail -30 $O/t_duet.log
python -m pytest tests/test_fulldepth_gpu.py -q -s -k "timed_path" > $O/t_full.log 2>&1; tail -30 $O/t_full.log
