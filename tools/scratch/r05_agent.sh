#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
python -m pytest tests/test_wrappers_gpu.py tests/test_hamt_gpu.py tests/test_duet_gpu.py tests/test_edges_gpu.py tests/test_entry_gpu.py tests/test_builders_gpu.py tests/test_bench_gpu.py -q > $O/t.log 2>&1; tail -3 $O/t.log; grep -n "^E  " $O/t.log | cut -c1-400 | head -20
