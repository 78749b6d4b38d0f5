#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py -q -x -m gpu -k "partials or store_mode or reduce_parts or wgrad" > $O/t.txt 2>&1; tail -3 $O/t.txt
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune $O/tune.json > $O/b.json 2> $O/b.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune $O/tune.json > $O/prof.json 2> $O/prof.err
python3 tools/step_profile.py $O/trace $O/prof.json r06x 6 > $O/breakdown.txt; grep -i "reduce\|span\|AdamW" $O/breakdown.txt | head -8; rm -rf $O/trace; rm -f profiles/r06x*
