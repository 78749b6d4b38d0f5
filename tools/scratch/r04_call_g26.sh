mkdir -p gpurun_out/r4g
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune /tmp/tune26.pkl > gpurun_out/r4g/b26.log 2>&1 || exit 1
cd /tmp
for v in once each; do
  if [ $v = each ]; then export VLNI_REDUCE_EACH=1; else unset VLNI_REDUCE_EACH; fi
  echo "start $v" >> $GRAFT_REPO_ROOT/gpurun_out/r4g/p26_progress.log
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr26_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune /tmp/tune26.pkl > $GRAFT_REPO_ROOT/gpurun_out/r4g/p26_$v.log 2>&1 || exit 1
  f=$(find /tmp/tr26_$v -name "*kernel_stats.csv" | head -1)
  grep -i "reduce_parts\|tn_ring\|tn_glds\|tn_big\|adamw\|sumsq" $f | cut -d, -f1-4 | cut -c1-140 > $GRAFT_REPO_ROOT/gpurun_out/r4g/p26_$v.csv
done
