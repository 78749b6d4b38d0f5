mkdir -p gpurun_out/r4g
cd /tmp && export TMPDIR=/tmp
for v in base fast; do
  if [ $v = fast ]; then export VLNI_LIB_PATH=$GRAFT_REPO_ROOT/vln-imagine_amd/build/variants/libvlni_adamf.so; else unset VLNI_LIB_PATH; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr24_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r4g/p24_$v.log 2>&1
  f=$(find /tmp/tr24_$v -name "*kernel_stats.csv" | head -1)
  grep -i "adamw\|sumsq\|reduce_parts" $f | cut -d, -f1-4 | cut -c1-160 > $GRAFT_REPO_ROOT/gpurun_out/r4g/p24_$v.csv
done
