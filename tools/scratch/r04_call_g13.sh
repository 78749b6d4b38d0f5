mkdir -p gpurun_out/r4g
cd /tmp && export TMPDIR=/tmp
for v in base vform; do
  if [ $v = vform ]; then export VLNI_LIB_PATH=$GRAFT_REPO_ROOT/vln-imagine_amd/build/variants/libvlni_vform.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr13_$v -- python3 $GRAFT_REPO_ROOT/tools/attn_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r4g/ap13_$v.log 2>&1
  f=$(find /tmp/tr13_$v -name "*kernel_stats.csv" | head -1)
  grep -i "attn" $f | cut -d, -f1-4 | cut -c1-200 > $GRAFT_REPO_ROOT/gpurun_out/r4g/ap13_$v.csv
done
cd $GRAFT_REPO_ROOT
unset VLNI_LIB_PATH
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune gpurun_out/r4g/tune13.pkl > gpurun_out/r4g/b13a.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_vform.so timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune13.pkl > gpurun_out/r4g/b13b.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune13.pkl > gpurun_out/r4g/b13c.log 2>&1
