mkdir -p gpurun_out/r4g
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune gpurun_out/r4g/tune25.pkl > gpurun_out/r4g/b25a.log 2>&1
VLNI_REDUCE_EACH=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune25.pkl > gpurun_out/r4g/b25b.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune25.pkl > gpurun_out/r4g/b25c.log 2>&1
VLNI_REDUCE_EACH=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune25.pkl > gpurun_out/r4g/b25d.log 2>&1
