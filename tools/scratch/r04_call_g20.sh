mkdir -p gpurun_out/r4g
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b20a.log 2>&1
VLNI_P8_MIN_ROWS=2048 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b20b.log 2>&1
VLNI_P8_MIN_ROWS=2048 VLNI_P8H_MIN_ROWS=256 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b20c.log 2>&1
VLNI_P8_MIN_ROWS=2048 VLNI_P8H_MIN_ROWS=256 timeout -k 10 300 python bench.py --model duet --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b20d.log 2>&1
timeout -k 10 300 python bench.py --model duet --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b20e.log 2>&1
