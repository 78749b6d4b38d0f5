mkdir -p gpurun_out/r4g
M0=5504 M1=2752 GRAPH=1 EPILOGUES=3072x768 NT_VARIANTS=5,14,15,32 timeout -k 10 300 python tools/gemm_step_probe.py > gpurun_out/r4g/nt_a.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_ntpre.so M0=5504 M1=2752 GRAPH=1 EPILOGUES=3072x768 NT_VARIANTS=5,14,15,32 timeout -k 10 300 python tools/gemm_step_probe.py > gpurun_out/r4g/nt_b.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune gpurun_out/r4g/tune9.pkl > gpurun_out/r4g/b9a.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_ntpre.so timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune9.pkl > gpurun_out/r4g/b9b.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune9.pkl > gpurun_out/r4g/b9c.log 2>&1
