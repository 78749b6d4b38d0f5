set -e
mkdir -p gpurun_out/r4g
export TMPDIR=/tmp
M0=5504 M1=2752 GRAPH=1 EPILOGUES=3072x768 NT_VARIANTS=5,12,14,15,32,1 timeout -k 10 300 python tools/gemm_step_probe.py > gpurun_out/r4g/epi3072.log 2>&1
M0=5504 M1=2752 GRAPH=1 EPILOGUES=2304x768 NT_VARIANTS=5,12,14,15,32,1 timeout -k 10 300 python tools/gemm_step_probe.py > gpurun_out/r4g/epi2304.log 2>&1
sed -i 's/(8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768)/(8256, 3072, 768), (8256, 768, 3072), (8256, 2304, 768), (8256, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768)/' tools/hipblaslt_names.py
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hn -- python3 $GRAFT_REPO_ROOT/tools/hipblaslt_names.py > $GRAFT_REPO_ROOT/gpurun_out/r4g/hn.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/hn -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r4g/hn_stats.csv
f=$(find /tmp/hn -name "*kernel_trace.csv" | head -1)
cut -d, -f8-20 $f | head -120 > gpurun_out/r4g/hn_trace_head.csv || true
