#!/bin/bash
# round 5, call 7: the HIP runtime's graph-execution knobs against the step time (the visual chain stalls ~180 us per step at the fork)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
A="--steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-parity --no-roofline"
run() { name=$1; shift; env VLNI_HISTORY_AFTER=0 "$@" python bench.py $A > $O/b_$name.json 2> $O/b_$name.err; python - <<PY
import json
try: print("$name", json.load(open("$O/b_$name.json"))["ms_per_step"])
except Exception as e: print("$name failed", e)
PY
}
run base A=1
run batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch16 DEBUG_HIP_GRAPH_BATCH_SIZE=16
run batch1024 DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run clrbatch DEBUG_CLR_MAX_BATCH_SIZE=4096
run q1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run q2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run q8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run aql64k ROC_AQL_QUEUE_SIZE=65536
run cap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run cap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run base2 A=1
