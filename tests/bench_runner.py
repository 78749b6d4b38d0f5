"""Started by tests/conftest.py at session start (before the pytest process touches the GPU; this process never does): waits for the
data-parallel workers of tests/dp_worker.py to finish, then runs two short bench.py jobs whose JSON lines tests/test_bench_gpu.py reads:

  bench2.json       python bench.py --gpus 2 ...   two ranks on the box's one GPU (VLNI_ONE_GPU=1) over gloo: the launcher, the agreed
                    capture decision, rank 0's kernel choices on every rank, graph replays with the exchange between them
  bench2_duet_f16.json  the same two gloo ranks with --model duet --dtype fp16 (BASELINE.json configs[4]'s argument path)
  bench_rccl1.json  python bench.py (one rank) with VLNI_FORCE_COLLECTIVES=1: the same pipeline through a 1-rank RCCL communicator -
                    the watchdog thread beside thread-local graph captures, asynchronous work objects on the side stream, the exchange
                    between the replayed graphs - the part a gloo rehearsal cannot reach on a one-GPU box
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--batch", "8", "--T", "3", "--no-cpu-baseline", "--no-extras", "--no-parity"]


def main():
    out = sys.argv[1]
    t0 = time.time()
    while time.time() - t0 < 1500 and not all(os.path.exists(os.path.join(out, n + ".pt")) for n in ("rank0", "rank1", "single")):
        if os.path.exists(os.path.join(out, "stop")):
            return
        time.sleep(1.0)
    jobs = (("bench2", ["--gpus", "2", "--no-roofline"], {"VLNI_ONE_GPU": "1", "VLNI_DIST_BACKEND": "gloo", "VLNI_BENCH_KEEP_GRAPH": "1"}),
            ("bench_rccl1", [], {"VLNI_FORCE_COLLECTIVES": "1", "VLNI_BENCH_KEEP_GRAPH": "1"}),
            # BASELINE.json configs[4]'s shape of run (DUET, data-parallel, float16 with the loss scaler, alignment head on) over two gloo ranks
            ("bench2_duet_f16", ["--gpus", "2", "--no-roofline", "--model", "duet", "--dtype", "fp16"],
             {"VLNI_ONE_GPU": "1", "VLNI_DIST_BACKEND": "gloo", "VLNI_BENCH_KEEP_GRAPH": "1"}))
    for name, extra, env in jobs:
        with open(os.path.join(out, name + ".log"), "w") as log:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, cwd=ROOT, env=dict(os.environ, **env),
                               stdout=subprocess.PIPE, stderr=log, timeout=900)
        with open(os.path.join(out, name + ".json.tmp"), "wb") as f:
            f.write(r.stdout)
        with open(os.path.join(out, name + ".rc"), "w") as f:
            f.write(str(r.returncode))
        os.replace(os.path.join(out, name + ".json.tmp"), os.path.join(out, name + ".json"))


if __name__ == "__main__":
    main()
