"""bench.py's multi-rank path on the one-GPU test box (jobs run by tests/bench_runner.py, started from tests/conftest.py before this
process initialises the GPU): two gloo ranks sharing the GPU, and ONE rank through a real RCCL communicator."""
import json
import os
import time

import pytest

from tests.conftest import DP

pytestmark = pytest.mark.gpu


def _job(name):
    if DP["dir"] is None:
        pytest.skip("bench jobs were not started (no GPU session)")
    path = os.path.join(DP["dir"], name + ".json")
    t0 = time.time()
    while not os.path.exists(path) and time.time() - t0 < 1500:
        time.sleep(1.0)
    log = open(os.path.join(DP["dir"], name + ".log")).read()[-3000:] if os.path.exists(os.path.join(DP["dir"], name + ".log")) else ""
    assert os.path.exists(path), f"{name} did not finish:\n{log}"
    rc = int(open(os.path.join(DP["dir"], name + ".rc")).read())
    assert rc == 0, f"{name} exited {rc}:\n{log}"
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (lines, log)
    return json.loads(lines[0])


def test_bench_two_ranks_replay_graphs_with_the_exchange_between_them():
    d = _job("bench2")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["launch"].startswith("hipGraph replay") and "ranges" in d["config"]["launch"], d["config"]["launch"]
    assert d["config"]["rccl"]["world_size"] == 2 and d["config"]["rccl"]["backend"] == "gloo"
    ex = d["config"]["rccl"]["exchange"]
    assert ex["ranges"] >= 3 and sum(ex["payload_bytes_per_range"]) > 2e8 and ex["exposed_ms"] >= 0.0
    assert d["value"] > 0 and d["steps"] == 2


def test_bench_exchange_pipeline_through_a_one_rank_rccl_communicator():
    """backend 'nccl' (= RCCL): init with device_id, constructor path, thread-local hipGraph captures beside the communicator's watchdog,
    asynchronous all-reduce work on the side stream between graph replays. One rank, so no bytes cross xGMI - the 8-GPU run is the driver's."""
    d = _job("bench_rccl1")
    assert d["n_gpus"] == 1 and d["config"]["rccl"]["backend"] == "nccl" and d["config"]["rccl"]["forced_single_rank"] is True
    assert d["config"]["launch"].startswith("hipGraph replay") and "ranges" in d["config"]["launch"], d["config"]["launch"]
    assert d["config"]["rccl"]["exchange"]["ranges"] >= 3
    assert d["roofline"] is not None and d["value"] > 0


def test_bench_two_ranks_duet_float16_configs4_argument_path():
    """BASELINE.json configs[4]: DUET, data parallel, float16 + dynamic loss scale, alignment head on - the first 8-GPU driver run of that
    configuration must not die on an argument path. Two gloo ranks on the one GPU; what cannot be shown here is bytes on xGMI."""
    d = _job("bench2_duet_f16")
    assert d["n_gpus"] == 2 and d["dtype"] == "fp16" and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 16
    assert d["metric"].startswith("episodes/sec (fwd+bwd) DUET") and d["value"] > 0
    assert d["config"]["launch"].startswith("hipGraph replay"), d["config"]["launch"]
    assert d["config"]["rccl"]["world_size"] == 2 and d["config"]["rccl"]["exchange"]["ranges"] >= 3
