import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

DP = {"dir": None, "procs": []}
PARITY = {}            # test id -> {"logits" | "loss" | "activations" | "grad": max abs error against the reference-made fixtures}


def note_parity(what, err):
    """Called by the golden comparisons (tests/test_hamt_gpu.py:_close): the largest error per quantity class of the running test, printed in
    the terminal summary so that the test log carries the measured distances, not just 'passed' (VERDICT round 5, item 6)."""
    tid = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    cls = "grad" if what.startswith("grad") else "loss" if what in ("loss", "aux", "og_loss") else \
        "logits" if any(k in what for k in ("logits", "fused", "global", "local")) else "activations"
    d = PARITY.setdefault(tid, {})
    d[cls] = max(d.get(cls, 0.0), float(err))


def pytest_terminal_summary(terminalreporter):
    if not PARITY:
        return
    tr = terminalreporter
    tr.section("parity against the reference-made fixtures: max |error| per test (bounds: 1e-4 absolute; gradient entries 2e-4 relative)")
    worst = {}
    for tid, d in sorted(PARITY.items()):
        tr.write_line(tid.split("/")[-1] + "  " + "  ".join(f"{k} {v:.2e}" for k, v in sorted(d.items())))
        for k, v in d.items():
            worst[k] = max(worst.get(k, 0.0), v)
    tr.write_line("WORST  " + "  ".join(f"{k} {v:.2e}" for k, v in sorted(worst.items())))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_session(session):
    expr = session.config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return False
    import torch
    return torch.cuda.device_count() > 0            # counting devices does not initialise the GPU in this process


def pytest_sessionstart(session):
    """The data-parallel test needs three more processes on the GPU (two gloo ranks + the single-process answer). They are started
    HERE, before any test has initialised the GPU in the pytest process (children of a process that already holds the device are
    what the GPU boxes refuse), run beside the first tests and are collected by tests/test_dp_gpu.py."""
    if not _gpu_session(session):
        return
    out = tempfile.mkdtemp(prefix="vlni_dp_")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    DP["dir"] = out
    for name, extra in (("rank0", {"RANK": "0"}), ("rank1", {"RANK": "1"}), ("single", None)):
        log = open(os.path.join(out, name + ".log"), "w")
        e = dict(env, **extra) if extra else {k: v for k, v in env.items() if k not in ("WORLD_SIZE",)}
        cmd = [sys.executable, "-m", "tests.dp_worker", out, "rank" if extra else "single"]
        DP["procs"].append((name, subprocess.Popen(cmd, cwd=ROOT, env=e, stdout=log, stderr=subprocess.STDOUT), log))
    # bench.py's own multi-rank jobs run AFTER those workers (at most 6 processes may hold the GPU): tests/bench_runner.py waits for
    # their result files, then starts the ranks itself - from a process that never touches the GPU (tests/test_bench_gpu.py)
    log = open(os.path.join(out, "bench_runner.log"), "w")
    benv = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    DP["bench"] = (subprocess.Popen([sys.executable, "-m", "tests.bench_runner", out], cwd=ROOT, env=benv, stdout=log, stderr=subprocess.STDOUT), log)


def pytest_sessionfinish(session, exitstatus):
    for _, p, log in DP["procs"]:
        if p.poll() is None:
            p.kill()
        log.close()
    if DP.get("bench"):
        p, log = DP["bench"]
        if p.poll() is None:
            open(os.path.join(DP["dir"], "stop"), "w").close()
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()                      # (its bench children finish on their own; they are bounded by --steps 2)
        log.close()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dp_results():
    """Waits for the workers started at session start; returns {name: loaded result}."""
    import torch
    if DP["dir"] is None:
        pytest.skip("data-parallel workers were not started (no GPU session)")
    res = {}
    for name, p, log in DP["procs"]:
        try:
            rc = p.wait(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = -9
        log.flush()
        text = open(os.path.join(DP["dir"], name + ".log")).read()[-4000:]
        assert rc == 0, f"dp worker {name} exited {rc}:\n{text}"
        res[name] = torch.load(os.path.join(DP["dir"], name + ".pt"))
    return res
