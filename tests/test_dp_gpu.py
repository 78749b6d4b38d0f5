"""Data parallelism as the reference runs it (one process per GPU, DDP: VLN-HAMT/finetune_src/r2r/agent_cmt.py:61-63, seed + rank:
r2r/main.py:446), rehearsed with two gloo ranks that share the one GPU of the test box. The ranks and the single-process answer
are separate processes started by tests/conftest.py at session start (tests/dp_worker.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_constructor_broadcast_makes_replicas_identical(dp_results):
    r0, r1 = dp_results["rank0"], dp_results["rank1"]
    assert not r0["changed_by_broadcast"] and r1["changed_by_broadcast"]         # rank 1 started from other values of the new heads
    assert torch.equal(r0["params0"], r1["params0"])


def test_rank_gradients_equal_the_mean_of_the_per_episode_gradients(dp_results):
    r0, r1, one = dp_results["rank0"], dp_results["rank1"], dp_results["single"]
    assert r0["n_ranges"] >= 3                                                    # the flush -> all-reduce pipeline really had stages
    assert torch.equal(r0["grads_f32"], r1["grads_f32"]) and torch.equal(r0["grads_bf16"], r1["grads_bf16"])
    ref = one["grads"]
    rel = ((r0["grads_f32"] - ref).norm() / ref.norm()).item()
    assert rel < 2e-5, rel                     # bf16 activations, float32 gradients: only the summation order differs
    rel16 = ((r0["grads_bf16"] - ref).norm() / ref.norm()).item()
    assert rel16 < 6e-3, rel16                 # bf16 payload: 8 bits of mantissa per addend


def test_graphed_dp_steps_track_the_single_process(dp_results):
    r0, r1, one = dp_results["rank0"], dp_results["rank1"], dp_results["single"]
    assert r0["n_flush_graphs"] >= 2
    assert torch.equal(r0["params"], r1["params"])                                # replicas stay in lock-step
    d = (r0["params"] - one["params"]).abs()
    # two Adam steps at lr 1e-3 move every parameter by <= 2e-3; sign flips of ~zero gradients bound the difference
    assert d.max().item() < 4.5e-3 and d.mean().item() < 1e-5, (d.max().item(), d.mean().item())
    assert (r0["params"] - r0["params0"]).abs().max().item() > 1e-4               # and the steps did move them
