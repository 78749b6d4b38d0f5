"""Shape-bucketed hipGraph cache (vln_imagine_amd/hamt/buckets.py): a ragged stream of batches - text length, view count and
episode length change from batch to batch as in VLN-HAMT/finetune_src/r2r/agent_cmt.py:130-176,498-606 - runs from captured
graphs, one per (L, V, T) bucket, with the batch padded into the bucket's static buffers the way the reference pads inside a batch."""
import time

import pytest
import torch

from tests.golden.variants import HAMT_C1
from tests.test_hamt_gpu import build_product
from vln_imagine_amd import synth
from vln_imagine_amd.hamt.config import HamtConfig
from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode

pytestmark = pytest.mark.gpu
B, I = 8, 4
# (L, V, T) of the stream's batches: three buckets ((48|64|80), (25|31|37), T) are hit, each more than once
STREAM = [(45, 23, 3), (80, 37, 3), (61, 30, 4), (48, 25, 3), (77, 35, 3), (64, 31, 4), (40, 25, 3), (70, 37, 3), (58, 28, 4)]


def _episode(i, L, V, T):
    return synth.HamtEpisode(tag=f"stream{i}", B=B, L=L, V=V, I=I, T=T, ragged=True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ragged_stream_replays_from_bucket_graphs_with_eager_logits(dtype):
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.buckets import HamtGraphBuckets
    from vln_imagine_amd.train import FlatTrainer
    cfg = HamtConfig(**HAMT_C1)
    model = build_product(cfg, dtype)
    tr = FlatTrainer(model, lr=1e-4)
    gb = HamtGraphBuckets(tr, model, B, I)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    try:
        seen, replays = set(), 0
        for i, (L, V, T) in enumerate(STREAM):
            ep = _episode(i, L, V, T)
            with torch.no_grad():                                     # the unpadded eager forward at the CURRENT weights
                ref = run_episode(model, EpisodeTensors(ep, "cuda"), criterion=ops.cross_entropy_sum)
            key = gb.key_for(ep)
            replay = key in seen
            replays += replay
            seen.add(key)
            pad_logits = None
            if replay:                                                # the eager forward on the PADDED batch, same weights
                bufs = gb.buckets[key][0].load(ep)
                gb.head.set_static_plan(bufs.plan)
                with torch.no_grad():
                    pad_logits = [t.clone() for t in run_episode(model, bufs, criterion=ops.cross_entropy_sum)["logits"]]
                gb.head.set_static_plan(None)
            p_before = tr.flat_p.clone()
            loss, logits = gb.step(ep)
            assert abs(float(loss) - float(ref["loss"])) <= tol * max(1.0, abs(float(ref["loss"])))
            for t in range(T):
                a, b = logits[t][:, :V].float(), ref["logits"][t].float()
                fin = torch.isfinite(b)
                assert torch.equal(torch.isfinite(a), fin)
                # against the unpadded run: the imagination tokens sit behind the text padding in the language stream, so the key order
                # of the softmax sums differs - equal up to rounding, not bitwise (the reference's own batches have the same property)
                assert (a[fin] - b[fin]).abs().max().item() <= tol * max(1.0, b[fin].abs().max().item()), (i, t)
                assert bool(torch.isinf(logits[t][:, V:]).all())                                       # padded views can never be chosen
                if pad_logits is not None:
                    assert torch.equal(logits[t], pad_logits[t]), (i, t)                               # replay == eager, bit for bit
            assert not torch.equal(tr.flat_p, p_before)                                              # and the step did train
        assert len(seen) == 3 and replays == len(STREAM) - 3 and tr.step_no == len(STREAM)
    finally:
        tr.close()


def test_bucket_replay_costs_what_a_fixed_shape_replay_costs():
    """Steady state: replaying the bucket graph of a padded batch takes the time of a fixed-shape captured step of the bucket's
    shape (the refill of the static buffers is a handful of small H2D copies)."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.buckets import HamtGraphBuckets
    from vln_imagine_amd.train import FlatTrainer
    cfg = HamtConfig(**HAMT_C1)
    model = build_product(cfg, torch.bfloat16)
    tr = FlatTrainer(model, lr=1e-5)
    try:
        gb = HamtGraphBuckets(tr, model, B, I)
        eps = [_episode(100 + i, 70 + i, 33 + (i % 4), 4) for i in range(6)]               # all land in bucket (80, 37, 4)
        for ep in eps[:2]:
            gb.step(ep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in eps[2:]:
            gb.step(ep)
        torch.cuda.synchronize()
        bucket_ms = (time.perf_counter() - t0) / 4 * 1e3
        et = EpisodeTensors(synth.HamtEpisode(tag="fixed", B=B, L=80, V=37, I=I, T=4, ragged=False), "cuda")

        def fwd_bwd():
            loss = run_episode(model, et, criterion=ops.cross_entropy_sum, keep=False)["loss"]
            loss.backward()
            return loss

        step = tr.capture(fwd_bwd, warmup=1)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        fixed_ms = (time.perf_counter() - t0) / 4 * 1e3
        print(f"\nbucket replay {bucket_ms:.2f} ms/step vs fixed-shape replay {fixed_ms:.2f} ms/step")
        assert bucket_ms <= 1.25 * fixed_ms + 0.5, (bucket_ms, fixed_ms)
    finally:
        tr.close()


def _grads(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def test_lagging_history_tape_equals_step_by_step_autograd():
    """TapedEpisode(lag_history=True) - the order a sampled rollout forces: history of step t - 1 opens step t - gives run_episode's
    logits, loss and gradients (fp32, ragged episode)."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.episode import run_episode_taped
    cfg = HamtConfig(**HAMT_C1)
    model = build_product(cfg)
    et = EpisodeTensors(synth.HamtEpisode(tag="lag", B=4, L=64, V=31, I=I, T=4, ragged=True), "cuda")
    ref = run_episode(model, et, criterion=ops.cross_entropy_sum)
    ref["loss"].backward()
    g_ref = _grads(model)
    model.zero_grad(set_to_none=True)
    seen = []
    out = run_episode_taped(model, et, criterion=ops.cross_entropy_sum, lag_history=True, on_step=lambda t, lg, st: seen.append((lg.clone(), st.clone())))
    out["loss"].backward()
    g_lag = _grads(model)
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) <= 1e-5 * max(1.0, abs(float(ref["loss"].detach())))
    for t in range(et.T):
        fin = torch.isfinite(ref["logits"][t])
        assert torch.equal(torch.isfinite(seen[t][0]), fin)
        assert torch.allclose(seen[t][0][fin], ref["logits"][t][fin], atol=2e-5), t
        assert torch.allclose(seen[t][1], ref["states"][t], atol=2e-5), t
    assert g_ref.keys() == g_lag.keys()
    for n in g_ref:
        assert torch.allclose(g_lag[n], g_ref[n], atol=1e-5 + 1e-4 * float(g_ref[n].abs().max())), n


def test_stepped_episode_graphs_train_like_the_eager_tape():
    """hamt.buckets.SteppedEpisodeGraphs: begin | T step graphs | backward + optimizer, the host writing step t's observation and step
    t - 1's history features only just before step t's replay (what a sampled rollout does) - the same losses, per-step logits and
    parameters as the eager lagging tape on the same stream of episodes."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.buckets import EpisodeBuffers, SteppedEpisodeGraphs
    from vln_imagine_amd.hamt.episode import run_episode_taped
    from vln_imagine_amd.train import FlatTrainer
    cfg = HamtConfig(**HAMT_C1)
    L, V, T = 64, 31, 3
    eps = [synth.HamtEpisode(tag=f"st{i}", B=B, L=L - 3 * i, V=V - i, I=I, T=T, ragged=True) for i in range(4)]
    m_g, m_e = build_product(cfg), build_product(cfg)
    LR = 1e-6           # Adam moves every element by about lr per step whatever its gradient's size: keep the two models' drift below the tolerances
    tr_g, tr_e = FlatTrainer(m_g, lr=LR), FlatTrainer(m_e, lr=LR)
    try:
        p0 = tr_g.flat_p.clone()
        # ---- eager reference: the lagging tape on its own static buffers ----
        bufs_e = EpisodeBuffers(B, L, V, I, T, "cuda")
        losses_e, logits_e = [], []
        head = m_e.contrastive_alignment_model
        for ep in eps:
            bufs_e.load(ep)
            tr_e.zero_grad()
            head.set_static_plan(bufs_e.plan)
            out = run_episode_taped(m_e, bufs_e, criterion=ops.cross_entropy_sum, lag_history=True)
            head.set_static_plan(None)
            out["loss"].backward()
            tr_e.allreduce_grads()
            tr_e.step()
            losses_e.append(float(out["loss"]))
            logits_e.append([t.detach().clone() for t in out["step_logits"]])
        # ---- graphs: the warm-up trains on episode 0, episodes 1.. replay ----
        bufs = EpisodeBuffers(B, L, V, I, T, "cuda").load(eps[0])
        g = SteppedEpisodeGraphs(tr_g, m_g, bufs)
        for i, ep in enumerate(eps[1:], 1):
            bufs.load(ep, steps=False)
            g.begin()
            for t in range(T):
                bufs.put_hist_lens(t, ep.hist_lens[t])
                bufs.put_step(t, ep.steps[t], keys=EpisodeBuffers.OBS_KEYS + ("target",))
                if t > 0:
                    bufs.put_step(t - 1, ep.steps[t - 1], keys=EpisodeBuffers.HIST_KEYS)
                g.step(t)
                a, b = g.logits(t), logits_e[i][t]
                fin = torch.isfinite(b)
                assert torch.equal(torch.isfinite(a), fin) and float((a[fin] - b[fin]).abs().max()) <= 1e-4 * max(1.0, float(b[fin].abs().max())), (i, t)
            bufs.put_step(T - 1, ep.steps[T - 1], keys=EpisodeBuffers.HIST_KEYS)
            loss = g.finish()
            assert abs(float(loss) - losses_e[i]) <= 1e-4 * max(1.0, abs(losses_e[i])), (i, float(loss), losses_e[i])
        assert tr_g.step_no == tr_e.step_no == len(eps)
        # elements whose gradient is rounding noise (atomic summation order differs between a replay and an eager run) may differ by a
        # few lr, the rest agree; and both did train
        d = (tr_g.flat_p - tr_e.flat_p).abs()
        moved = (tr_g.flat_p - p0).abs()
        assert float(d.max()) <= 2 * LR * len(eps) and float(d.mean()) <= 0.05 * float(moved.mean()), (float(d.max()), float(d.mean()), float(moved.mean()))
    finally:
        tr_g.close(); tr_e.close()


@pytest.mark.parametrize("lag", [False, True])
def test_taped_episode_with_ended_samples_equals_step_by_step_autograd(lag):
    """Samples that ended early keep their history length (model_HAMT.py:62-63: later history tokens are masked) and their targets become the ignore
    index (agent_cmt.py:547): the tape - teacher-forced order and the lagging order of a sampled rollout - still gives run_episode's logits, loss
    and gradients."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.episode import run_episode_taped
    cfg = HamtConfig(**HAMT_C1)
    model = build_product(cfg)
    ep = synth.HamtEpisode(tag="ended", B=4, L=64, V=31, I=I, T=4, ragged=True)
    stop = {1: 2, 3: 1}                                            # sample -> the step after which it has ended
    for t in range(ep.T):
        for b, s in stop.items():
            ep.hist_lens[t][b] = min(t + 1, s + 1)
            if t > s:
                ep.steps[t]["target"][b] = -100
    et = EpisodeTensors(ep, "cuda")
    ref = run_episode(model, et, criterion=ops.cross_entropy_sum)
    ref["loss"].backward()
    g_ref = _grads(model)
    model.zero_grad(set_to_none=True)
    out = run_episode_taped(model, et, criterion=ops.cross_entropy_sum, lag_history=lag)
    out["loss"].backward()
    g_tape = _grads(model)
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) <= 1e-5 * max(1.0, abs(float(ref["loss"].detach())))
    for t in range(et.T):
        fin = torch.isfinite(ref["logits"][t])
        assert torch.equal(torch.isfinite(out["step_logits"][t]), fin)
        assert torch.allclose(out["step_logits"][t][fin], ref["logits"][t][fin], atol=2e-5), t
    assert g_ref.keys() == g_tape.keys()
    for n in g_ref:
        assert torch.allclose(g_tape[n], g_ref[n], atol=1e-5 + 1e-4 * float(g_ref[n].abs().max())), n


@pytest.mark.parametrize("want_states", [True, False])
def test_stepped_inference_graphs_give_the_eager_forward(want_states):
    """hamt.buckets.SteppedInferenceGraphs (validation rollouts: forward only, begin | T step graphs, host between the steps): the logits of
    the eager no_grad rollout on the same padded buffers, for a stream of episodes through ONE capture - and for several captures in one
    process (allocation patterns differ between them; with and without the extra state output)."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.buckets import EpisodeBuffers, SteppedInferenceGraphs
    cfg = HamtConfig(**HAMT_C1)
    L, V = 64, 31
    model = build_product(cfg)
    for T in (2, 3, 3):
        eps = [synth.HamtEpisode(tag=f"inf{T}_{i}", B=B, L=L - 5 * i, V=V - 2 * i, I=I, T=T, ragged=True) for i in range(3)]
        bufs = EpisodeBuffers(B, L, V, I, T, "cuda").load(eps[0])
        g = SteppedInferenceGraphs(model, bufs, want_states=want_states)
        for ep in eps:
            with torch.no_grad():
                ref = run_episode(model, EpisodeBuffers(B, L, V, I, T, "cuda").load(ep), use_aux=False, criterion=ops.cross_entropy_sum)
            bufs.load(ep, steps=False)
            g.begin()
            for t in range(T):
                bufs.put_hist_lens(t, ep.hist_lens[t])
                bufs.put_step(t, ep.steps[t], keys=EpisodeBuffers.OBS_KEYS)
                if t > 0:
                    bufs.put_step(t - 1, ep.steps[t - 1], keys=EpisodeBuffers.HIST_KEYS)
                g.step(t)
                a, b = g.logits(t), ref["logits"][t]
                fin = torch.isfinite(b)
                assert torch.equal(torch.isfinite(a), fin) and torch.allclose(a[fin], b[fin], atol=2e-5), (T, ep.L, t)
                if want_states:
                    assert torch.allclose(g.state(t), ref["states"][t], atol=2e-5), (T, ep.L, t)


def test_history_lengths_changed_mid_rollout_on_episode_tensors():
    """A sampled rollout on plain EpisodeTensors (no static buffers): the caller ends samples early with et.put_hist_lens(t, lens) before
    step t (writing et.hist_lens_dev alone used to be silently ignored). The lagging tape then gives the logits of the
    step-by-step run on an episode that was BUILT with those lengths."""
    import copy
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.episode import TapedEpisode
    cfg = HamtConfig(**HAMT_C1)
    model = build_product(cfg)
    ep = synth.HamtEpisode(tag="lens", B=4, L=64, V=31, I=I, T=4, ragged=False)
    new_lens = [list(l) for l in ep.hist_lens]
    for t in range(2, ep.T):                          # samples 1 and 3 end after step 1: their history stops growing (model_HAMT.py:62-63)
        new_lens[t][1] = new_lens[t][3] = 2
    ep2 = copy.copy(ep)
    ep2.hist_lens = new_lens
    with torch.no_grad():
        ref = run_episode(model, EpisodeTensors(ep2, "cuda"), criterion=ops.cross_entropy_sum)
    et = EpisodeTensors(ep, "cuda")
    te = TapedEpisode(model, et, criterion=ops.cross_entropy_sum, lag_history=True)
    te.begin()
    for t in range(ep.T):
        et.put_hist_lens(t, new_lens[t])              # known only now in a real rollout
        lg, _ = te.step(t)
        fin = torch.isfinite(ref["logits"][t])
        assert torch.equal(torch.isfinite(lg), fin) and torch.allclose(lg[fin], ref["logits"][t][fin], atol=2e-5), t
    out = te.finish()
    assert abs(float(out["loss"].detach()) - float(ref["loss"])) <= 1e-5 * max(1.0, abs(float(ref["loss"])))


def test_history_mask_computed_inside_the_captured_steps():
    """Regression test for the round-3 anomaly (with `arange < lengths[t]` computed INSIDE a step's graph some capture
    sequences were seen replaying step 0's mask at later steps). The forward-only stepped graphs with mask_in_graph=True, five captures
    in one process sharing the model, three episodes through each: every step's logits are the eager rollout's."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.buckets import EpisodeBuffers, SteppedInferenceGraphs
    cfg = HamtConfig(**HAMT_C1)
    L, V = 64, 31
    model = build_product(cfg)
    for cap, T in enumerate((2, 3, 3, 4, 3)):
        eps = [synth.HamtEpisode(tag=f"ing{T}_{i}", B=B, L=L - 5 * i, V=V - 2 * i, I=I, T=T, ragged=True) for i in range(3)]
        bufs = EpisodeBuffers(B, L, V, I, T, "cuda").load(eps[0])
        g = SteppedInferenceGraphs(model, bufs, want_states=bool(cap & 1), mask_in_graph=True)
        for ep in eps:
            with torch.no_grad():
                ref = run_episode(model, EpisodeBuffers(B, L, V, I, T, "cuda").load(ep), use_aux=False, criterion=ops.cross_entropy_sum)
            bufs.load(ep, steps=False)
            g.begin()
            for t in range(T):
                bufs.put_hist_lens(t, ep.hist_lens[t])
                bufs.put_step(t, ep.steps[t], keys=EpisodeBuffers.OBS_KEYS)
                if t > 0:
                    bufs.put_step(t - 1, ep.steps[t - 1], keys=EpisodeBuffers.HIST_KEYS)
                g.step(t)
                a, b = g.logits(t), ref["logits"][t]
                fin = torch.isfinite(b)
                assert torch.equal(torch.isfinite(a), fin) and torch.allclose(a[fin], b[fin], atol=2e-5), (cap, T, ep.L, t)
