"""FlatTrainer against torch.optim.AdamW: the three parameter groups and the stage table of the shipped "variant4" warm-up
(VLN-HAMT/finetune_src/r2r/agent_cmt.py:82-96, r2r/main.py:202-255), optimizer checkpoints (agent_cmt.py:837-870), and the
scoping of its gradient-accumulation marks to its own parameters."""
import pytest
import torch

from tests.golden.variants import hamt_variant_setup
from tests.test_hamt_gpu import build_product
from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode

pytestmark = pytest.mark.gpu
LR = 1e-3
# (contrastive + imagine lr, rest lr, rest trainable) per stage: r2r/main.py:209-212 with args.lr = LR
STAGES = [(LR * 10, None, False), (LR * 5, LR * 0.1, True), (LR * 0.1, LR * 0.1, True)]


def _groups(m):
    a = list(m.contrastive_alignment_model.parameters())
    b = list(m.imagine_embeddings.parameters())
    skip = {id(p) for p in a + b}
    rest = [p for p in m.parameters() if id(p) not in skip]
    return a, b, rest


def test_param_groups_follow_the_variant4_stage_table():
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_language")
    et = EpisodeTensors(ep, "cuda")
    ref, m = build_product(cfg), build_product(cfg)
    ra, rb, rrest = _groups(ref)
    opt = torch.optim.AdamW([{"params": ra, "lr": LR}, {"params": rb, "lr": LR}, {"params": rrest, "lr": LR}], weight_decay=0.01)
    a, b, rest = _groups(m)
    tr = FlatTrainer(m, lr=LR, groups=[{"params": a, "name": "contrastive_alignment_model"},
                                       {"params": b, "name": "imagine_embeddings"}, {"params": rest, "name": "rest"}])
    try:
        for lr_new, lr_rest, rest_on in STAGES:
            for p in rrest:
                p.requires_grad_(rest_on)
            opt.param_groups[0]["lr"] = opt.param_groups[1]["lr"] = lr_new
            if lr_rest is not None:
                opt.param_groups[2]["lr"] = lr_rest
            tr.set_group("contrastive_alignment_model", lr=lr_new)
            tr.set_group("imagine_embeddings", lr=lr_new)
            tr.set_group("rest", lr=lr_rest, trainable=rest_on)
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                run_episode(ref, et, criterion=ops.cross_entropy_sum)["loss"].backward()
                torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 40.0)
                opt.step()
                tr.zero_grad()
                run_episode(m, et, criterion=ops.cross_entropy_sum)["loss"].backward()
                tr.step()
            worst = 0.0
            for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
                worst = max(worst, (p - q).abs().max().item())
                if not rest_on and id(q) in {id(x) for x in rrest}:
                    assert torch.equal(p, q), n                   # a frozen group is not touched at all (no weight decay either)
            # Adam turns rounding noise on ~zero gradients into +-lr: the bound is a few learning rates, the typical error far lower
            assert worst < 6 * 2 * lr_new, (lr_new, worst)
            mean = torch.cat([(p - q).abs().reshape(-1) for p, q in zip(m.parameters(), ref.parameters())]).mean().item()
            assert mean < 2e-5, mean
        # per-group step counts: the rest group sat out stage 1
        sd, sr = tr.state_dict(), opt.state_dict()
        assert [len(g["params"]) for g in sd["param_groups"]] == [len(g["params"]) for g in sr["param_groups"]]
        i_rest = sd["param_groups"][2]["params"][0]
        assert float(sd["state"][i_rest]["step"]) == float(sr["state"][i_rest]["step"]) == 4.0
        assert float(sd["state"][0]["step"]) == float(sr["state"][0]["step"]) == 6.0
    finally:
        tr.close()


def test_optimizer_state_dict_round_trip():
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_language")
    et = EpisodeTensors(ep, "cuda")

    def steps(tr, m, n):
        for _ in range(n):
            tr.zero_grad()
            run_episode(m, et, criterion=ops.cross_entropy_sum)["loss"].backward()
            tr.step()

    m1 = build_product(cfg)
    t1 = FlatTrainer(m1, lr=LR)
    steps(t1, m1, 2)
    sd_opt, sd_model = t1.state_dict(), {k: v.clone() for k, v in m1.state_dict().items()}
    steps(t1, m1, 1)
    final1 = t1.flat_p.clone()
    t1.close()
    m2 = build_product(cfg)
    m2.load_state_dict(sd_model)
    t2 = FlatTrainer(m2, lr=LR)
    try:
        t2.load_state_dict(sd_opt)
        assert t2.step_no == 2 and float(t2.gstate[0]) == 2.0
        steps(t2, m2, 1)
        d = (t2.flat_p - final1).abs()
        # moments and bias corrections resumed: a restart from zero moments moves EVERY element by ~lr in step 3; what is left is
        # Adam's amplification of atomics-order noise on ~zero gradients
        assert d.max().item() < 5e-4 and d.mean().item() < 1e-6, (d.max().item(), d.mean().item())
        # and the layout is what torch.optim.AdamW loads
        plist = list(m2.parameters())
        opt = torch.optim.AdamW(plist, lr=LR)
        tmpl = opt.state_dict()
        tmpl["state"] = sd_opt["state"]
        opt.load_state_dict(tmpl)
        assert torch.equal(opt.state[plist[3]]["exp_avg"].cpu(), sd_opt["state"][3]["exp_avg"].cpu())
    finally:
        t2.close()


def test_optimizer_checkpoints_of_a_model_with_frozen_parameters_travel_both_ways():
    """The shipped run passes --fix_lang_embedding (update_lang_bert=False): the reference builds AdamW from ALL vln_bert.parameters()
    (r2r/agent_cmt.py:98), so indices count the frozen tensors too and only the optimised ones have state. FlatTrainer(groups=None) keeps
    that index space: a torch.optim.AdamW state_dict loads into it, and its own state_dict loads into torch.optim.AdamW."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_shipped")
    et = EpisodeTensors(ep, "cuda")
    ref, m = build_product(cfg), build_product(cfg)
    frozen = [i for i, p in enumerate(ref.parameters()) if not p.requires_grad]
    assert frozen and len(frozen) < len(list(ref.parameters()))
    opt = torch.optim.AdamW(ref.parameters(), lr=LR, weight_decay=0.01)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        run_episode(ref, et, criterion=ops.cross_entropy_sum)["loss"].backward()
        torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 40.0)
        opt.step()
    sr = opt.state_dict()
    assert not any(i in sr["state"] for i in frozen)                       # torch skips parameters without a gradient
    m.load_state_dict(ref.state_dict())
    flags = [p.requires_grad for p in m.parameters()]
    tr = FlatTrainer(m, lr=LR)
    try:
        assert [p.requires_grad for p in m.parameters()] == flags           # the constructor unfreezes nothing
        assert all(id(p) not in tr._off for p, f in zip(m.parameters(), flags) if not f)
        tr.load_state_dict(sr)
        plist = list(m.parameters())
        live = [i for i, f in enumerate(flags) if f and i in sr["state"]]
        for i in (live[0], live[len(live) // 2], live[-1]):
            o, nel = tr._off[id(plist[i])], plist[i].numel()
            assert torch.equal(tr.m[o:o + nel].view(plist[i].shape), sr["state"][i]["exp_avg"])
            assert torch.equal(tr.v[o:o + nel].view(plist[i].shape), sr["state"][i]["exp_avg_sq"])
        assert tr.step_no == 2
        # one more step on both sides from the loaded state: the replicas stay together
        opt.zero_grad(set_to_none=True)
        run_episode(ref, et, criterion=ops.cross_entropy_sum)["loss"].backward()
        torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 40.0)
        opt.step()
        tr.zero_grad()
        run_episode(m, et, criterion=ops.cross_entropy_sum)["loss"].backward()
        tr.step()
        worst = max((p - q).abs().max().item() for p, q in zip(m.parameters(), ref.parameters()))
        assert worst < 4 * LR, worst
        for i in frozen:
            assert torch.equal(plist[i], list(ref.parameters())[i])           # frozen tensors: untouched on both sides
        # and back: this trainer's state_dict is what torch.optim.AdamW(model.parameters()) loads
        sd = tr.state_dict()
        assert sd["param_groups"][0]["params"] == sr["param_groups"][0]["params"]
        assert set(sd["state"]) == set(opt.state_dict()["state"])
        opt2 = torch.optim.AdamW(ref.parameters(), lr=LR, weight_decay=0.01)
        opt2.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
        rp = list(ref.parameters())
        assert torch.allclose(opt2.state[rp[live[0]]]["exp_avg"], opt.state[rp[live[0]]]["exp_avg"], rtol=0, atol=2e-5)
        assert float(opt2.state[rp[live[0]]]["step"]) == 3.0
    finally:
        tr.close()


def test_marks_are_scoped_to_the_trainers_parameters():
    """A second model in the process keeps plain autograd accumulation while a FlatTrainer is alive."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_language")
    et = EpisodeTensors(ep, "cuda")
    ma, mb, mc = build_product(cfg), build_product(cfg), build_product(cfg)
    tr = FlatTrainer(ma)
    try:
        run_episode(mb, et, criterion=ops.cross_entropy_sum)["loss"].backward()      # trainer alive, model b is not its model
        assert not ops._WQ
    finally:
        tr.close()
    run_episode(mc, et, criterion=ops.cross_entropy_sum)["loss"].backward()
    for (n, p), (_, q) in zip(mb.named_parameters(), mc.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            assert torch.equal(p.grad, q.grad) or (p.grad - q.grad).abs().max().item() <= 1e-6 * max(1.0, q.grad.abs().max().item()), n


def test_dynamic_loss_scaling_is_transparent_and_skips_overflowed_steps():
    """GradScaler semantics of the fused step (VLN-DUET/pretrain_src/train_r2r.py:201-234; what an fp16 run needs): the loss is
    multiplied by the device-resident scale S before backward and the step divides it out again (same update as without scaling);
    a non-finite gradient skips the step entirely and halves S; `growth_interval` clean steps double it."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_language")
    et = EpisodeTensors(ep, "cuda")
    ma, mb = build_product(cfg), build_product(cfg)
    ta = FlatTrainer(ma, lr=LR)
    ta.zero_grad()
    run_episode(ma, et, criterion=ops.cross_entropy_sum)["loss"].backward()
    ta.step()
    pa, na = ta.flat_p.clone(), ta.grad_norm()
    ta.close()
    tb = FlatTrainer(mb, lr=LR, loss_scale=1024.0, growth_interval=2)
    try:
        p0 = tb.flat_p.clone()
        tb.zero_grad()
        (run_episode(mb, et, criterion=ops.cross_entropy_sum)["loss"] * tb.loss_scale).backward()
        tb.step()
        assert abs(tb.grad_norm() - na) < 1e-4 * na and float(tb.state[5]) == 0.0 and float(tb.state[4]) == 1024.0
        d = (tb.flat_p - pa).abs()
        assert d.max().item() < 2.5e-3 and d.mean().item() < 1e-6, (d.max().item(), d.mean().item())      # same step (power-of-two scale)
        p1, m1 = tb.flat_p.clone(), tb.m.clone()
        tb.zero_grad()
        (run_episode(mb, et, criterion=ops.cross_entropy_sum)["loss"] * tb.loss_scale).backward()
        tb.flush()
        tb.flat_g[12345] = float("inf")                               # an overflowed fp16 gradient
        tb.step()
        assert float(tb.state[5]) == 1.0 and float(tb.state[4]) == 512.0                  # skipped, scale halved
        assert torch.equal(tb.flat_p, p1) and torch.equal(tb.m, m1) and float(tb.gstate[0]) == 1.0
        for _ in range(2):                                                               # two clean steps: scale doubles again
            tb.zero_grad()
            (run_episode(mb, et, criterion=ops.cross_entropy_sum)["loss"] * tb.loss_scale).backward()
            tb.step()
        assert float(tb.state[4]) == 1024.0 and float(tb.gstate[0]) == 3.0 and not torch.equal(tb.flat_p, p1)
        assert (tb.flat_p - p0).abs().max().item() > 1e-4
    finally:
        tb.close()


def test_float16_training_with_dynamic_loss_scaling_tracks_float32():
    """BASELINE.json configs[4] names fp16: float16 compute (float32 masters, a float16 mirror written by the optimizer kernel) under
    the GradScaler-style dynamic loss scale keeps up with the float32 path over a few steps; no step is skipped at S = 2^12."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_language")
    et = EpisodeTensors(ep, "cuda")
    m32, m16 = build_product(cfg), build_product(cfg, torch.float16)
    t32 = FlatTrainer(m32, lr=1e-4)
    losses32 = []
    for _ in range(3):
        t32.zero_grad()
        l = run_episode(m32, et, criterion=ops.cross_entropy_sum)["loss"]
        l.backward()
        t32.step()
        losses32.append(float(l))
    p32 = t32.flat_p.clone()
    t32.close()
    t16 = FlatTrainer(m16, lr=1e-4, loss_scale=4096.0, growth_interval=1000)
    try:
        assert t16.flat_b.dtype == torch.float16
        for i in range(3):
            t16.zero_grad()
            l = run_episode(m16, et, criterion=ops.cross_entropy_sum)["loss"]
            (l * t16.loss_scale).backward()
            t16.step()
            assert float(t16.state[5]) == 0.0                                  # not skipped
            assert abs(float(l) - losses32[i]) < 2e-2, (i, float(l), losses32[i])
        d = (t16.flat_p - p32).abs()
        assert d.mean().item() < 2e-5 and d.max().item() < 1e-3, (d.mean().item(), d.max().item())     # 3 steps at lr 1e-4
    finally:
        t16.close()


def test_embedding_rows_no_token_touches_are_the_one_documented_deviation_from_torch_adamw():
    """the fused step leaves an ELEMENT alone while its g, m and v are all exactly zero (that is how it tells a parameter torch would
    skip for grad None - an unused head, embeddings behind a detach - from a live one, without a per-parameter flag). For the word-embedding table the
    rule also catches the rows no token of any batch so far has touched: torch.optim.AdamW decays those too, by (1 - lr wd) per step. This test pins
    the size of that deviation (exactly the missing decay, lr x wd x steps x |w| = 3e-5 |w| here; 1e-7 |w| per step at the reference's lr 1e-5) and
    that touched rows follow torch."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_language")
    et = EpisodeTensors(ep, "cuda")
    ref, m = build_product(cfg), build_product(cfg)
    w0 = m.embeddings.word_embeddings.weight.detach().clone()
    wd, steps = 0.01, 3
    opt = torch.optim.AdamW(ref.parameters(), lr=LR, weight_decay=wd)
    tr = FlatTrainer(m, lr=LR)
    try:
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            run_episode(ref, et, criterion=ops.cross_entropy_sum)["loss"].backward()
            torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 40.0)
            opt.step()
            tr.zero_grad()
            run_episode(m, et, criterion=ops.cross_entropy_sum)["loss"].backward()
            tr.step()
        used = torch.zeros(w0.shape[0], dtype=torch.bool, device="cuda")
        used[et.txt_ids.reshape(-1)] = True
        assert int((~used).sum()) > 1000 and int(used.sum()) > 10
        wp, wt = m.embeddings.word_embeddings.weight.detach(), ref.embeddings.word_embeddings.weight.detach()
        assert torch.equal(wp[~used], w0[~used])                                               # untouched rows: left alone here ...
        decay = (1.0 - LR * wd) ** steps
        assert torch.allclose(wt[~used], w0[~used] * decay, rtol=0, atol=1e-7)                 # ... decayed by torch: the whole deviation
        dev = (wp[~used] - wt[~used]).abs().max().item()
        assert dev <= 1.01 * (1.0 - decay) * w0[~used].abs().max().item()
        assert (wp[used] - wt[used]).abs().max().item() < 4 * 2 * LR                            # touched rows: torch's update (Adam on rounding noise: a few lr)
    finally:
        tr.close()


def test_store_mode_reduction_equals_the_zero_filled_accumulation():
    """ops.GradArena: from the second step on the arena's zero fill leaves out what the batched partial reduction will write, the reduction
    stores instead of adding and leaves the stored gradients' sum of squares for clip_grad_norm_ (r2r/agent_cmt.py:829). Gradients, norms and
    weights must equal the zero-fill + add + separate-norm program (VLNI_STORE_PARTS=0) step by step - through a step in which a module gets
    no gradient (its stale range is zero-filled by the reduction), a step with two backward passes (the second adds; the norm falls back to
    the arena pass), an explicit flush() before step() (gradients edited in between are seen) and a captured step."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    from vln_imagine_amd import synth
    cfg, _ = hamt_variant_setup("c1_language")
    # rows enough for the weight-gradient launches to split (only split launches go through the partial reduction)
    et = EpisodeTensors(synth.HamtEpisode(tag="store", B=32, L=80, V=37, I=4, T=3, ragged=True), "cuda")

    def program(store, kinds=("plain", "plain", "no_aux", "plain", "twice", "flushed", "plain"), warmup=0):
        was = ops.STORE_PARTS
        ops.STORE_PARTS = store
        ops.reseed(11)
        m = build_product(cfg, torch.bfloat16)                 # 16-bit operands: the weight gradients that are queued, grouped and row-split
        tr = FlatTrainer(m, lr=0.0, weight_decay=0.0)          # the weights stay put: every step's gradients are comparable on their own
        out = []

        def bwd(**kw):
            loss = run_episode(m, et, criterion=ops.cross_entropy_sum, **kw)["loss"]
            loss.backward()
            return loss.detach()
        try:
            for kind in kinds:
                tr.zero_grad()
                stored = sum(ops.GRADS.pending.values())
                bwd(use_aux=kind != "no_aux")
                if kind == "twice":
                    tr.flush()
                    bwd()
                if kind == "flushed":
                    tr.flush()
                    tr.flat_g[::1001] *= 3.0
                tr.step()
                out.append((kind, tr.flat_g.clone(), tr.grad_norm(), stored, (len(ops.GRADS.seen), len(ops.GRADS.never), len(ops._PART_TABLES))))
            step = tr.capture(bwd, warmup=warmup)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            out.append(("graph", tr.flat_g.clone(), tr.grad_norm(), tr.n, None))
        finally:
            tr.close()
            ops.STORE_PARTS = was
        return out

    ref, got = program(False), program(True)
    assert all(r[3] == 0 for r in ref[:-1])
    assert got[0][3] == 0 and all(g[3] > 0.5 * got[0][1].numel() for g in got[1:]), [g[3:] for g in got]     # most of the arena is never zero-filled
    # nothing of a default step is both queued for the reduction and added into directly (the cross-attention blocks' capability
    # check used to move every one of their projection gradients out of the stored ranges for good)
    assert got[0][4][1] == 0 and got[1][4][1] == 0, [g[4] for g in got[:-1]]
    for (kind, g0, n0, _, _), (_, g1, n1, _, _) in zip(ref, got):
        scale = g0.abs().max().item()
        assert (g0 - g1).abs().max().item() <= 2e-5 * scale, (kind, (g0 - g1).abs().max().item(), scale)
        assert n0 > 0 and abs(n0 - n1) <= 2e-5 * n0, (kind, n0, n1)
    # a capture straight after ONE eager step: the captured step is the first to store, its launch tables were pre-built by that eager step
    # (or are uploaded through the pinned staging buffer as nodes of the graph)
    (_, g0, n0, _, _), (_, g1, n1, _, _) = program(False, (), 1)[-1], program(True, (), 1)[-1]
    assert (g0 - g1).abs().max().item() <= 2e-5 * g0.abs().max().item() and abs(n0 - n1) <= 2e-5 * n0, (n0, n1)
    kinds = {r[0]: r for r in ref}
    assert kinds["twice"][2] > 1.5 * kinds["plain"][2] and (kinds["no_aux"][1] - kinds["plain"][1]).abs().max().item() > 0
