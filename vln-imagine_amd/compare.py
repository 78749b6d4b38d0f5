"""Error of one run of an episode against another run of the same episode (bf16 path vs fp32 path, HIP path vs CPU oracle):
|d loss|, max |d logit| over the finite logits, relative L2 error of the whole gradient and the worst single parameter.
Used by bench.py (`bf16_vs_fp32`) and by the full-depth parity tests; pure bookkeeping, no kernels."""
import torch


def _grad_of(p):
    return None if p.grad is None else p.grad.detach().double().cpu()


def compare_runs(out_a, out_b, params_a, params_b, logits_key="logits"):
    """out_*: dicts of run_episode (loss + per-step logits under `logits_key`); params_*: name -> parameter (with .grad).
    b is the reference side of every relative figure."""
    res = {"loss_abs": abs(float(out_a["loss"].detach()) - float(out_b["loss"].detach()))}
    worst = 0.0
    for a, b in zip(out_a[logits_key], out_b[logits_key]):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        fin = torch.isfinite(b)
        if not bool((torch.isfinite(a) == fin).all()):
            worst = float("inf")                          # a masked (-inf) logit on one side only
        elif bool(fin.any()):
            worst = max(worst, float((a[fin] - b[fin]).abs().max()))
    res["logit_max_abs"] = worst
    pairs = []
    for name, pb in params_b.items():
        gb = _grad_of(pb)
        ga = _grad_of(params_a[name]) if name in params_a else None
        if gb is None or float(gb.abs().max()) == 0.0:
            assert ga is None or float(ga.abs().max()) == 0.0, f"{name}: gradient on one side only"
            continue
        assert ga is not None, f"{name}: gradient missing"
        pairs.append((name, float((ga - gb).pow(2).sum()), float(gb.pow(2).sum())))
    num, den, n_cmp = sum(d for _, d, _ in pairs), sum(n for _, _, n in pairs), len(pairs)
    # per parameter: relative to its own norm, floored at 1e-3 of the whole gradient's norm - gradients that are zero in exact
    # arithmetic (the key bias of a softmax attention) are rounding noise on both sides and have no meaningful relative error
    floor = 1e-6 * den
    worst_p, worst_name = 0.0, None
    for name, d2, n2 in pairs:
        rel = (d2 / max(n2, floor)) ** 0.5
        if rel > worst_p:
            worst_p, worst_name = rel, name
    res["grad_rel_l2"] = (num / den) ** 0.5 if den > 0 else 0.0
    res["grad_worst_param_rel_l2"] = worst_p
    res["grad_worst_param"] = worst_name
    res["params_compared"] = n_cmp
    return res
