"""Round 6: where the 16-bit paths' distance from the float32 path comes from, and whether it matters for training.

  python tools/bf16_ablation.py ablate <tag>     one line of the table: the timed program (HAMT 9L+4X+2pano, B = 64, T = 6, eval mode so both sides see
                                                 the same arithmetic, episode tape + one batched backward) in VLNI_ABL_DTYPE (bf16 | fp16) against the float32
                                                 path at the same weights - |d loss|, max |d logit|, gradient rel-L2, worst parameter - plus ms per step of
                                                 the captured train-mode step. The environment switches under test are read at import: one process per line.
  python tools/bf16_ablation.py train <dtype>    300 optimizer steps (FlatTrainer, lr 5e-5, clip 40, dropout on) over 8 fixed synthetic batches of 16 episodes at
                                                 full depth; prints the loss of every step as JSON (fp32 | bf16 | fp16: same seeds, same batches, same init).
  python tools/bf16_ablation.py report <dir>     profiles/r06_bf16_ablation.md from the lines collected under <dir>."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _args(B, T=6):
    return argparse.Namespace(batch=B, T=T, L=80, V=37, I=6)


def ablate(tag):
    import torch
    import bench
    from vln_imagine_amd import ops
    from vln_imagine_amd.compare import compare_runs
    from vln_imagine_amd.train import FlatTrainer
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("VLNI_ABL_DTYPE", "bf16")]
    B = int(os.environ.get("VLNI_ABL_BATCH", "64"))
    mode = os.environ.get("VLNI_ABL_MODE", "taped")
    dev = torch.device("cuda")
    w = bench.Workload("hamt", _args(B), False, dev, dt, batch=B, tag="abl")
    w32 = w.build(dev, torch.float32)
    w32.load_state_dict(w.model.state_dict())
    scale = 16384.0 if dt == torch.float16 else 1.0
    o16 = w.run(criterion=ops.cross_entropy_sum, keep=True, mode=mode)
    (o16["loss"] * scale).backward()
    if scale != 1.0:
        for p in w.model.parameters():
            if p.grad is not None:
                p.grad.mul_(1.0 / scale)
    o32 = w.run(criterion=ops.cross_entropy_sum, keep=True, model=w32)
    o32["loss"].backward()
    if mode == "taped":
        o16 = dict(o16, logits=o16["step_logits"])
    res = compare_runs(o16, o32, dict(w.model.named_parameters()), dict(w32.named_parameters()), "logits")
    # per parameter group (text encoder / cross-modal layers / history + observation embeddings / heads)
    groups = {"text encoder": ("embeddings.", "encoder.layer."), "cross-modal layers": ("encoder.x_layers.",),
              "history / observation / imagination embeddings": ("hist_embeddings.", "img_embeddings.", "imagine_embeddings."),
              "heads": ("next_action.", "contrastive_alignment_model.")}
    p32 = dict(w32.named_parameters())
    per = {}
    for g, pre in groups.items():
        num = den = 0.0
        for n, p in w.model.named_parameters():
            if n.startswith(pre) and p.grad is not None and p32[n].grad is not None:
                num += float((p.grad.double() - p32[n].grad.double()).pow(2).sum())
                den += float(p32[n].grad.double().pow(2).sum())
        per[g] = round((num / den) ** 0.5, 4) if den > 0 else None
    del w32, o16, o32
    for p in w.model.parameters():
        p.grad = None
    # ms per step of the timed program under the same switches (train mode, captured)
    w.model.train()
    kw = dict(loss_scale=16384.0, growth_interval=2000) if dt == torch.float16 else {}
    tr = FlatTrainer(w.model, lr=1e-5, **kw)
    tape = ops.EpisodeTape(w.T)

    def fwd_bwd():
        from vln_imagine_amd.hamt.episode import run_episode_taped
        o = run_episode_taped(w.model, w.et, tape=tape, criterion=ops.cross_entropy_sum)
        (o["loss"] * tr.loss_scale if dt == torch.float16 else o["loss"]).backward()
        return o["loss"]
    ms = None
    if B == 64 and mode == "taped":
        step = tr.capture(fwd_bwd, warmup=2)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ms = round((time.perf_counter() - t0) / 20 * 1e3, 2)
    tr.close()
    sw = {k: v for k, v in os.environ.items() if k.startswith("VLNI_") and k not in ("VLNI_ABL_DTYPE", "VLNI_ABL_BATCH", "VLNI_ABL_MODE")}
    print(json.dumps({"tag": tag, "dtype": str(dt).replace("torch.", ""), "batch": B, "mode": mode, "switches": sw,
                      "loss_abs": res["loss_abs"], "logit_max_abs": res["logit_max_abs"], "grad_rel_l2": res["grad_rel_l2"],
                      "worst": res["grad_worst_param_rel_l2"], "worst_name": res["grad_worst_param"], "groups": per, "ms_per_step": ms}), flush=True)


def train(dtype_name):
    import torch
    import bench
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode_taped
    from vln_imagine_amd import synth
    from vln_imagine_amd.train import FlatTrainer
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype_name]
    dev = torch.device("cuda")
    B, T, NB, STEPS = 16, 6, 8, int(os.environ.get("VLNI_ABL_STEPS", "300"))
    w = bench.Workload("hamt", _args(B, T), False, dev, dt, batch=B, tag="trn0")
    ets = [w.et] + [EpisodeTensors(synth.HamtEpisode(tag=f"trn{i}", B=B, L=80, V=37, I=6, T=T, ragged=False), dev) for i in range(1, NB)]
    w.model.train()
    ops.reseed(1234)
    torch.manual_seed(7)
    kw = dict(loss_scale=16384.0, growth_interval=2000) if dt == torch.float16 else {}
    tr = FlatTrainer(w.model, lr=5e-5, **kw)
    losses = []
    for s in range(STEPS):
        tr.zero_grad()
        o = run_episode_taped(w.model, ets[s % NB], criterion=ops.cross_entropy_sum)
        (o["loss"] * tr.loss_scale if dt == torch.float16 else o["loss"]).backward()
        tr.step()
        losses.append(o["loss"].detach())
    torch.cuda.synchronize()
    print(json.dumps({"train": dtype_name, "steps": STEPS, "batches": NB, "batch": B, "loss": [round(float(x), 5) for x in losses]}), flush=True)
    tr.close()


def report(d):
    rows, curves = [], {}
    for f in sorted(os.listdir(d)):
        if not f.endswith(".json"):
            continue
        for line in open(os.path.join(d, f)):
            line = line.strip()
            if not line.startswith("{"):
                continue
            j = json.loads(line)
            if "train" in j:
                curves[j["train"]] = j
            else:
                rows.append(j)
    print("# 16-bit paths against the float32 path: what each switch buys (r06)\n")
    print("`tools/bf16_ablation.py`: HAMT-Imagine 9L + 4X + 2pano, B = 64, T = 6 (BASELINE.json configs[1]), the timed program (episode tape, ONE batched "
          "backward) in eval mode against the float32 MFMA path (the parity path held to the reference goldens at 1e-4) at the same weights; `ms / step` = "
          "the captured train-mode step under the same switches, same box, one process per line.\n")
    print("| configuration | switches | d loss | max d logit | gradient rel-L2 | worst parameter | text encoder | cross-modal | embeddings | heads | ms / step |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for j in rows:
        g = j["groups"]
        sw = ", ".join(f"{k}={v}" for k, v in sorted(j["switches"].items())) or "-"
        print(f"| {j['tag']} ({j['dtype']}, B = {j['batch']}, {j['mode']}) | {sw} | {j['loss_abs']:.2e} | {j['logit_max_abs']:.4f} | {j['grad_rel_l2']:.4f} | "
              f"{j['worst']:.3f} ({j['worst_name']}) | {g['text encoder']} | {g['cross-modal layers']} | {g['history / observation / imagination embeddings']} | "
              f"{g['heads']} | {j['ms_per_step'] if j['ms_per_step'] is not None else '-'} |")
    if curves:
        print("\n## 300 optimizer steps: fp32 vs bf16 vs fp16 through `FlatTrainer` (8 fixed synthetic batches of 16 episodes, full depth, lr 5e-5, clip 40, "
              "dropout 0.1, same seeds)\n")
        names = [n for n in ("fp32", "bf16", "fp16") if n in curves]
        print("| step | " + " | ".join(names) + " |")
        print("|---|" + "---|" * len(names))
        n = curves[names[0]]["steps"]
        nb = curves[names[0]]["batches"]
        for s0 in list(range(0, n, 25)) + [n - nb]:
            avg = lambda c: sum(c["loss"][s0:s0 + nb]) / len(c["loss"][s0:s0 + nb])
            print(f"| {s0}-{s0 + nb - 1} (mean over the {nb} batches) | " + " | ".join(f"{avg(curves[k]):.4f}" for k in names) + " |")
        ref = curves.get("fp32")
        if ref:
            for k in names[1:]:
                dev_ = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(curves[k]["loss"], ref["loss"]))
                tail = sum(curves[k]["loss"][-nb:]) / nb - sum(ref["loss"][-nb:]) / nb
                print(f"\n{k} against fp32: largest relative difference of a step's loss over the run {dev_:.3f}; mean loss of the last {nb} steps differs by {tail:+.4f}.")


if __name__ == "__main__":
    {"ablate": ablate, "train": train, "report": report}[sys.argv[1]](sys.argv[2])
