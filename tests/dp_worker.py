"""Child process of tests/test_dp_gpu.py (started by tests/conftest.py BEFORE the pytest process touches the GPU).

  python -m tests.dp_worker <outdir> rank      RANK / WORLD_SIZE / MASTER_* from the environment; gloo, every rank on cuda:0
  python -m tests.dp_worker <outdir> single    the single-process answer the ranks must reproduce

What is checked afterwards (data-parallel semantics of VLN-HAMT/finetune_src/r2r/agent_cmt.py:61-63,827-832 = DDP):
  * ranks start from DIFFERENT parameters (seed + rank, r2r/main.py:446); FlatTrainer's constructor broadcast makes them rank 0's;
  * after backward on different episodes, every rank's gradient arena == the mean of the per-episode gradients of one process
    (chunked flush -> all-reduce pipeline, float32 and bf16 payload);
  * two optimizer steps replayed from captured hipGraphs (flush cut into ranges, all-reduce between the graphs) leave every
    rank with the parameters of the single process.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORLD = 2
STEPS = 2


def build(rank_init):
    from vln_imagine_amd import synth
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT
    from vln_imagine_amd.hamt.spec import param_shapes
    cfg = HamtConfig(num_l_layers=1, num_x_layers=2, num_h_pano_layers=1)
    m = NavCMT(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()}
    if rank_init:                      # what a checkpoint does not cover starts different on every rank
        g = torch.Generator().manual_seed(100 + rank_init)
        for k in sd:
            if "contrastive_alignment_model" in k or "imagine_embeddings" in k or "next_action" in k:
                sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=g)
    m.load_state_dict(sd)
    return cfg, m.cuda().eval().set_compute_dtype(torch.bfloat16)


def episode(rank):
    from vln_imagine_amd import synth
    from vln_imagine_amd.hamt.episode import EpisodeTensors
    return EpisodeTensors(synth.HamtEpisode(tag=f"dp{rank}", B=4, L=80, V=37, I=4, T=3, ragged=True), "cuda")


def main():
    outdir, mode = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.episode import run_episode
    from vln_imagine_amd.train import FlatTrainer
    import torch.distributed as dist

    def fwd_bwd_on(model, et):
        loss = run_episode(model, et, criterion=ops.cross_entropy_sum, keep=False)["loss"]
        loss.backward()
        return loss

    if mode == "single":
        _, m = build(0)
        tr = FlatTrainer(m, lr=1e-3)
        ets = [episode(r) for r in range(WORLD)]
        tr.zero_grad()
        for et in ets:
            fwd_bwd_on(m, et)
        tr.flush()
        grads = tr.flat_g.clone() / WORLD
        for _ in range(STEPS):
            tr.zero_grad()
            for et in ets:
                fwd_bwd_on(m, et)
            tr.flush()
            tr.flat_g.mul_(1.0 / WORLD)
            tr.step()
        torch.cuda.synchronize()
        torch.save({"grads": grads.cpu(), "params": tr.flat_p.cpu()}, os.path.join(outdir, "single.pt"))
        return

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, m = build(rank)                                   # rank r starts from its own values of the new heads
    before = torch.cat([p.detach().reshape(-1).cpu() for p in m.parameters()])
    tr = FlatTrainer(m, lr=1e-3, overlap_chunks=3, chunk_mb=8)           # constructor broadcast; several chunks per range
    after = torch.cat([p.detach().reshape(-1).cpu() for p in m.parameters()])
    et = episode(rank)
    res = {"changed_by_broadcast": bool((before != after).any()), "n_ranges": len(tr.comm_ranges())}
    for name, cd in (("grads_f32", None), ("grads_bf16", torch.bfloat16)):
        tr.grad_comm_dtype = cd
        tr.zero_grad()
        fwd_bwd_on(m, et)
        tr.allreduce_grads()
        torch.cuda.synchronize()
        res[name] = tr.flat_g.cpu().clone()
    tr.grad_comm_dtype = None
    res["params0"] = tr.flat_p.cpu().clone()
    step = tr.capture(lambda: fwd_bwd_on(m, et), warmup=0)
    res["n_flush_graphs"] = sum(g is not None for g in step.g_flush)
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    res["params"] = tr.flat_p.cpu().clone()
    torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
