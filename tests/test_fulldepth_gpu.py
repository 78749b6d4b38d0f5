"""Full-depth parity of the TIMED path. The goldens pin 2-layer models at B = 4 in fp32; bench.py times 9 L + 4 X + 2 pano
(HAMT) and 9 L + 2 pano + 4 + 4 X (DUET) in bf16 at the batch where the GEMM autotune picks the big-tile / transposing-read /
ring kernels. Here, forward AND backward at the shipped depth:
  * fp32 HIP path vs the CPU oracle (itself held to the reference's goldens at 2e-5) at a batch the oracle finishes in seconds;
  * bf16 HIP path vs the fp32 HIP path at the bench's batch and shapes with AUTOTUNE on, so the variants bench.py ends up
    with (LDS-DMA 8-wave, 192x128, NN dgrad at >= 4096 rows, grouped tile order, ring wgrad) are the ones compared.
Bounds for bf16 are what was measured on MI355X (printed by the test) x 2; SURVEY.md 8(d) expects ~1e-2 relative on logits.
bench.py reports the same three figures for its model in `bf16_vs_fp32`."""
import pytest
import torch

from vln_imagine_amd import synth
from vln_imagine_amd.compare import compare_runs

pytestmark = pytest.mark.gpu


def _hamt(B, T, tag):
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode
    return HamtConfig(), synth.HamtEpisode(tag=tag, B=B, L=80, V=37, I=6, T=T, ragged=True), EpisodeTensors, run_episode, "logits"


def _duet(B, T, tag):
    from vln_imagine_amd.duet.config import DuetConfig
    from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode
    return DuetConfig(), synth.DuetEpisode(tag=tag, B=B, L=80, V=36, I=6, T=T, ragged=True), DuetEpisodeTensors, run_episode, "fused"


def _product(family, cfg, dtype):
    if family == "hamt":
        from tests.test_hamt_gpu import build_product
    else:
        from tests.test_duet_gpu import build_product
    return build_product(cfg, dtype)


def _oracle(family, cfg):
    if family == "hamt":
        from oracle.hamt_oracle import HamtOracle as O
        from vln_imagine_amd.hamt.spec import param_shapes
    else:
        from oracle.duet_oracle import DuetOracle as O
        from vln_imagine_amd.duet.spec import param_shapes
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()}
    return O(cfg, sd), sd


@pytest.mark.parametrize("family,B", [("hamt", 16), ("duet", 8)])
def test_fp32_full_depth_fwd_bwd_matches_cpu_oracle(family, B):
    from vln_imagine_amd import ops
    cfg, ep, ET, run, key = (_hamt if family == "hamt" else _duet)(B, 2, "full32")
    torch.set_num_threads(16)
    oracle, sd = _oracle(family, cfg)
    ref = run(oracle, ET(ep, "cpu"))
    ref["loss"].backward()
    model = _product(family, cfg, torch.float32)
    out = run(model, ET(ep, "cuda"), criterion=ops.cross_entropy_sum)
    out["loss"].backward()
    r = compare_runs(out, ref, dict(model.named_parameters()), sd, key)
    print(f"\n[{family} fp32 HIP vs CPU oracle, full depth, B={B}] {r}")
    assert r["loss_abs"] <= 1e-4 and r["logit_max_abs"] <= 1e-4, r            # BASELINE.json north_star tolerance
    assert r["grad_rel_l2"] <= 2e-5 and r["grad_worst_param_rel_l2"] <= 5e-3, r          # measured 1.0e-6 / 1.5e-6
    assert r["params_compared"] > 100


# REQUIREMENTS, not 1.3 x the last measurement. profiles/r06_bf16_ablation.md: the 16-bit paths' distance from the float32
# path is the storage format's rounding spread over every activation - no single spot (stored GELU', the language Q / K / V reduced once, the taped
# against the stepwise program, the batch) moves the bf16 gradient error by more than 0.004, float16 on the SAME kernels is 2.8 x closer
# (8 against 11 significand bits), and 300 optimizer steps in bf16 / fp16 / fp32 reach the same loss. So the bound is the format's own yardstick:
#   gradient rel-L2 <= 0.112 for bf16 = what plain PyTorch gives with torch.autocast(bfloat16) on the CPU oracle at this depth,
#   a quarter of it (two more... three more significand bits = 8 x finer, sqrt-summed over the path) <= 0.04 for float16;
#   worst single parameter <= 0.25 / 0.10 (gradients that are small by softmax shift invariance sit there);
#   logits <= 2^-4 / 2^-7 (half an ulp of a 16-bit value of magnitude 16 / of a float16 value of magnitude 8), loss <= 1e-3 / 2.5e-4.
# Measured, rounds 2 - 6 (loss, logits, gradient, worst parameter): hamt bf16 2.4e-6 ... 3.5e-4 / 0.030 - 0.045 / 0.086 - 0.094 / 0.15 - 0.18, duet bf16
# 3.3e-5 ... 2.4e-4 / 0.016 - 0.020 / 0.083 - 0.085 / 0.13; float16 (loss scale 2^14) 2.7e-5 ... 8e-5 / 0.0020 - 0.0044 / 0.027 - 0.030 / 0.041 - 0.075.
BF16_BOUNDS = {"hamt": (1e-3, 0.0625, 0.112, 0.25), "duet": (1e-3, 0.0625, 0.112, 0.25)}
F16_BOUNDS = {"hamt": (2.5e-4, 0.0078125, 0.04, 0.10), "duet": (2.5e-4, 0.0078125, 0.04, 0.10)}


@pytest.mark.parametrize("family,B,low", [("hamt", 64, torch.bfloat16), ("duet", 32, torch.bfloat16), ("hamt", 64, torch.float16),
                                          ("duet", 32, torch.float16)])
def test_bf16_timed_path_tracks_fp32_at_bench_shapes(family, B, low):
    """fwd + bwd, the bench's batch and depth, THE PROGRAM bench.py TIMES: the taped episode (step-by-step forward, one episode-batched
    backward) captured by FlatTrainer.capture and REPLAYED, kernels chosen by the autotune during the warm-up steps (direct gradient
    accumulation + deferred grouped weight gradients, the ring / partial-slab kernels) - against the float32 step-by-step run, which the
    reference's goldens pin directly (tests/test_hamt_gpu.py, tests/test_duet_gpu.py). lr = 0: the replayed optimizer step leaves the weights."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    if family == "hamt":
        from vln_imagine_amd.hamt.episode import run_episode_taped as run_taped
    else:
        from vln_imagine_amd.duet.episode import run_episode_taped as run_taped
    assert ops.AUTOTUNE
    cfg, ep, ET, run, key = (_hamt if family == "hamt" else _duet)(B, 2, "full16")
    et = ET(ep, "cuda")
    m32 = _product(family, cfg, torch.float32)
    o32 = run(m32, et, criterion=ops.cross_entropy_sum)
    o32["loss"].backward()
    m16 = _product(family, cfg, low)
    # float16 needs the loss scale (activation gradients of the deep layers are below its 6e-5 normal range): the backward runs on
    # S * loss, exactly as a training step does, and the arena is divided by S before the comparison
    S = 16384.0 if low == torch.float16 else 1.0
    tr = FlatTrainer(m16, lr=0.0, weight_decay=0.0, loss_scale=S)
    try:
        before = set(ops._GEMM_BEST.values())
        tape, stash = ops.EpisodeTape(ep.T), {}

        def fwd_bwd():
            o = run_taped(m16, et, tape=tape, criterion=ops.cross_entropy_sum)
            (o["loss"] * tr.loss_scale).backward()
            stash["out"] = o
            return o["loss"]

        step = tr.capture(fwd_bwd, warmup=2)     # second warm-up step: every launch runs its cached autotune winner, as the capture does
        step()
        torch.cuda.synchronize()
        o16 = stash["out"]
        tr.flat_g.mul_(1.0 / S)
        if family == "duet":                     # maps padded to the episode's largest: the step's own columns (the rest is -inf)
            o16 = dict(o16, fused=[a[:, :b.shape[1]] for a, b in zip(o16["fused"], o32["fused"])])
        r = compare_runs(o16, o32, dict(m16.named_parameters()), dict(m32.named_parameters()), key)
        picked = sorted(set(ops._GEMM_BEST.values()) | before)
        print(f"\n[{family} {str(low)[6:]} taped + graph replay vs fp32 stepwise, full depth, B={B}] {r}\n  GEMM variants the autotune picked: {picked}; wgrad choices: {sorted(set(ops._TN_BEST.values()))}")
        la, lg, gr, gw = (BF16_BOUNDS if low == torch.bfloat16 else F16_BOUNDS)[family]
        assert r["loss_abs"] <= la and r["logit_max_abs"] <= lg and r["grad_rel_l2"] <= gr and r["grad_worst_param_rel_l2"] <= gw, r
        assert 32 in picked or family == "duet", picked        # the 256 x 128 loader-wave kernel ran (the step's 8 k-row launches)
    finally:
        tr.close()
        ops.set_seed_base(None)
        ops._WQ.clear()


@pytest.mark.parametrize("family", ["duet", "hamt"])
def test_fp32_parity_path_runs_200_token_instructions(family):
    """The reference's DUET scripts run --max_instr_len 200 (VLN-DUET/map_nav_src/scripts/run_r2r.sh): 200 + 6 keys are beyond the
    float32 tile attention kernels (128 keys) and go through the generic ones. fp32 HIP vs the CPU oracle, forward + backward, two
    layers of each kind."""
    from tests.golden.variants import DUET_C1, HAMT_C1
    from vln_imagine_amd import ops
    if family == "duet":
        from vln_imagine_amd.duet.config import DuetConfig
        from vln_imagine_amd.duet.episode import DuetEpisodeTensors as ET, run_episode as run
        cfg, ep, key = DuetConfig(**DUET_C1), synth.DuetEpisode(tag="long", B=2, L=200, V=36, I=6, T=2, ragged=True), "fused"
    else:
        from vln_imagine_amd.hamt.config import HamtConfig
        from vln_imagine_amd.hamt.episode import EpisodeTensors as ET, run_episode as run
        cfg, ep, key = HamtConfig(**HAMT_C1), synth.HamtEpisode(tag="long", B=2, L=200, V=37, I=6, T=2, ragged=True), "logits"
    oracle, sd = _oracle(family, cfg)
    ref = run(oracle, ET(ep, "cpu"))
    ref["loss"].backward()
    model = _product(family, cfg, torch.float32)
    out = run(model, ET(ep, "cuda"), criterion=ops.cross_entropy_sum)
    out["loss"].backward()
    r = compare_runs(out, ref, dict(model.named_parameters()), sd, key)
    print(f"\n[{family} fp32 HIP vs CPU oracle, L = 200] {r}")
    assert r["loss_abs"] <= 1e-4 and r["logit_max_abs"] <= 1e-4, r
    assert r["grad_rel_l2"] <= 2e-5 and r["grad_worst_param_rel_l2"] <= 5e-3, r
