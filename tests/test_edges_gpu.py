"""GPU: edge cases of the hot path - smallest shapes, longest sequences the kernels take, annotation lists with nothing to score."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _r(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 8, 16), (3, 768, 8), (1, 768, 768), (129, 136, 72)])
def test_gemm_smallest_shapes(dtype, M, N, K):
    from vln_imagine_amd import ops
    K = max(K, 16 // (4 if dtype == torch.float32 else 2))
    a, b = _r((M, K), dtype, 1, 0.5), _r((N, K), dtype, 2, 0.1)
    bias = _r((N,), torch.float32, 3, 0.1)
    out = ops.gemm_nt(a, b, bias=bias, act=2)
    ref = torch.relu(a.double() @ b.double().t() + bias.double())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (out.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_attention_single_key_and_masked_tail():
    """One key (softmax = 1) and a row whose only visible key is the first one (additive -10000 on the rest)."""
    from vln_imagine_amd import ops
    for dtype in (torch.float32, torch.bfloat16):
        B, Sq, Sk = 2, 5, 1
        q, k, v = _r((B * Sq, 768), dtype, 4), _r((B * Sk, 768), dtype, 5), _r((B * Sk, 768), dtype, 6)
        out, _ = ops.attn_fwd(q, k, v, B, Sq, Sk, torch.zeros(B, Sk, device="cuda"))
        assert torch.allclose(out.view(B, Sq, 768).float(), v.view(B, 1, 768).float().expand(B, Sq, 768), atol=1e-6)
        Sk = 70
        k, v = _r((B * Sk, 768), dtype, 7), _r((B * Sk, 768), dtype, 8)
        km = torch.full((B, Sk), -10000.0, device="cuda")
        km[:, 0] = 0
        out, _ = ops.attn_fwd(q, k, v, B, Sq, Sk, km)
        ref = v.view(B, Sk, 768)[:, :1].float().expand(B, Sq, 768)
        assert (out.view(B, Sq, 768).float() - ref).abs().max().item() < (1e-5 if dtype == torch.float32 else 1e-2)


def test_aux_head_with_nothing_to_score_returns_python_zero():
    """No flag-'True' slot with a noun phrase: the reference returns the int 0 and the untouched imaginations (vilmodel_cmt.py:786-790)."""
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.models.vilmodel_cmt import AlignWithContrastiveLoss
    head = AlignWithContrastiveLoss(HamtConfig()).cuda()
    B, L, I = 2, 12, 3
    txt, img = _r((B, L, 768), torch.float32, 9), _r((B, I, 768), torch.float32, 10)
    tm, im = torch.ones(B, L, dtype=torch.bool, device="cuda"), torch.ones(B, I, dtype=torch.bool, device="cuda")
    segs = [[[1, 3], [4, 6], [7, 9]]] * B
    flags = [["False", "True", "False"], ["False", "False", "False"]]
    nps = [[[[1, 2]], [], [[7, 8]]], [[], [], []]]            # the one flagged slot has no noun phrase
    loss, new_img = head(align_txt_embeds=txt, txt_masks=tm, align_imagine_embeds=img, imagine_masks=im, sub_instr_segs=segs,
                         sub_instr_imag_flag=flags, noun_phrase_segs=nps)
    assert loss == 0 and not torch.is_tensor(loss) and torch.equal(new_img, img)


def test_longest_history_the_agent_can_build():
    """max_action_steps history tokens + 37 observation tokens still fit the attention kernels; logits stay finite where navigable."""
    from vln_imagine_amd import synth
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode
    from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT
    from vln_imagine_amd.hamt.spec import param_shapes
    cfg = HamtConfig(num_l_layers=1, num_x_layers=1, num_h_pano_layers=1)
    T = 15                                                     # R2R cap (r2r/parser.py:38)
    ep = synth.HamtEpisode(tag="edge", B=2, L=80, V=37, I=6, T=T, ragged=True)
    m = NavCMT(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
    m = m.cuda().eval().set_compute_dtype(torch.bfloat16)
    out = run_episode(m, EpisodeTensors(ep, "cuda"))
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and out["hist_o"][-1].shape[1] == T
    for t in range(T):
        nav = torch.from_numpy(ep.steps[t]["ob_nav_types"]).cuda() != 0
        assert torch.isfinite(out["logits"][t][nav]).all() and torch.isinf(out["logits"][t][~nav]).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("table_rows,rows,H", [(3, 2368, 768), (2, 100, 768), (8, 37, 520), (1, 33, 256)])
def test_small_table_scatter_equals_index_add(dtype, table_rows, rows, H):
    """Embedding gradient of a table with a handful of rows (navigation types): block-level accumulators instead of per-element atomics."""
    from vln_imagine_amd import _lib, ops
    g = torch.Generator().manual_seed(table_rows * 100 + rows)
    src = torch.randint(-3, 4, (rows, H), generator=g).float().to(dtype).cuda()        # integers: exact in any summation order
    idx = torch.randint(0, table_rows, (rows,), generator=g).cuda()
    tab = torch.randint(-2, 3, (table_rows, H), generator=g).float().cuda()
    ref = tab.clone().index_add_(0, idx, src.float())
    _lib.call("vlni_scatter_add_rows_small", ops._dt(src), src.data_ptr(), src.stride(0), idx.data_ptr(), tab.data_ptr(), rows, H, table_rows,
              torch.cuda.current_stream().cuda_stream)
    assert torch.equal(tab, ref)
