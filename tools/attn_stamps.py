"""Per-phase cycle stamps of the resident attention backward kernel (a -DVLNI_RES_STAMPS build, tools/build_variant.sh): every wave's
lane 0 records s_memtime at kernel start, after its staging wait, after the first barrier, after phase 1 and at the end.
usage: VLNI_LIB_PATH=.../stamps.so python tools/attn_stamps.py [drop_p]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vln_imagine_amd import _lib, ops  # noqa: E402

dt, B, H, nh = torch.bfloat16, int(os.environ.get("B", "384")), 768, 12
drop_p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
for Sq, Sk in ((86, 43), (43, 86), (86, 86), (43, 43), (80, 80)):
    q, kv = r(B * Sq, 3 * H), r(B * Sk, 3 * H)
    km = torch.zeros(B, Sk, device="cuda")
    qs, ks, vs = q[:, :H], kv[:, H:2 * H], kv[:, 2 * H:]
    o, l = ops.attn_fwd(qs, ks, vs, B, Sq, Sk, km, drop=(drop_p, 7))
    do, dq, dkv = torch.randn_like(o), torch.empty_like(q), torch.empty_like(kv)
    st = torch.zeros(B * nh * 4 * 8, dtype=torch.int64, device="cuda")
    args = (1, qs.data_ptr(), qs.stride(0), ks.data_ptr(), ks.stride(0), vs.data_ptr(), vs.stride(0), km.data_ptr(), 0, o.data_ptr(), o.stride(0),
            do.data_ptr(), do.stride(0), l.data_ptr(), dq.data_ptr(), dq.stride(0), dkv[:, H:2 * H].data_ptr(), dkv.stride(0), dkv[:, 2 * H:].data_ptr(),
            dkv.stride(0), st.data_ptr(), B, nh, Sq, Sk, 0.125, drop_p, 7, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.call("vlni_attn_bwd", *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.call("vlni_attn_bwd", *args); e1.record(); torch.cuda.synchronize()
    t = st.cpu().numpy().reshape(B * nh, 4, 8).astype(np.float64)
    span = (t[:, :, 4].max() - t[:, :, 0].min())
    med = lambda a: float(np.median(a))
    print(f"Sq {Sq} Sk {Sk} drop {drop_p}: launch {e0.elapsed_time(e1) * 1e3:.1f} us; clock span {span:.0f} ticks (s_memtime runs at 100 MHz: {span / 100:.1f} us)")
    for w in range(4):
        print(f"   wave {w}: dma-issue {med(t[:, w, 5] - t[:, w, 0]):6.0f} ld-issue {med(t[:, w, 6] - t[:, w, 5]):6.0f} delta {med(t[:, w, 7] - t[:, w, 6]):6.0f} wait {med(t[:, w, 1] - t[:, w, 7]):6.0f}  barrier1 {med(t[:, w, 2] - t[:, w, 1]):7.0f}  phase1 {med(t[:, w, 3] - t[:, w, 2]):7.0f}"
              f"  barrier2+phase2 {med(t[:, w, 4] - t[:, w, 3]):7.0f}  | block total {med(t[:, w, 4] - t[:, w, 0]):7.0f} ticks")
    # how many blocks are alive at once (by wave 0's start / end), sampled mid-launch
    starts, ends = t[:, 0, 0], t[:, 0, 4]
    mid = (starts.min() + ends.max()) / 2
    print(f"   blocks alive at mid-launch: {int(((starts <= mid) & (ends >= mid)).sum())} of {B * nh} (256 CUs)")
