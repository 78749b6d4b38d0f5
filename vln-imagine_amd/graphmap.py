"""DUET's topological maps of one rank's B episodes, resident on the device.

Replaces, behind the same dictionary keys, what the reference agent does on the host every step:
  GraphMap / FloydGraph        VLN-DUET/map_nav_src/models/graph_utils.py:43-161   (python dict-of-dicts per episode)
  _nav_gmap_variable           VLN-DUET/map_nav_src/r2r/agent.py:98-168            (python double loop over node pairs, numpy, H2D copies)
  _nav_vp_variable             VLN-DUET/map_nav_src/r2r/agent.py:170-207
  node image bookkeeping       r2r/agent.py:464-479 (update_node_embed / get_node_embed, graph_utils.py:115-128)

Split of labour: viewpoint NAMES are strings, so the host keeps name -> slot tables and decides the node ORDER of each step's
tensors (a few hundred integers); distances, intermediate-node marks, positions, position features, hop counts and pair distances
live in HBM and are produced by the kernels of csrc/graphmap.hip (float64 where the reference computes in python floats).
Node images stay autograd tensors: a running sum per slot, updated out of place so that gradients reach every panorama
encoding that contributed, exactly as `pad_tensors_wgrad` over `get_node_embed` does in the reference.
No CPU fallback: without libvlni.so the constructor raises.
"""
import numpy as np
import torch

from . import _lib, ops


def _dev(a, dtype, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype, non_blocking=True)


class DeviceGraphMap:
    def __init__(self, obs, cap=128, device="cuda", enc_full_graph=True, act_visited_nodes=False, hidden=768):
        """`obs`: the first observations (one per episode); like the reference (r2r/agent.py:397-399) the start viewpoints are
        recorded and observed at once. `cap`: node slots per episode (<= 256)."""
        _lib.load()
        assert 2 <= cap <= 256
        self.B, self.G, self.dev, self.H = len(obs), cap, torch.device(device), hidden
        self.enc_full_graph, self.act_visited_nodes = enc_full_graph, act_visited_nodes
        B, G = self.B, cap
        self.pos = torch.zeros((B, G, 3), dtype=torch.float64, device=self.dev)
        self.dis = torch.empty((B, G, G), dtype=torch.float64, device=self.dev)
        self.via = torch.empty((B, G, G), dtype=torch.int32, device=self.dev)
        self.seen = torch.empty((B, G), dtype=torch.uint8, device=self.dev)
        self.status = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        _lib.call("vlni_graph_init", self.dis.data_ptr(), self.via.data_ptr(), self.seen.data_ptr(), B, G, ops._st())
        self.start_vp = [ob["viewpoint"] for ob in obs]
        self.names = [[] for _ in range(B)]              # insertion order = the reference's node_positions dict order
        self.slot = [dict() for _ in range(B)]
        self.visited = [set() for _ in range(B)]         # host mirror of `seen` (only used to order nodes)
        self.step_id = [dict() for _ in range(B)]
        self.emb_sum = None                              # [B*G + 1, H] autograd running sums (last row: zeros for [stop] / padding)
        self.emb_cnt = np.zeros((B * G + 1,), np.float32)
        self._via_host = None
        self.observe(obs)

    # ---- graph updates -------------------------------------------------------------------------------------------------
    def _slot(self, b, name):
        s = self.slot[b].get(name)
        if s is None:
            s = len(self.names[b])
            if s >= self.G:
                raise ValueError(f"episode {b}: more than {self.G} map nodes (raise cap)")
            self.slot[b][name] = s
            self.names[b].append(name)
        return s

    def observe(self, obs, ended=None):
        """GraphMap.update_graph for every episode that has not ended (r2r/agent.py:397-399,604-608)."""
        B = self.B
        C = max(1, max(len(ob["candidate"]) for ob in obs))
        cur = np.full((B,), -1, np.int32)
        cand = np.full((B, C), -1, np.int32)
        cur_pos = np.zeros((B, 3), np.float64)
        cand_pos = np.zeros((B, C, 3), np.float64)
        cand_dist = np.zeros((B, C), np.float64)
        for b, ob in enumerate(obs):
            if ended is not None and ended[b]:
                continue
            cur[b] = self._slot(b, ob["viewpoint"])
            cur_pos[b] = ob["position"]
            for j, c in enumerate(ob["candidate"]):
                cand[b, j] = self._slot(b, c["viewpointId"])
                cand_pos[b, j] = c["position"]
                # the reference's own expression (graph_utils.py:7-13): python `**` is libm pow, one ulp off x*x now and then,
                # and the map compares these lengths with `<` - a handful of flops per step, kept on the host for bit equality
                dx, dy, dz = (c["position"][a] - ob["position"][a] for a in range(3))
                cand_dist[b, j] = np.sqrt(dx ** 2 + dy ** 2 + dz ** 2)
            self.visited[b].add(ob["viewpoint"])
        n_nodes = np.array([len(n) for n in self.names], np.int32)
        t = [_dev(cur, torch.int32, self.dev), _dev(cand, torch.int32, self.dev), _dev(cur_pos, torch.float64, self.dev),
             _dev(cand_pos, torch.float64, self.dev), _dev(cand_dist, torch.float64, self.dev), _dev(n_nodes, torch.int32, self.dev)]
        _lib.call("vlni_graph_observe", self.pos.data_ptr(), self.dis.data_ptr(), self.via.data_ptr(), self.seen.data_ptr(),
                  t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), B, self.G, C, ops._st())
        self._via_host = None

    def mark_step(self, obs, t, ended=None):
        """node_step_ids[current viewpoint] = t + 1 (r2r/agent.py:455-457)."""
        for b, ob in enumerate(obs):
            if ended is None or not ended[b]:
                self.step_id[b][ob["viewpoint"]] = t + 1

    def path(self, b, x, y):
        """FloydGraph.path (graph_utils.py:75-93) for the simulator side (make_equiv_action): one D2H copy of the marks per step."""
        if self._via_host is None:
            self._via_host = self.via.cpu().numpy()
        via, names = self._via_host[b], self.names[b]

        def walk(i, j):
            if i == j:
                return []
            k = int(via[i, j])
            return [names[j]] if k < 0 else walk(i, k) + walk(k, j)
        return walk(self.slot[b][x], self.slot[b][y])

    # ---- node images (autograd) ------------------------------------------------------------------------------------------
    def update_node_embeds(self, obs, pano_embeds, pano_masks, cand_vpids, ended=None):
        """r2r/agent.py:461-479: the current viewpoint's image becomes the masked mean of its panorama (rewrite), every still
        unvisited candidate accumulates the embedding of the view it is seen through."""
        B, G, V = self.B, self.G, pano_embeds.shape[1]
        m = pano_masks.to(pano_embeds.dtype)
        avg = (pano_embeds * m.unsqueeze(2)).sum(1) / m.sum(1, keepdim=True)
        if self.emb_sum is None:
            self.emb_sum = torch.zeros((B * G + 1, self.H), dtype=pano_embeds.dtype, device=self.dev)
        dst, src, rewrite = [], [], []
        for b, ob in enumerate(obs):
            if ended is not None and ended[b]:
                continue
            s = b * G + self.slot[b][ob["viewpoint"]]
            rewrite.append(s)
            dst.append(s)
            src.append(b)                                                  # row b of `avg`
            self.emb_cnt[s] = 1
            for j, name in enumerate(cand_vpids[b]):
                if name not in self.visited[b]:
                    s = b * G + self.slot[b][name]
                    dst.append(s)
                    src.append(B + b * V + j)                              # row (b, j) of the panorama
                    self.emb_cnt[s] += 1
        if not dst:
            return
        rows = torch.cat([avg, pano_embeds.reshape(B * V, self.H)], 0).index_select(0, _dev(np.array(src), torch.long, self.dev))
        keep = torch.ones((B * G + 1, 1), dtype=pano_embeds.dtype, device=self.dev)
        keep[_dev(np.array(rewrite), torch.long, self.dev)] = 0
        self.emb_sum = (self.emb_sum * keep).index_add(0, _dev(np.array(dst), torch.long, self.dev), rows)

    def _node_embeds(self, flat_nodes, B, N):
        cnt = _dev(np.maximum(self.emb_cnt, 1), self.emb_sum.dtype, self.dev).unsqueeze(1)
        return (self.emb_sum / cnt).index_select(0, flat_nodes).view(B, N, self.H)

    # ---- per-step tensors -------------------------------------------------------------------------------------------------
    def _pos_fts(self, cur, nodes, heading, elevation, out, col0, width):
        B, N = nodes.shape
        _lib.call("vlni_graph_pos_fts", self.pos.data_ptr(), self.dis.data_ptr(), self.via.data_ptr(), cur.data_ptr(), nodes.data_ptr(),
                  heading.data_ptr(), elevation.data_ptr(), out.data_ptr() + 4 * col0, width, N * width, self.status.data_ptr(),
                  B, self.G, N, 4, ops._st())

    def _pose(self, obs):
        cur = _dev(np.array([self.slot[b][ob["viewpoint"]] for b, ob in enumerate(obs)], np.int32), torch.int32, self.dev)
        heading = _dev(np.array([ob["heading"] for ob in obs], np.float64), torch.float64, self.dev)
        elevation = _dev(np.array([ob["elevation"] for ob in obs], np.float64), torch.float64, self.dev)
        return cur, heading, elevation

    def check(self):
        """Raises if a hop-count walk did not terminate (inconsistent marks); one small D2H copy, call it outside the hot loop."""
        s = int(self.status.item())
        if s:
            raise RuntimeError(f"graph map of episode {s - 1}: hop-count walk did not terminate")

    def nav_gmap_variable(self, obs):
        """The `navigation` batch entries of _nav_gmap_variable (same keys; `gmap_img_embeds` present once node images exist)."""
        B, G = self.B, self.G
        vpids, flags, no_left = [], [], []
        for b, ob in enumerate(obs):
            if self.act_visited_nodes:
                done = [k for k in self.names[b] if k == ob["viewpoint"]]
            else:
                done = [k for k in self.names[b] if k in self.visited[b]]
            dset = set(done)
            todo = [k for k in self.names[b] if k not in dset]
            no_left.append(len(todo) == 0)
            if self.enc_full_graph:
                vpids.append([None] + done + todo)
                flags.append([0] + [1] * len(done) + [0] * len(todo))
            else:
                vpids.append([None] + todo)
                flags.append([0] * (1 + len(todo)))
        N = max(len(v) for v in vpids)
        nodes = np.full((B, N), -2, np.int32)
        step_ids = np.zeros((B, N), np.int64)
        vis = np.zeros((B, N), bool)
        lens = np.array([len(v) for v in vpids])
        for b, ids in enumerate(vpids):
            nodes[b, 0] = -1
            nodes[b, 1:len(ids)] = [self.slot[b][k] for k in ids[1:]]
            step_ids[b, :len(ids)] = [self.step_id[b].get(k, 0) for k in ids]
            vis[b, :len(ids)] = flags[b]
        nodes_t = _dev(nodes, torch.int32, self.dev)
        cur, heading, elevation = self._pose(obs)
        pos_fts = torch.empty((B, N, 7), dtype=torch.float32, device=self.dev)
        self._pos_fts(cur, nodes_t, heading, elevation, pos_fts, 0, 7)
        pair = torch.empty((B, N, N), dtype=torch.float32, device=self.dev)
        _lib.call("vlni_graph_pair_dists", self.dis.data_ptr(), nodes_t.data_ptr(), pair.data_ptr(), B, G, N, ops._st())
        out = {"gmap_vpids": vpids, "gmap_step_ids": _dev(step_ids, torch.long, self.dev), "gmap_pos_fts": pos_fts,
               "gmap_visited_masks": _dev(vis, torch.bool, self.dev), "gmap_pair_dists": pair,
               "gmap_masks": _dev(np.arange(N)[None, :] < lens[:, None], torch.bool, self.dev), "no_vp_left": no_left}
        if self.emb_sum is not None:
            flat = np.where(nodes >= 0, nodes + (np.arange(B) * G)[:, None], B * G)
            out["gmap_img_embeds"] = self._node_embeds(_dev(flat.reshape(-1), torch.long, self.dev), B, N)
        return out

    def nav_vp_variable(self, obs, pano_embeds, cand_vpids, view_lens, nav_types):
        """_nav_vp_variable: [stop] + views; vp_pos_fts = [start-node features on every row | candidate features on rows 1..n]."""
        B, V = pano_embeds.shape[0], pano_embeds.shape[1]
        cur, heading, elevation = self._pose(obs)
        start = np.array([[self.slot[b][self.start_vp[b]]] * (V + 1) for b in range(B)], np.int32)
        cands = np.full((B, V + 1), -2, np.int32)
        for b, names in enumerate(cand_vpids):
            cands[b, 1:1 + len(names)] = [self.slot[b][k] for k in names]
        pos = torch.empty((B, V + 1, 14), dtype=torch.float32, device=self.dev)
        self._pos_fts(cur, _dev(start, torch.int32, self.dev), heading, elevation, pos, 0, 14)
        self._pos_fts(cur, _dev(cands, torch.int32, self.dev), heading, elevation, pos, 7, 14)
        view_lens = torch.as_tensor(view_lens, device=self.dev)
        return {"vp_img_embeds": torch.cat([torch.zeros_like(pano_embeds[:, :1]), pano_embeds], 1), "vp_pos_fts": pos,
                "vp_masks": torch.arange(V + 1, device=self.dev)[None, :] < (view_lens + 1)[:, None],
                "vp_nav_masks": torch.cat([torch.ones((B, 1), dtype=torch.bool, device=self.dev), nav_types == 1], 1),
                "vp_cand_vpids": [[None] + list(c) for c in cand_vpids]}
