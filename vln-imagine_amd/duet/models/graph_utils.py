"""Host-side topological map of the DUET agent (`from models.graph_utils import GraphMap`, r2r/agent.py:23).

Out of the GPU hot path (pure Python bookkeeping); provided so the `models` package is
self-contained. Behaviour follows VLN-DUET/map_nav_src/models/graph_utils.py:7-170: incremental all-pairs shortest
paths relaxed through each newly VISITED viewpoint, running-mean node embeddings, 7-d relative position features."""
import math

import numpy as np

MAX_DIST = 30
MAX_STEP = 10
_INF = 95959595


def calc_position_distance(a, b):
    return math.sqrt(sum((b[i] - a[i]) ** 2 for i in range(3)))


def calculate_vp_rel_pos_fts(a, b, base_heading=0, base_elevation=0):
    """heading / elevation / distance of b seen from a (the simulator's x-y axes are transposed)."""
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    # `**` on python floats is libm pow (1 ulp off x * x now and then): kept so that the features equal the reference's bit for bit
    xy = max(math.sqrt(dx ** 2 + dy ** 2), 1e-8)
    xyz = max(math.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
    heading = math.asin(dx / xy)
    if b[1] < a[1]:
        heading = math.pi - heading
    return heading - base_heading, math.asin(dz / xyz) - base_elevation, xyz


def get_angle_fts(headings, elevations, angle_feat_size):
    f = np.stack([np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)], 1).astype(np.float32)
    return np.tile(f, (1, max(1, angle_feat_size // 4)))


class FloydGraph(object):
    """All-pairs shortest paths that only route through visited nodes; `update(k)` relaxes every pair through k."""

    def __init__(self):
        self._dis = {}          # node -> {node: distance}
        self._via = {}          # node -> {node: intermediate node or ""}
        self._visited = set()

    def _row(self, x):
        if x not in self._dis:
            self._dis[x], self._via[x] = {}, {}
        return self._dis[x]

    def distance(self, x, y):
        return 0 if x == y else self._dis.get(x, {}).get(y, _INF)

    def add_edge(self, x, y, dis):
        if dis < self.distance(x, y) or (x != y and y not in self._row(x)):
            if dis < self._row(x).get(y, _INF):
                self._row(x)[y] = self._row(y)[x] = dis
                self._via[x][y] = self._via[y][x] = ""

    def update(self, k):
        nodes = list(self._dis)
        dk = self._row(k)
        for x in nodes:
            if x == k:
                continue
            dxk = self.distance(x, k)
            if dxk >= _INF:
                continue
            for y in nodes:
                if y == x or y == k:
                    continue
                cand = dxk + dk.get(y, _INF)
                if cand < self._dis[x].get(y, _INF):
                    self._dis[x][y] = self._dis[y][x] = cand
                    self._via[x][y] = self._via[y][x] = k
        self._visited.add(k)

    def visited(self, k):
        return k in self._visited

    def path(self, x, y):
        """nodes after x up to and including y along the stored shortest path."""
        if x == y:
            return []
        k = self._via.get(x, {}).get(y, "")
        return [y] if k == "" else self.path(x, k) + self.path(k, y)


class GraphMap(object):
    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions = {}
        self.graph = FloydGraph()
        self.node_embeds = {}            # viewpoint -> [sum of embeddings, count]
        self.node_stop_scores = {}
        self.node_nav_scores = {}
        self.node_step_ids = {}

    def update_graph(self, ob):
        self.node_positions[ob["viewpoint"]] = ob["position"]
        for cc in ob["candidate"]:
            self.node_positions[cc["viewpointId"]] = cc["position"]
            self.graph.add_edge(ob["viewpoint"], cc["viewpointId"], calc_position_distance(ob["position"], cc["position"]))
        self.graph.update(ob["viewpoint"])

    def update_node_embed(self, vp, embed, rewrite=False):
        if rewrite or vp not in self.node_embeds:
            self.node_embeds[vp] = [embed, 1]
        else:
            self.node_embeds[vp][0] = self.node_embeds[vp][0] + embed      # out of place: keeps autograd history intact
            self.node_embeds[vp][1] += 1

    def get_node_embed(self, vp):
        s, n = self.node_embeds[vp]
        return s / n

    def get_pos_fts(self, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4):
        """[len(gmap_vpids), angle_feat_size + 3]: sin/cos heading, sin/cos elevation, line / shortest distance, path steps."""
        ang = np.zeros((len(gmap_vpids), 2), np.float32)
        dist = np.zeros((len(gmap_vpids), 3), np.float32)
        for i, vp in enumerate(gmap_vpids):
            if vp is None:
                continue
            h, e, d = calculate_vp_rel_pos_fts(self.node_positions[cur_vp], self.node_positions[vp], cur_heading, cur_elevation)
            ang[i] = (h, e)
            dist[i] = (d / MAX_DIST, self.graph.distance(cur_vp, vp) / MAX_DIST, len(self.graph.path(cur_vp, vp)) / MAX_STEP)
        return np.concatenate([get_angle_fts(ang[:, 0], ang[:, 1], angle_feat_size), dist], 1)

    def save_to_json(self):
        nodes = {}
        for vp, pos in self.node_positions.items():
            n = {"location": pos, "visited": self.graph.visited(vp)}
            if n["visited"]:
                n["stop_prob"] = self.node_stop_scores[vp]["stop"]
                n["og_objid"] = self.node_stop_scores[vp]["og"]
            else:
                n["nav_prob"] = self.node_nav_scores[vp]
            nodes[vp] = n
        edges = [(k, kk) for k, v in self.graph._dis.items() for kk in v]
        return {"nodes": nodes, "edges": edges}
