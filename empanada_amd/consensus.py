"""Cross-plane consensus, reference names and semantics (``empanada/consensus.py``):
``merge_objects_from_trackers`` :348-469, ``merge_semantic_from_trackers`` :289-346,
``object_iou_graph`` :233-287, ``bounding_box_screening`` :197-231, ``create_graph_of_clusters`` :35-74,
``merge_clusters`` :86-142, ``merge_overlapping`` :166-195, ``merge_instances`` :144-164.

How the work is split (MI355X-first)
  * Every O(#voxel-runs) step runs on the device over ONE run store (runs of all objects sorted by
    (object, start); for the whole-stack path it is the concatenation of the three planes' ``PlaneTracks`` arrays and
    never leaves HBM): box screening (emp_box_pairs), all pairwise run-length intersections in one launch
    (emp_rle_pair_intersections), the voxel vote of every cluster in one launch (emp_track_expand +
    emp_vote_ranges), the unions of merged instances, the fill.
  * The graph logic works on O(#objects) numpy tables.  In the reference every connected component of the IoU graph
    goes through two graph copies and a cluster graph; here a component whose edges all pass the IoU cut is one
    cluster by construction (the cluster graph would have a single node) and is handled by array operations:
    scipy's ``connected_components`` labels, components ordered by their first node (= networkx's enumeration
    order, which fixes the final instance ids).  Only components with an edge at or below the cut go through the
    literal cluster-graph procedure on networkx -- the third-party library the reference calls (consensus.py:2),
    whose set / adjacency iteration order is part of the result.
  * Several ranks: every rank holds the instance tables of all objects and the runs inside its z-slab of the output
    volume; intersections and voted areas are summed over ranks (``reduce``), the graph logic is replicated, and
    every rank votes and paints its own slab.
"""
from itertools import combinations

import networkx as nx
import numpy as np
import torch
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components as _cc_labels

from . import _hip
from .array_utils import merge_boxes, vote_groups

__all__ = ['merge_objects_from_trackers', 'merge_semantic_from_trackers', 'merge_objects_from_tiles',
           'merge_semantic_from_tiles', 'merge_objects3d', 'object_iou_graph',
           'bounding_box_screening', 'create_graph_of_clusters', 'merge_clusters', 'RunStore', 'consensus_objects',
           'consensus_semantic',
           'ConsensusResult']

MIN_OVERLAP = 100
MIN_IOU = 1e-2
# which branches of consensus_objects ran (tests assert that their random scenes reach all of them)
BRANCH_COUNTS = {'fast_components': 0, 'general_components': 0, 'extra_memberships': 0, 'overlap_joins': 0}


class RunStore:
    """Device-resident run lists of a set of objects, sorted by (object, start): ``st``, ``ln`` int64 arrays and
    CSR offsets ``off`` (device int64, n_obj + 1)."""

    def __init__(self, st, ln, off):
        self.st, self.ln, self.off = st, ln, off
        self.n_obj = int(off.numel()) - 1

    @classmethod
    def from_lists(cls, starts_list, runs_list):
        """per-object numpy (starts, runs); each object's runs are put in stable start order (emp_sort_u64_i32)"""
        _hip.require_gpu()
        sizes = np.array([len(s) for s in starts_list], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(off[-1])
        if n:
            st = np.concatenate([np.asarray(s, dtype=np.int64) for s in starts_list])
            ln = np.concatenate([np.asarray(r, dtype=np.int64) for r in runs_list])
            inst = np.repeat(np.arange(len(sizes), dtype=np.int64), sizes)
            if st.max() >= 2 ** 40 or len(sizes) >= 2 ** 23:
                raise ValueError("volume or instance count too large for the 40/23-bit sort key")
            keys = torch.from_numpy(((inst << 40) | st)).cuda().view(torch.uint64)
            vals = torch.arange(n, dtype=torch.int32, device='cuda')
            _, order = _hip.sort_u64_i32(keys, vals, 0, 63)
            order = order.long()
            st_d = torch.from_numpy(st).cuda()[order].contiguous()
            ln_d = torch.from_numpy(ln).cuda()[order].contiguous()
        else:
            st_d = torch.zeros(0, dtype=torch.int64, device='cuda')
            ln_d = torch.zeros(0, dtype=torch.int64, device='cuda')
        return cls(st_d, ln_d, torch.from_numpy(off).cuda())

    @property
    def n_runs(self):
        return int(self.st.numel())

    def intersections(self, pairs):
        """(k, 2) object pairs -> int64 (k,) intersections of their run lists (local runs only)"""
        pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        if len(pairs) == 0:
            return np.zeros(0, dtype=np.int64)
        out = _hip.rle_pair_intersections(self.st, self.ln, self.off, torch.from_numpy(pairs).cuda())
        return out.cpu().numpy()

    def run_counts(self):
        off = self.off.cpu().numpy()
        return off[1:] - off[:-1]

    def expand(self, obj_val):
        """int32 value per object -> int32 value per run (emp_track_expand)"""
        out = torch.empty((max(self.n_runs, 1),), dtype=torch.int32, device=self.st.device)
        val = torch.from_numpy(np.ascontiguousarray(obj_val, dtype=np.int32)).to(self.st.device)
        _hip.call('emp_track_expand', _hip._ptr(self.off), _hip._ptr(val), self.n_obj, self.n_runs, _hip._ptr(out),
                  _hip.stream())
        return out[:self.n_runs]


class ConsensusResult:
    """Consensus instances 1..n in the reference's order: boxes, GLOBAL voxel counts, and the rank's part of their
    voted ranges on the device (``ranges`` (m, 2) int64 sorted by (instance, start); ``off`` CSR offsets, host)."""

    def __init__(self, boxes, areas, ranges, off):
        self.boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 6)
        self.areas = np.asarray(areas, dtype=np.int64)
        self.ranges = ranges
        self.off = np.asarray(off, dtype=np.int64)
        self.alive = np.ones(len(self.areas), dtype=bool)

    @property
    def n(self):
        return len(self.areas)

    def remove_small_objects(self, min_size=64):
        self.alive &= ~(self.areas < min_size)

    def remove_pancakes(self, min_span=4):
        b = self.boxes
        self.alive &= ~((b[:, 3:] - b[:, :3]) < min_span).any(axis=1)

    def instances(self):
        """{instance id: {'box', 'starts', 'runs'}} of the surviving instances (local ranges; one rank: all)"""
        rng = self.ranges.cpu().numpy() if self.ranges is not None and self.ranges.numel() else np.zeros((0, 2), np.int64)
        out = {}
        for i in range(self.n):
            if self.alive[i]:
                r = rng[self.off[i]:self.off[i + 1]]
                out[i + 1] = {'box': tuple(int(x) for x in self.boxes[i]), 'starts': r[:, 0], 'runs': r[:, 1] - r[:, 0]}
        return out

    def paint(self, vol_flat, lo=0, ids=None):
        """write ids[i] (default i + 1; uint32 device volume, or uint8 with one value) over the voxels of every
        surviving instance; vol_flat covers the flat voxel interval starting at lo.  Later instances overwrite
        earlier ones (fill_volume, patterns.py:204-220)."""
        m = int(self.off[-1]) if len(self.off) else 0
        if m == 0 or not self.alive.any():
            return vol_flat
        dev = vol_flat.device
        st = (self.ranges[:m, 0] - int(lo)).contiguous()
        ln = (self.ranges[:m, 1] - self.ranges[:m, 0]).contiguous()
        if vol_flat.dtype == torch.uint8:
            keep = torch.from_numpy(np.repeat(self.alive, np.diff(self.off))).to(dev)
            _hip.fill_runs_u8(vol_flat, st[keep].contiguous(), ln[keep].contiguous(), 1 if ids is None else int(ids))
            return vol_flat
        val = np.arange(1, self.n + 1, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
        val = np.where(self.alive, val, 0)
        order = torch.empty((m,), dtype=torch.int32, device=dev)
        off_d = torch.from_numpy(self.off).to(dev)
        iota = torch.arange(self.n, dtype=torch.int32, device=dev)
        _hip.call('emp_track_expand', _hip._ptr(off_d), _hip._ptr(iota), self.n, m, _hip._ptr(order), _hip.stream())
        _hip.fill_runs_u32(vol_flat, st, ln, order, _hip.np_to_dev_u32(val))
        return vol_flat


def bounding_box_screening(boxes, source_indices):
    """consensus.py:197-231 -> (k,2) unique pairs i<j from different sources whose boxes intersect."""
    boxes = np.asarray(boxes)
    if len(boxes) == 0:
        return np.zeros((0, 2), dtype=np.int64)
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.int32)).cuda()
    src = torch.from_numpy(np.ascontiguousarray(source_indices, dtype=np.int32)).cuda()
    pairs = _hip.box_pairs(b, src_a=src, upper_only=True).cpu().numpy().astype(np.int64)
    if len(pairs) == 0:
        return pairs.reshape(0, 2)
    return np.unique(pairs, axis=0)          # row-sorted like np.unique(..., axis=0) in the reference


def object_iou_graph(source_indices, object_labels, object_boxes, object_starts, object_runs, store=None):
    """consensus.py:233-287"""
    box_matches = bounding_box_screening(object_boxes, source_indices)
    graph = nx.Graph()
    for node_id in range(len(object_labels)):
        graph.add_node(node_id, box=object_boxes[node_id], starts=object_starts[node_id],
                       runs=object_runs[node_id])
    if len(box_matches):
        store = store or RunStore.from_lists(object_starts, object_runs)
        areas = np.array([int(np.sum(r)) for r in object_runs], dtype=np.int64)
        inters = store.intersections(box_matches)
        ious = inters / (areas[box_matches[:, 0]] + areas[box_matches[:, 1]] - inters)   # int64 / int64 -> fp64
        for (r1, r2), pair_iou, inter_area in zip(box_matches, ious, inters):
            if pair_iou > 0:
                graph.add_edge(int(r1), int(r2), iou=pair_iou, overlap=inter_area)
    return graph


# ----------------------------------------------------------------------------- cluster graph (general path)
def _mean_edge(G, members_a, members_b, key):
    """mean of G's edge attribute `key` over all pairs (a, b), absent edges counting 0 (consensus.py:10-33)"""
    total, count = 0, 0
    for a in members_a:
        row = G[a]
        for b in members_b:
            total += row[b][key] if b in row else 0
            count += 1
    return total / count


def create_graph_of_clusters(G, cluster_iou_thr):
    """consensus.py:35-74: nodes of the result = connected components of G after cutting edges with
    iou <= cluster_iou_thr (attribute 'cluster' = set of G's nodes); two clusters are linked when their mean pairwise
    iou exceeds MIN_IOU or their mean pairwise overlap exceeds MIN_OVERLAP."""
    strong = G.copy()
    strong.remove_edges_from([(u, v) for u, v, iou in G.edges(data='iou') if iou <= cluster_iou_thr])
    out = nx.Graph()
    for index, members in enumerate(nx.connected_components(strong)):
        out.add_node(index, cluster=members)
    for a, b in combinations(out.nodes, 2):
        ca, cb = out.nodes[a]['cluster'], out.nodes[b]['cluster']
        mean_iou = _mean_edge(G, ca, cb, 'iou')
        mean_overlap = _mean_edge(G, ca, cb, 'overlap')
        if mean_iou > MIN_IOU or mean_overlap > MIN_OVERLAP:
            out.add_edge(a, b, iou=mean_iou, overlap=mean_overlap)
    return out


def merge_clusters(G):
    """consensus.py:86-142.  Repeatedly take the cluster-graph node with the most neighbours (first in node order on
    ties).  If its largest neighbour holds more objects than it does, the hub is dissolved into every neighbour;
    otherwise every neighbour is absorbed by the hub, and for each of the absorbed neighbour's own neighbours that the
    hub does not yet touch, the edge (hub, neighbour) is re-added carrying that second neighbour's iou -- the
    reference's line :138 links the hub to the node it is about to delete, so second neighbours are dropped; results
    must match, so this does too."""
    H = G.copy()
    while H.number_of_edges() > 0:
        hub = max(H.nodes, key=lambda node: len(H[node]))        # max keeps the first of equals, like the stable sort
        around = sorted(H[hub], key=lambda node: len(H.nodes[node]['cluster']), reverse=True)
        if len(H.nodes[around[0]]['cluster']) > len(H.nodes[hub]['cluster']):
            for other in around:
                H.nodes[other]['cluster'] = H.nodes[other]['cluster'] | H.nodes[hub]['cluster']
                H.remove_edge(hub, other)
            H.remove_node(hub)
            continue
        for other in around:
            H.nodes[hub]['cluster'] = H.nodes[hub]['cluster'] | H.nodes[other]['cluster']
            H.remove_edge(other, hub)
            for second in list(H[other]):
                if not H.has_edge(hub, second):
                    H.add_edge(hub, other, iou=H[other][second]['iou'])
            H.remove_node(other)
    return H


def _general_clusters(n_nodes, edges, edge_iou, edge_overlap, comp_first_nodes, cluster_iou_thr):
    """Clusters of the components that have an edge at or below the IoU cut, through the literal procedure.
    The graph holds every node (the subgraph view's iteration order depends on len(graph)) but only the edges of these
    components, in the order the reference inserts them (sorted pairs): BFS sets, adjacency and node order inside the
    components are then exactly the reference's."""
    graph = nx.Graph()
    graph.add_nodes_from(range(n_nodes))
    for (a, b), iou, ov in zip(edges.tolist(), edge_iou.tolist(), edge_overlap.tolist()):
        graph.add_edge(a, b, iou=iou, overlap=ov)
    out = {}
    for first in comp_first_nodes:
        comp = nx.node_connected_component(graph, first)       # same BFS (and set layout) as nx.connected_components
        cg = merge_clusters(create_graph_of_clusters(graph.subgraph(comp), cluster_iou_thr))
        out[first] = [list(cg.nodes[node]['cluster']) for node in cg.nodes]
    return out


# ----------------------------------------------------------------------------- the consensus on tables
def consensus_objects(src, boxes, areas, store, n_votes, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False,
                      reduce=None, store_index=None):
    """merge_objects_from_trackers (consensus.py:348-469) on tables.

    src (n,) tracker index, boxes (n, 6), areas (n,) GLOBAL voxel counts of the n objects in enumeration order
    (tracker by tracker, dict order inside); store: RunStore of the same objects holding the rank's runs.
    reduce: sums an int64 numpy vector over the ranks (None = single rank).  store_index[i] = position of object i
    in the store (default i): the store may hold more objects (other classes, instances a filter removed).
    Returns a ConsensusResult (instances 1..N in the reference's order)."""
    reduce = reduce or (lambda a: a)
    n = len(src)
    sidx = np.arange(n, dtype=np.int64) if store_index is None else np.asarray(store_index, dtype=np.int64)
    src = np.asarray(src, dtype=np.int64)
    boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 6)
    areas = np.asarray(areas, dtype=np.int64)
    min_cluster_size = 1 if bypass else (n_votes // 2) + 1
    if pixel_vote_thr < min_cluster_size:
        cluster_iou_thr = 0
    empty = ConsensusResult(np.zeros((0, 6)), np.zeros(0), None, np.zeros(1))
    if n == 0:
        return empty

    # ---- IoU graph: screened pairs -> intersections (one launch) -> edges with iou > 0
    pairs = bounding_box_screening(boxes, src)
    inter = reduce(store.intersections(sidx[pairs])) if len(pairs) else np.zeros(0, np.int64)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = inter / (areas[pairs[:, 0]] + areas[pairs[:, 1]] - inter) if len(pairs) else np.zeros(0)
    keep = iou > 0
    edges, e_iou, e_ov = pairs[keep], iou[keep], inter[keep]

    # ---- connected components, in order of their first node (networkx's enumeration order)
    if len(edges):
        adj = coo_matrix((np.ones(len(edges), np.int8), (edges[:, 0], edges[:, 1])), shape=(n, n))
        n_comp, comp_of = _cc_labels(adj, directed=False)
    else:
        n_comp, comp_of = n, np.arange(n)
    first = np.full(n_comp, n, dtype=np.int64)
    np.minimum.at(first, comp_of, np.arange(n))
    comp_rank = np.empty(n_comp, dtype=np.int64)
    comp_rank[np.argsort(first, kind='stable')] = np.arange(n_comp)
    rank_of = comp_rank[comp_of]                       # component rank per node
    size = np.bincount(rank_of, minlength=n_comp)
    weakest = np.full(n_comp, np.inf)
    if len(edges):
        np.minimum.at(weakest, rank_of[edges[:, 0]], e_iou)
    eligible = size >= min_cluster_size
    general = eligible & ~(weakest > cluster_iou_thr)  # some edge does not survive the cut: literal cluster procedure
    clusters_of = {}
    if general.any():
        in_general = general[rank_of[edges[:, 0]]]
        first_by_rank = np.sort(first)
        clusters_of = _general_clusters(n, edges[in_general], e_iou[in_general], e_ov[in_general],
                                        [int(first_by_rank[r]) for r in np.flatnonzero(general)], cluster_iou_thr)
        clusters_of = {int(comp_rank[comp_of[f]]): [c for c in cl if len(c) >= min_cluster_size]
                       for f, cl in clusters_of.items()}

    BRANCH_COUNTS['general_components'] += int(general.sum())
    BRANCH_COUNTS['fast_components'] += int((eligible & ~general).sum())
    # ---- groups to vote on: one per cluster, component by component
    slots = np.where(eligible, 1, 0).astype(np.int64)
    for r, cl in clusters_of.items():
        slots[r] = len(cl)
    base = np.concatenate([[0], np.cumsum(slots)])
    n_groups = int(base[-1])
    if n_groups == 0:
        return empty
    # membership rows (node, group): clusters of the literal procedure may SHARE nodes (a hub dissolved into all of
    # its neighbours, consensus.py:120-124), so this is a relation, not a partition
    simple = eligible & ~general
    m_node = np.flatnonzero(simple[rank_of])
    m_group = base[rank_of[m_node]]
    for r, cl in clusters_of.items():
        for k, members in enumerate(cl):
            m_node = np.concatenate([m_node, np.asarray(members, dtype=np.int64)])
            m_group = np.concatenate([m_group, np.full(len(members), base[r] + k, dtype=np.int64)])
    group_comp = np.repeat(np.arange(n_comp), slots)
    big = np.iinfo(np.int64).max
    g_lo = np.full((n_groups, 3), big, dtype=np.int64)
    g_hi = np.full((n_groups, 3), -big, dtype=np.int64)
    np.minimum.at(g_lo, m_group, boxes[m_node, :3])
    np.maximum.at(g_hi, m_group, boxes[m_node, 3:])
    g_box = np.concatenate([g_lo, g_hi], axis=1)                                  # merge_boxes over the members
    counts = reduce(store.run_counts()[sidx])                                     # runs per object, all ranks
    voters = np.bincount(m_group, weights=(counts[m_node] > 0), minlength=n_groups)
    votes_ok = voters >= pixel_vote_thr if pixel_vote_thr > 1 else np.ones(n_groups, bool)   # vote_by_ranges :611-615
    if pixel_vote_thr == 1:
        runs_in = np.bincount(m_group, weights=counts[m_node], minlength=n_groups)
        if (runs_in == 0).any():
            raise ValueError("need at least one array to concatenate")                       # join_ranges :665-671
        if (runs_in == 1).any():
            raise UnboundLocalError("local variable 'range2' referenced before assignment")  # _join_ranges :659-661

    # ---- the vote: every cluster in one launch.  Each object's runs are tagged with its (first) group in place;
    # the runs of an object that sits in further clusters are appended once per extra membership
    rows = np.flatnonzero(votes_ok[m_group])
    first_row = np.full(n, -1, dtype=np.int64)
    first_row[m_node[rows][::-1]] = rows[::-1]
    per_obj = np.full(store.n_obj, n_groups, dtype=np.int64)                      # dummy group n_groups is discarded
    prim = first_row >= 0
    per_obj[sidx[prim]] = m_group[first_row[prim]]
    v_st, v_ln, v_grp = store.st, store.ln, store.expand(per_obj)
    extra = np.setdiff1d(rows, first_row[prim], assume_unique=True)
    BRANCH_COUNTS['extra_memberships'] += int(len(extra))
    if len(extra):
        off_h = store.off.cpu().numpy()
        lo_e, hi_e = off_h[sidx[m_node[extra]]], off_h[sidx[m_node[extra]] + 1]
        sizes = hi_e - lo_e
        if sizes.sum():
            idx = np.concatenate([np.arange(a0, b0) for a0, b0 in zip(lo_e, hi_e)])
            idx_d = torch.from_numpy(idx).to(store.st.device)
            tag_d = torch.from_numpy(np.repeat(m_group[extra], sizes).astype(np.int32)).to(store.st.device)
            v_st = torch.cat([v_st, store.st[idx_d]])
            v_ln = torch.cat([v_ln, store.ln[idx_d]])
            v_grp = torch.cat([v_grp, tag_d])
    ranges, off = _vote(v_st, v_ln, v_grp, n_groups + 1, pixel_vote_thr)
    off = off[:n_groups + 1]
    ranges = ranges[:int(off[-1])]
    voted = reduce(_segment_sums(ranges, off))                                    # voxels per voted cluster, all ranks
    nonempty = voted > 0

    # ---- clusters of one component whose votes overlap are joined (merge_overlapping, consensus.py:166-195)
    final_of = np.full(n_groups, -1, dtype=np.int64)
    per_comp = np.bincount(group_comp[nonempty], minlength=n_comp)
    joined_any = False
    if (per_comp >= 2).any():
        cand = []
        for r in np.flatnonzero(per_comp >= 2):
            gs = np.flatnonzero((group_comp == r) & nonempty)
            cand.extend(combinations(gs.tolist(), 2))
        cand = np.asarray(cand, dtype=np.int64)
        vstore = RunStore((ranges[:, 0]).contiguous(), (ranges[:, 1] - ranges[:, 0]).contiguous(),
                          torch.from_numpy(np.concatenate([off, [off[-1]]]).astype(np.int64)).to(ranges.device))
        c_inter = reduce(vstore.intersections(cand))
        c_iou = c_inter / (voted[cand[:, 0]] + voted[cand[:, 1]] - c_inter)
        link = cand[(c_iou > MIN_IOU) | (c_inter > MIN_OVERLAP)]
        root = np.arange(n_groups)
        for a, b in link.tolist():                        # union by smaller index: the root is the first cluster
            ra, rb = a, b
            while root[ra] != ra:
                ra = root[ra]
            while root[rb] != rb:
                rb = root[rb]
            if ra != rb:
                root[max(ra, rb)] = min(ra, rb)
                joined_any = True
        for g in range(n_groups):
            r = g
            while root[r] != r:
                r = root[r]
            root[g] = r
    else:
        root = np.arange(n_groups)
    BRANCH_COUNTS['overlap_joins'] += int(joined_any)
    heads = nonempty & (root == np.arange(n_groups))
    final_of[heads] = np.arange(int(heads.sum()))
    final_of[nonempty] = final_of[root[nonempty]]
    n_final = int(heads.sum())
    if n_final == 0:
        return empty
    f_lo = np.full((n_final, 3), big, dtype=np.int64)
    f_hi = np.full((n_final, 3), -big, dtype=np.int64)
    np.minimum.at(f_lo, final_of[nonempty], g_box[nonempty, :3])
    np.maximum.at(f_hi, final_of[nonempty], g_box[nonempty, 3:])
    if joined_any:
        # union of the joined clusters' voted ranges = coverage vote with threshold 1 (join_ranges)
        tag2 = np.where(nonempty, final_of, n_final)
        grp = _expand(off, tag2, ranges.device)
        ranges, off2 = _vote(ranges[:, 0].contiguous(), (ranges[:, 1] - ranges[:, 0]).contiguous(), grp, n_final + 1, 1)
        off_f = off2[:n_final + 1]
        f_area = reduce(_segment_sums(ranges, off_f))
    else:
        off_f = np.concatenate([[0], np.cumsum((off[1:] - off[:-1])[nonempty])])
        assert int(off_f[-1]) == int(off[-1])             # empty groups own no rows: the ranges stay as they are
        f_area = voted[nonempty]
    return ConsensusResult(np.concatenate([f_lo, f_hi], axis=1), f_area, ranges, off_f)


def consensus_semantic(src, boxes, store, pixel_vote_thr=2, reduce=None, store_index=None):
    """merge_semantic_from_trackers (consensus.py:289-346) on tables: at most one instance per tracker, their runs
    are voted on as one group and the result is instance 1."""
    reduce = reduce or (lambda a: a)
    n = len(src)
    empty = ConsensusResult(np.zeros((0, 6)), np.zeros(0), None, np.zeros(1))
    if n == 0:
        return empty
    assert np.bincount(np.asarray(src, dtype=np.int64)).max() <= 1, 'Semantic classes only have 1 label!'
    sidx = np.arange(n, dtype=np.int64) if store_index is None else np.asarray(store_index, dtype=np.int64)
    boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 6)
    box = np.concatenate([boxes[:, :3].min(axis=0), boxes[:, 3:].max(axis=0)])
    counts = reduce(store.run_counts()[sidx])
    if pixel_vote_thr == 1 and counts.sum() < 2:
        raise UnboundLocalError("local variable 'range2' referenced before assignment")      # _join_ranges :659-661
    per_obj = np.full(store.n_obj, 1, dtype=np.int64)
    if pixel_vote_thr == 1 or (counts > 0).sum() >= pixel_vote_thr:                          # vote_by_ranges :611-615
        per_obj[sidx] = 0
    ranges, off = _vote(store.st, store.ln, store.expand(per_obj), 2, pixel_vote_thr)
    ranges = ranges[:int(off[1])]
    area = reduce(_segment_sums(ranges, off[:2]))
    if int(area[0]) == 0:
        raise IndexError("too many indices for array: array is 1-dimensional, but 2 were indexed")   # seg_ranges[:, 0]
    return ConsensusResult(box[None], area, ranges, off[:2])


def _vote(st, ln, grp, n_groups, thr):
    """coverage vote per group (emp_vote_ranges) -> (ranges (m, 2) device, CSR offsets numpy int64)"""
    if st.numel() == 0:
        return torch.zeros((0, 2), dtype=torch.int64, device=st.device), np.zeros(n_groups + 1, np.int64)
    out, off = _hip.vote_ranges(st, (st + ln).contiguous(), grp, n_groups, int(thr))
    off = off.cpu().numpy().astype(np.int64)
    return out[:int(off[-1])], off


def _segment_sums(ranges, off):
    """sum of range lengths per CSR segment -> numpy int64"""
    if ranges.numel() == 0:
        return np.zeros(len(off) - 1, dtype=np.int64)
    cs = torch.cumsum(ranges[:, 1] - ranges[:, 0], 0)
    cs = torch.cat([torch.zeros(1, dtype=cs.dtype, device=cs.device), cs])
    idx = torch.from_numpy(off).to(cs.device)
    tot = cs[idx]
    return (tot[1:] - tot[:-1]).cpu().numpy()


def _expand(off, val, dev):
    n_seg = len(off) - 1
    m = int(off[-1])
    out = torch.empty((max(m, 1),), dtype=torch.int32, device=dev)
    off_d = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).to(dev)
    val_d = torch.from_numpy(np.ascontiguousarray(val, dtype=np.int32)).to(dev)
    _hip.call('emp_track_expand', _hip._ptr(off_d), _hip._ptr(val_d), n_seg, m, _hip._ptr(out), _hip.stream())
    return out[:m]


def _ranges(starts, runs):
    starts = np.asarray(starts, dtype=np.int64)
    return np.stack([starts, starts + np.asarray(runs, dtype=np.int64)], axis=1)


def merge_semantic_from_trackers(semantic_trackers, pixel_vote_thr=2):
    """consensus.py:289-346"""
    boxes, ranges = [], []
    for tr in semantic_trackers:
        assert len(tr.instances.keys()) <= 1, 'Semantic classes only have 1 label!'
        for attrs in tr.instances.values():
            boxes.append(attrs['box'])
            ranges.append(_ranges(attrs['starts'], attrs['runs']))
    if not boxes:
        return {}
    merged_box = boxes[0]
    for box in boxes[1:]:
        merged_box = merge_boxes(merged_box, box)
    from .array_utils import vote_by_ranges
    # like the reference, an empty vote (fewer trackers than votes) fails on the 2-D indexing below
    seg_ranges = vote_by_ranges(ranges, pixel_vote_thr)
    return {1: {'box': merged_box, 'starts': seg_ranges[:, 0], 'runs': seg_ranges[:, 1] - seg_ranges[:, 0]}}


def merge_objects_from_trackers(object_trackers, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False):
    """consensus.py:348-469 for trackers given as dicts of numpy run lists (the per-slice protocol and the reference's
    own tests); the whole-stack path calls consensus_objects on device-resident PlaneTracks instead."""
    src, boxes, starts, runs = [], [], [], []
    for tr_index, tr in enumerate(object_trackers):
        for attr in tr.instances.values():
            src.append(tr_index)
            boxes.append(attr['box'])
            starts.append(attr['starts'])
            runs.append(attr['runs'])
    if not boxes:
        return {}
    areas = np.array([int(np.sum(r)) for r in runs], dtype=np.int64)
    res = consensus_objects(np.array(src), np.array(boxes), areas, RunStore.from_lists(starts, runs),
                            len(object_trackers), pixel_vote_thr, cluster_iou_thr, bypass)
    return res.instances()


def merge_semantic_from_tiles(tiles):
    """consensus.py:471-524 -- union of the (single) semantic instance over all tiles."""
    from .array_utils import join_ranges
    label_id = None
    boxes, ranges = [], []
    for tile_instances in tiles:
        for instance_id, attr in tile_instances.items():
            if label_id is None:
                label_id = instance_id
            boxes.append(attr['box'])
            ranges.append(_ranges(attr['starts'], attr['runs']))
    if len(boxes) == 0:
        return {}
    merged_box = np.array(boxes)[0]
    for box in np.array(boxes)[1:]:
        merged_box = merge_boxes(merged_box, box)
    seg_ranges = join_ranges(ranges)
    return {label_id: {'box': merged_box, 'starts': seg_ranges[:, 0], 'runs': seg_ranges[:, 1] - seg_ranges[:, 0]}}


def merge_objects_from_tiles(tiles, overlap_rle=None):
    """consensus.py:526-625 -- objects of overlapping tiles that intersect are joined; with `overlap_rle`, an object
    seen in a single tile that lies by more than 10 % inside the overlap region is dropped.  Pair intersections and
    all joins run on the GPU in one launch each."""
    tile_indices, object_labels, object_boxes, object_starts, object_runs = [], [], [], [], []
    for tile_idx, tile_instances in enumerate(tiles):
        for instance_id, attr in tile_instances.items():
            tile_indices.append(tile_idx)
            object_labels.append(int(instance_id))
            object_boxes.append(attr['box'])
            object_starts.append(attr['starts'])
            object_runs.append(attr['runs'])
    tile_indices = np.array(tile_indices)
    object_labels = np.array(object_labels)
    object_boxes = np.array(object_boxes)
    if len(object_boxes) == 0:
        return {}
    graph = object_iou_graph(tile_indices, object_labels, object_boxes, object_starts, object_runs)
    if overlap_rle is not None:
        overlap_starts, overlap_runs = overlap_rle

    clusters = [list(c) for c in nx.connected_components(graph)]
    groups = []
    for cluster in clusters:
        lst = [_ranges(graph.nodes[n]['starts'], graph.nodes[n]['runs']) for n in cluster]
        if sum(len(r) for r in lst) < 2:
            raise UnboundLocalError("local variable 'range2' referenced before assignment")   # _join_ranges :659-661
        groups.append(lst)
    joined = vote_groups(groups, 1)

    # rle_ioa(overlap, object) of every object seen in one tile only (consensus.py:606-613), all in ONE launch
    dropped = set()
    if overlap_rle is not None:
        single = [ci for ci, (cluster, vr) in enumerate(zip(clusters, joined)) if len(cluster) < 2 and np.any(vr)]
        if single:
            from .array_utils import rle_pair_intersections
            inter = rle_pair_intersections([np.asarray(overlap_starts)] + [joined[ci][:, 0] for ci in single],
                                           [np.asarray(overlap_runs)] + [joined[ci][:, 1] - joined[ci][:, 0] for ci in single],
                                           [[0, k + 1] for k in range(len(single))])
            for ci, it in zip(single, inter):
                area = int(np.sum(joined[ci][:, 1] - joined[ci][:, 0]))
                if np.float64(int(it)) / np.float64(area) > 0.1:
                    dropped.add(ci)

    instance_id = int(np.min(object_labels))
    instances = {}
    for ci, (cluster, voted_ranges) in enumerate(zip(clusters, joined)):
        merged_box = graph.nodes[cluster[0]]['box']
        for node_id in cluster[1:]:
            merged_box = merge_boxes(merged_box, graph.nodes[node_id]['box'])
        if ci in dropped:
            voted_ranges = np.zeros((0, 2), dtype=np.int64)
        if np.any(voted_ranges):
            instances[instance_id] = {'box': tuple(int(x) for x in merged_box), 'starts': voted_ranges[:, 0],
                                      'runs': voted_ranges[:, 1] - voted_ranges[:, 0]}
            instance_id += 1
    return instances


# the name scripts/inference3d_multigpu.py:30 imports from `empanada.aggregation.consensus`
merge_objects3d = merge_objects_from_trackers
