"""Cross-plane consensus, reference names and semantics (``empanada/consensus.py``):
``merge_objects_from_trackers`` :348-469, ``merge_semantic_from_trackers`` :289-346,
``object_iou_graph`` :233-287, ``bounding_box_screening`` :197-231, ``create_graph_of_clusters`` :35-74,
``merge_clusters`` :86-142, ``merge_overlapping`` :166-195, ``merge_instances`` :144-164.

Split of work
  * libemp_hip.so: box screening (emp_box_pairs), every pairwise run-length intersection
    (emp_rle_pair_intersections, one launch over all screened pairs), every cluster's voxel vote and
    every final union (emp_vote_ranges, one launch over all clusters).
  * host: the graph logic on O(#objects) nodes with networkx -- the same third-party library the
    reference calls (consensus.py:2); its enumeration order defines the final instance ids, so it is
    used as is rather than re-derived.
"""
from itertools import combinations

import networkx as nx
import numpy as np
import torch

from . import _hip
from .array_utils import merge_boxes, vote_groups

__all__ = ['merge_objects_from_trackers', 'merge_semantic_from_trackers', 'merge_objects_from_tiles',
           'merge_semantic_from_tiles', 'merge_objects3d', 'object_iou_graph',
           'bounding_box_screening', 'create_graph_of_clusters', 'merge_clusters']

MIN_OVERLAP = 100
MIN_IOU = 1e-2


class _RunStore:
    """Device-resident, start-sorted run lists of a set of instances + batched pair intersections."""

    def __init__(self, starts_list, runs_list):
        _hip.require_gpu()
        sizes = np.array([len(s) for s in starts_list], dtype=np.int64)
        self.off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.areas = np.array([int(np.sum(r)) for r in runs_list], dtype=np.int64)
        n = int(self.off[-1])
        if n:
            st = np.concatenate([np.asarray(s, dtype=np.int64) for s in starts_list])
            ln = np.concatenate([np.asarray(r, dtype=np.int64) for r in runs_list])
            inst = np.repeat(np.arange(len(sizes), dtype=np.int64), sizes)
            if st.max() >= 2 ** 40 or len(sizes) >= 2 ** 23:
                raise ValueError("volume or instance count too large for the 40/23-bit sort key")
            # stable radix sort by (instance, start): emp_sort_u64_i32
            keys = torch.from_numpy(((inst << 40) | st)).cuda().view(torch.uint64)
            vals = torch.arange(n, dtype=torch.int32, device='cuda')
            _, order = _hip.sort_u64_i32(keys, vals, 0, 63)
            order = order.long()
            self.st = torch.from_numpy(st).cuda()[order].contiguous()
            self.ln = torch.from_numpy(ln).cuda()[order].contiguous()
        else:
            self.st = torch.zeros(0, dtype=torch.int64, device='cuda')
            self.ln = torch.zeros(0, dtype=torch.int64, device='cuda')
        self.off_dev = torch.from_numpy(self.off).cuda()

    def intersections(self, pairs):
        pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        if len(pairs) == 0:
            return np.zeros(0, dtype=np.int64)
        out = _hip.rle_pair_intersections(self.st, self.ln, self.off_dev, torch.from_numpy(pairs).cuda())
        return out.cpu().numpy()

    def iou(self, pairs):
        pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        inter = self.intersections(pairs)
        union = self.areas[pairs[:, 0]] + self.areas[pairs[:, 1]] - inter
        return inter / union, inter          # int64 / int64 -> fp64, as rle_iou (array_utils.py:424-427)


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
        store = store or _RunStore(object_starts, object_runs)
        ious, inters = store.iou(box_matches)
        for (r1, r2), pair_iou, inter_area in zip(box_matches, ious, inters):
            if pair_iou > 0:
                graph.add_edge(int(r1), int(r2), iou=pair_iou, overlap=inter_area)
    return graph


def average_edge_between_clusters(G, cluster1, cluster2, key='iou'):
    """consensus.py:10-33"""
    weights = [G[a][b][key] if G.has_edge(a, b) else 0 for a in cluster1 for b in cluster2]
    return sum(weights) / len(weights)


def create_graph_of_clusters(G, cluster_iou_thr):
    """consensus.py:35-74"""
    H = G.copy()
    for (u, v, d) in G.edges(data=True):
        if d['iou'] <= cluster_iou_thr:
            H.remove_edge(u, v)
    cluster_graph = nx.Graph()
    for i, cluster in enumerate(nx.connected_components(H)):
        cluster_graph.add_node(i, cluster=cluster)
    for node1, node2 in combinations(cluster_graph.nodes, 2):
        cluster1 = cluster_graph.nodes[node1]['cluster']
        cluster2 = cluster_graph.nodes[node2]['cluster']
        iou_weight = average_edge_between_clusters(G, cluster1, cluster2, 'iou')
        overlap_weight = average_edge_between_clusters(G, cluster1, cluster2, 'overlap')
        if iou_weight > MIN_IOU or overlap_weight > MIN_OVERLAP:
            cluster_graph.add_edge(node1, node2, iou=iou_weight, overlap=overlap_weight)
    return cluster_graph


def push_cluster(G, src, dst):
    """consensus.py:76-84"""
    G.nodes[dst]['cluster'] = G.nodes[dst]['cluster'].union(G.nodes[src]['cluster'])
    G.remove_edge(src, dst)
    return G


def merge_clusters(G):
    """consensus.py:86-142 (the edge re-added at :138 is (most_connected, neighbor), reproduced)."""
    H = G.copy()
    while len(H.edges()) > 0:
        most_connected = sorted(H.nodes, key=lambda x: len(list(H.neighbors(x))), reverse=True)[0]
        neighbors = sorted(H.neighbors(most_connected), key=lambda x: len(H.nodes[x]['cluster']), reverse=True)
        most_connected_cluster = H.nodes[most_connected]['cluster']
        push_most_connected = len(H.nodes[neighbors[0]]['cluster']) > len(most_connected_cluster)
        if push_most_connected:
            for neighbor in neighbors:
                push_cluster(H, most_connected, neighbor)
            H.remove_node(most_connected)
        else:
            for neighbor in neighbors:
                push_cluster(H, neighbor, most_connected)
                for sn in list(H.neighbors(neighbor)):
                    if not H.has_edge(most_connected, sn):
                        H.add_edge(most_connected, neighbor, iou=H[neighbor][sn]['iou'])
                H.remove_node(neighbor)
    return H


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
    """consensus.py:348-469.  Same graph walk as the reference; the three O(#voxel-runs) steps are batched on
    the GPU: (1) all screened pair IoUs, (2) all cluster votes, (3) overlaps + unions of voted instances."""
    n_votes = len(object_trackers)
    min_cluster_size = 1 if bypass else (n_votes // 2) + 1
    if pixel_vote_thr < min_cluster_size:
        cluster_iou_thr = 0

    tracker_indices, object_labels, object_boxes, object_starts, object_runs = [], [], [], [], []
    for tr_index, tr in enumerate(object_trackers):
        for instance_id, attr in tr.instances.items():
            tracker_indices.append(tr_index)
            object_labels.append(int(instance_id))
            object_boxes.append(attr['box'])
            object_starts.append(attr['starts'])
            object_runs.append(attr['runs'])
    tracker_indices = np.array(tracker_indices)
    object_labels = np.array(object_labels)
    object_boxes = np.array(object_boxes)
    if len(object_boxes) == 0:
        return {}

    graph = object_iou_graph(tracker_indices, object_labels, object_boxes, object_starts, object_runs)

    # ---- pass 1 (host): walk the graph exactly like the reference and collect the clusters to vote on
    comps = []            # per connected component: list of (merged_box, member node list)
    for comp in nx.connected_components(graph):
        if len(comp) < min_cluster_size:
            continue
        if all(iou > cluster_iou_thr for _, _, iou in graph.edges(comp, data='iou')):
            # every edge survives the IoU cut, so the component stays one cluster (create_graph_of_clusters yields a
            # single node, merge_clusters has nothing to do): skip the two graph copies.  The member order does
            # not matter downstream (box merging is commutative, the vote counts coverage).
            cluster_sets = [set(comp)]
        else:
            cluster_graph = merge_clusters(create_graph_of_clusters(graph.subgraph(comp), cluster_iou_thr))
            cluster_sets = [cluster_graph.nodes[node]['cluster'] for node in cluster_graph.nodes]
        clusters = []
        for cset in cluster_sets:
            cluster = list(cset)
            if len(cluster) < min_cluster_size:
                continue
            merged_box = graph.nodes[cluster[0]]['box']
            for node_id in cluster[1:]:
                merged_box = merge_boxes(merged_box, graph.nodes[node_id]['box'])
            clusters.append((merged_box, cluster))
        comps.append(clusters)

    # ---- pass 2 (GPU): vote inside every cluster at once
    flat = [c for clusters in comps for c in clusters]
    groups = []
    for _, cluster in flat:
        lst = [_ranges(object_starts[n], object_runs[n]) for n in cluster]
        lst = [r for r in lst if len(r) > 0]
        if pixel_vote_thr > 1 and len(lst) < pixel_vote_thr:
            lst = []                             # vote_by_ranges :611-615
        groups.append(lst)
    if pixel_vote_thr == 1:
        for lst in groups:
            if sum(len(r) for r in lst) == 1:
                raise UnboundLocalError("local variable 'range2' referenced before assignment")  # _join_ranges
    voted = vote_groups(groups, pixel_vote_thr) if flat else []

    # ---- pass 3: overlaps between the voted instances of each component, then unions
    k = 0
    comp_instances = []
    for clusters in comps:
        cluster_instances = {}
        cluster_id = 1
        for merged_box, _ in clusters:
            vr = voted[k]
            k += 1
            if len(vr) > 0:
                cluster_instances[cluster_id] = {
                    'box': tuple(int(x) for x in merged_box), 'starts': vr[:, 0], 'runs': vr[:, 1] - vr[:, 0]}
                cluster_id += 1
        comp_instances.append(cluster_instances)

    # all candidate pairs of all components in one launch
    node_starts, node_runs, base, pair_list = [], [], [], []
    for ci in comp_instances:
        base.append(len(node_starts))
        ids = list(ci.keys())
        for i in ids:
            node_starts.append(ci[i]['starts'])
            node_runs.append(ci[i]['runs'])
        if len(ids) >= 2:
            for a, b in combinations(range(len(ids)), 2):
                pair_list.append((base[-1] + a, base[-1] + b))
    pair_iou = {}
    if pair_list:
        store = _RunStore(node_starts, node_runs)
        ious, inters = store.iou(pair_list)
        pair_iou = {p: (i, n) for p, i, n in zip(pair_list, ious, inters)}

    instance_id = 1
    instances = {}
    to_join = []          # (instance_id, list of range arrays)
    for ci, b0 in zip(comp_instances, base):
        if len(ci) < 2:
            merged = list(ci.values())
        else:
            ids = list(ci.keys())
            merge_graph = nx.Graph()
            merge_graph.add_nodes_from(ids)
            for a, b in combinations(range(len(ids)), 2):
                iou, inter = pair_iou[(b0 + a, b0 + b)]
                if iou > MIN_IOU or inter > MIN_OVERLAP:
                    merge_graph.add_edge(ids[a], ids[b])
            merged = []
            for comp in nx.connected_components(merge_graph):
                members = {key: v for key, v in ci.items() if key in comp}
                if len(members) < 2:
                    merged.append(list(members.values())[0])
                else:
                    box = None
                    for attrs in members.values():
                        box = attrs['box'] if box is None else merge_boxes(box, attrs['box'])
                    merged.append({'box': box, 'join': [_ranges(a['starts'], a['runs']) for a in members.values()]})
        for attrs in merged:
            if 'join' in attrs:
                to_join.append((instance_id, attrs.pop('join')))
            instances[instance_id] = attrs
            instance_id += 1
    if to_join:
        joined = vote_groups([lst for _, lst in to_join], 1)
        for (iid, _), rng in zip(to_join, joined):
            instances[iid]['starts'] = rng[:, 0]
            instances[iid]['runs'] = rng[:, 1] - rng[:, 0]
            instances[iid] = dict(box=instances[iid]['box'], starts=instances[iid]['starts'],
                                  runs=instances[iid]['runs'])
    return instances


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
    from .array_utils import rle_ioa
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

    instance_id = int(np.min(object_labels))
    instances = {}
    for cluster, voted_ranges in zip(clusters, joined):
        merged_box = graph.nodes[cluster[0]]['box']
        for node_id in cluster[1:]:
            merged_box = merge_boxes(merged_box, graph.nodes[node_id]['box'])
        if overlap_rle is not None and len(cluster) < 2 and np.any(voted_ranges):
            ov_ioa = rle_ioa(np.asarray(overlap_starts), np.asarray(overlap_runs), voted_ranges[:, 0],
                             voted_ranges[:, 1] - voted_ranges[:, 0])
            if ov_ioa > 0.1:
                voted_ranges = np.zeros((0, 2), dtype=np.int64)
        if np.any(voted_ranges):
            instances[instance_id] = {'box': tuple(int(x) for x in merged_box), 'starts': voted_ranges[:, 0],
                                      'runs': voted_ranges[:, 1] - voted_ranges[:, 0]}
            instance_id += 1
    return instances


# the name scripts/inference3d_multigpu.py:30 imports from `empanada.aggregation.consensus`
merge_objects3d = merge_objects_from_trackers
