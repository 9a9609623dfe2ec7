"""CPU oracle for the cross-plane consensus (TEST INFRASTRUCTURE ONLY).

Restatement of empanada/consensus.py:10-469 (tracker consensus) on numpy + networkx
(networkx is the same third-party library the reference calls; graph enumeration
order is part of the result, so it is used, not re-derived).
"""
from itertools import combinations

import networkx as nx
import numpy as np

from .rle_ops import box_pairs, merge_boxes, merge_rles, rle_iou, vote_by_ranges

MIN_OVERLAP = 100
MIN_IOU = 1e-2


def average_edge_between_clusters(G, cluster1, cluster2, key='iou'):
    """consensus.py:10-33"""
    weights = [G[a][b][key] if G.has_edge(a, b) else 0 for a in cluster1 for b in cluster2]
    return sum(weights) / len(weights)


def create_graph_of_clusters(G, cluster_iou_thr):
    """consensus.py:35-74"""
    H = G.copy()
    for u, v, d in G.edges(data=True):
        if d['iou'] <= cluster_iou_thr:
            H.remove_edge(u, v)
    CG = nx.Graph()
    for i, cluster in enumerate(nx.connected_components(H)):
        CG.add_node(i, cluster=cluster)
    for n1, n2 in combinations(CG.nodes, 2):
        c1, c2 = CG.nodes[n1]['cluster'], CG.nodes[n2]['cluster']
        iou_w = average_edge_between_clusters(G, c1, c2, 'iou')
        ov_w = average_edge_between_clusters(G, c1, c2, 'overlap')
        if iou_w > MIN_IOU or ov_w > MIN_OVERLAP:
            CG.add_edge(n1, n2, iou=iou_w, overlap=ov_w)
    return CG


def push_cluster(G, src, dst):
    """consensus.py:76-84"""
    G.nodes[dst]['cluster'] = G.nodes[dst]['cluster'].union(G.nodes[src]['cluster'])
    G.remove_edge(src, dst)
    return G


def merge_clusters(G):
    """consensus.py:86-142 -- including :134-140, which re-adds the edge (most_connected, neighbor)
    instead of (most_connected, sn), so second neighbours are dropped with the neighbour."""
    H = G.copy()
    while len(H.edges()) > 0:
        most = sorted(H.nodes, key=lambda x: len(list(H.neighbors(x))), reverse=True)[0]
        nbrs = sorted(H.neighbors(most), key=lambda x: len(H.nodes[x]['cluster']), reverse=True)
        push_most = len(H.nodes[nbrs[0]]['cluster']) > len(H.nodes[most]['cluster'])
        if push_most:
            for nb in nbrs:
                push_cluster(H, most, nb)
            H.remove_node(most)
        else:
            for nb in nbrs:
                push_cluster(H, nb, most)
                for sn in list(H.neighbors(nb)):
                    if not H.has_edge(most, sn):
                        H.add_edge(most, nb, iou=H[nb][sn]['iou'])
                H.remove_node(nb)
    return H


def merge_instances(instances_dict):
    """consensus.py:144-164"""
    if len(instances_dict) < 2:
        return list(instances_dict.values())[0]
    box = starts = runs = None
    for a in instances_dict.values():
        if box is None:
            box, starts, runs = a['box'], a['starts'], a['runs']
        else:
            box = merge_boxes(box, a['box'])
            starts, runs = merge_rles(starts, runs, a['starts'], a['runs'])
    return dict(box=box, starts=starts, runs=runs)


def merge_overlapping(cluster_instances):
    """consensus.py:166-195"""
    if len(cluster_instances) < 2:
        return list(cluster_instances.values())
    ids = list(cluster_instances.keys())
    g = nx.Graph()
    g.add_nodes_from(ids)
    for ci, cj in combinations(ids, 2):
        iou, inter = rle_iou(cluster_instances[ci]['starts'], cluster_instances[ci]['runs'],
                             cluster_instances[cj]['starts'], cluster_instances[cj]['runs'],
                             return_intersection=True)
        if iou > MIN_IOU or inter > MIN_OVERLAP:
            g.add_edge(ci, cj)
    out = []
    for comp in nx.connected_components(g):
        out.append(merge_instances({k: v for k, v in cluster_instances.items() if k in comp}))
    return out


def bounding_box_screening(boxes, source_indices):
    """consensus.py:197-231 -- pairs with positive box intersection from different sources, i<j, unique."""
    rows, cols, _, _ = box_pairs(boxes)
    m = np.stack([rows, cols], axis=1)
    m = m[source_indices[m[:, 0]] != source_indices[m[:, 1]]]
    m = np.sort(m, axis=-1)
    return np.unique(m, axis=0)


def object_iou_graph(source_indices, object_labels, object_boxes, object_starts, object_runs):
    """consensus.py:233-287"""
    matches = bounding_box_screening(object_boxes, source_indices)
    g = nx.Graph()
    for i in range(len(object_labels)):
        g.add_node(i, box=object_boxes[i], starts=object_starts[i], runs=object_runs[i])
    for r1, r2 in matches:
        r1, r2 = int(r1), int(r2)
        iou, inter = rle_iou(object_starts[r1], object_runs[r1], object_starts[r2], object_runs[r2],
                             return_intersection=True)
        if iou > 0:
            g.add_edge(r1, r2, iou=iou, overlap=inter)
    return g


def merge_semantic_from_trackers(semantic_trackers, pixel_vote_thr=2):
    """consensus.py:289-346"""
    boxes, starts, runs = [], [], []
    for tr in semantic_trackers:
        assert len(tr.instances.keys()) <= 1, 'Semantic classes only have 1 label!'
        for a in tr.instances.values():
            boxes.append(a['box'])
            starts.append(a['starts'])
            runs.append(a['runs'])
    if not boxes:
        return {}
    box = boxes[0]
    for b in boxes[1:]:
        box = merge_boxes(box, b)
    rng = vote_by_ranges([np.stack([s, s + r], axis=1) for s, r in zip(starts, runs)], pixel_vote_thr)
    return {1: {'box': box, 'starts': rng[:, 0], 'runs': rng[:, 1] - rng[:, 0]}}


def merge_objects_from_trackers(object_trackers, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False):
    """consensus.py:348-469"""
    n_votes = len(object_trackers)
    min_cluster_size = 1 if bypass else (n_votes // 2) + 1
    if pixel_vote_thr < min_cluster_size:
        cluster_iou_thr = 0

    tr_idx, labels, boxes, starts, runs = [], [], [], [], []
    for ti, tr in enumerate(object_trackers):
        for iid, a in tr.instances.items():
            tr_idx.append(ti)
            labels.append(int(iid))
            boxes.append(a['box'])
            starts.append(a['starts'])
            runs.append(a['runs'])
    tr_idx, labels, boxes = np.array(tr_idx), np.array(labels), np.array(boxes)
    if len(boxes) == 0:
        return {}

    graph = object_iou_graph(tr_idx, labels, boxes, starts, runs)
    instance_id = 1
    instances = {}
    for comp in nx.connected_components(graph):
        if len(comp) < min_cluster_size:
            continue
        cg = merge_clusters(create_graph_of_clusters(graph.subgraph(comp), cluster_iou_thr))
        cluster_id = 1
        cluster_instances = {}
        for node in cg.nodes:
            cluster = list(cg.nodes[node]['cluster'])
            if len(cluster) < min_cluster_size:
                continue
            box = graph.nodes[cluster[0]]['box']
            for nid in cluster[1:]:
                box = merge_boxes(box, graph.nodes[nid]['box'])
            all_ranges = [np.stack([graph.nodes[n]['starts'],
                                    graph.nodes[n]['starts'] + graph.nodes[n]['runs']], axis=1)
                          for n in cluster]
            voted = vote_by_ranges(all_ranges, pixel_vote_thr)
            if len(voted) > 0:
                cluster_instances[cluster_id] = {
                    'box': tuple(int(x) for x in box),
                    'starts': voted[:, 0], 'runs': voted[:, 1] - voted[:, 0]}
                cluster_id += 1
        for attrs in merge_overlapping(cluster_instances):
            instances[instance_id] = attrs
            instance_id += 1
    return instances


def _join(list_of_ranges, single_run):
    """join_ranges (array_utils.py:665-671), which raises UnboundLocalError on ONE range in total (:659-661) -- the
    reference's behaviour, `single_run='raise'`; 'keep' returns that range (the volume drivers' choice, documented in
    empanada_amd/inference/tiled.py)"""
    from .rle_ops import join_ranges
    if single_run == 'keep' and sum(len(r) for r in list_of_ranges) == 1:
        return np.concatenate(list_of_ranges).reshape(1, 2)
    return join_ranges(list_of_ranges)


def merge_semantic_from_tiles(tiles, single_run='raise'):
    """consensus.py:471-524"""
    from .rle_ops import join_ranges
    label_id, boxes, rngs = None, [], []
    for tile_instances in tiles:
        for iid, a in tile_instances.items():
            if label_id is None:
                label_id = iid
            boxes.append(a['box'])
            rngs.append(np.stack([a['starts'], a['starts'] + a['runs']], axis=1))
    boxes = np.array(boxes)
    if len(boxes) == 0:
        return {}
    box = boxes[0]
    for b in boxes[1:]:
        box = merge_boxes(box, b)
    r = _join(rngs, single_run)
    return {label_id: {'box': box, 'starts': r[:, 0], 'runs': r[:, 1] - r[:, 0]}}


def merge_objects_from_tiles(tiles, overlap_rle=None, single_run='raise'):
    """consensus.py:526-625"""
    from .rle_ops import join_ranges, ranges_to_rle, rle_ioa
    tile_idx, labels, boxes, starts, runs = [], [], [], [], []
    for ti, tile_instances in enumerate(tiles):
        for iid, a in tile_instances.items():
            tile_idx.append(ti)
            labels.append(int(iid))
            boxes.append(a['box'])
            starts.append(a['starts'])
            runs.append(a['runs'])
    tile_idx, labels, boxes = np.array(tile_idx), np.array(labels), np.array(boxes)
    if len(boxes) == 0:
        return {}
    graph = object_iou_graph(tile_idx, labels, boxes, starts, runs)
    instance_id = int(np.min(labels))
    instances = {}
    for cluster in nx.connected_components(graph):
        cluster = list(cluster)
        box = graph.nodes[cluster[0]]['box']
        for nid in cluster[1:]:
            box = merge_boxes(box, graph.nodes[nid]['box'])
        voted = _join([np.stack([graph.nodes[n]['starts'], graph.nodes[n]['starts'] + graph.nodes[n]['runs']],
                                axis=1) for n in cluster], single_run)
        if overlap_rle is not None and len(cluster) < 2 and np.any(voted):
            rle = ranges_to_rle(voted)
            if rle_ioa(np.asarray(overlap_rle[0]), np.asarray(overlap_rle[1]), rle[:, 0], rle[:, 1]) > 0.1:
                voted = []
        if np.any(voted):
            instances[instance_id] = {'box': tuple(int(x) for x in box), 'starts': voted[:, 0],
                                      'runs': voted[:, 1] - voted[:, 0]}
            instance_id += 1
    return instances


def create_instance_consensus(class_trackers, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False):
    """patterns.py:168-186"""
    from .rle_seg import InstanceTracker
    t0 = class_trackers[0]
    out = InstanceTracker(t0.class_id, t0.label_divisor, t0.shape3d, 'xy')
    out.instances = merge_objects_from_trackers(class_trackers, pixel_vote_thr, cluster_iou_thr, bypass)
    return out


def create_semantic_consensus(class_trackers, pixel_vote_thr=2):
    """patterns.py:188-202"""
    from .rle_seg import InstanceTracker
    t0 = class_trackers[0]
    out = InstanceTracker(t0.class_id, t0.label_divisor, t0.shape3d, 'xy')
    out.instances = merge_semantic_from_trackers(class_trackers, pixel_vote_thr)
    return out
