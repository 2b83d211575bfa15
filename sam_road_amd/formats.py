"""Output formats consumed by the reference's downstream APLS/TOPO scripts.

sat2graph adjacency-dict format, mirroring reference graph_utils.py:82-93,383-434:
    {(row, col): [(row, col), ...]}  with integer-rounded coordinates and undirected edges.
"""
import numpy as np


def edge_list_to_adj_table(nodes, edges):
    adj = [set() for _ in range(len(nodes))]
    # plain Python ints (same hashes, hence the same set order as the reference's numpy ints; 10x faster to iterate)
    for a, b in (edges.tolist() if isinstance(edges, np.ndarray) else edges):
        adj[a].add(b)
    return adj


def convert_to_sat2graph_format(nodes, edges):
    """nodes [N,2] (row, col), edges [E,2] index pairs -> dict; every edge is stored in both directions."""
    edges = np.asarray(edges).reshape(-1, 2)
    both = np.concatenate((edges, edges[:, ::-1]), axis=0)
    adj = edge_list_to_adj_table(nodes, both)
    int_nodes = [(round(a), round(b)) for a, b in (nodes.tolist() if isinstance(nodes, np.ndarray) else nodes)]
    return {int_nodes[i]: [int_nodes[j] for j in nbrs] for i, nbrs in enumerate(adj)}


def convert_from_sat2graph_format(graph):
    """Inverse: dict -> (nodes [N,2], edges list of (src, dst)); edges are NOT de-duplicated."""
    index = {}
    for node, nbrs in graph.items():
        index.setdefault(node, len(index))
        for nb in nbrs:
            index.setdefault(nb, len(index))
    edges = [(index[node], index[nb]) for node, nbrs in graph.items() for nb in nbrs]
    nodes = [None] * len(index)
    for node, i in index.items():
        nodes[i] = node
    return np.array(nodes), edges
