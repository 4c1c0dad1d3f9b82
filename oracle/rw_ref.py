"""TEST INFRASTRUCTURE ONLY (oracle) -- dense restatement of IRNet's random-walk propagation exactly as upstream
computes it (jiwoon-ahn/irn misc/indexing.py; the reference's `misc` package is missing from its tree, so
this parity is UNPINNED -- only the affinity step is in-tree, net/vgg16_irn.py:247-261):
  edge_to_affinity (index_select over path indices, 1 - max over the path) -> affinity_sparse2dense (symmetric
  dense matrix with unit diagonal on the PADDED grid, cropped back) -> to_transition_matrix (pow beta, column
  normalise, exp_times squarings) -> x * (1 - edge) @ T.   O((hw)^2) memory: small grids only."""
import numpy as np
import torch
import torch.nn.functional as F

from wsscam.misc.indexing import PathIndex  # the host-side path tables are shared (numpy, no device code)


def edge_to_affinity(edge, paths_indices):
    """upstream edge_to_affinity == AffinityDisplacementLoss.to_affinity (vgg16_irn.py:247-261)."""
    aff_list = []
    edge = edge.view(edge.size(0), -1)
    for ind in paths_indices:
        ind = torch.from_numpy(ind) if isinstance(ind, np.ndarray) else ind
        ind_flat = ind.view(-1)
        dist = torch.index_select(edge, dim=-1, index=ind_flat)
        dist = dist.view(dist.size(0), ind.size(0), ind.size(1), ind.size(2))
        aff = torch.squeeze(1 - F.max_pool2d(dist, (dist.size(2), 1)), dim=2)
        aff_list.append(aff)
    return torch.cat(aff_list, dim=1)


def affinity_sparse2dense(affinity_sparse, ind_from, ind_to, n_vertices):
    ind_from = torch.from_numpy(ind_from)
    ind_to = torch.from_numpy(ind_to)
    affinity_sparse = affinity_sparse.view(-1)
    ind_from = ind_from.repeat(ind_to.size(0)).view(-1)
    ind_to = ind_to.view(-1)
    dense = torch.zeros(n_vertices, n_vertices, dtype=affinity_sparse.dtype)
    dense[ind_from, ind_to] += affinity_sparse
    dense[ind_to, ind_from] += affinity_sparse
    dense += torch.eye(n_vertices, dtype=affinity_sparse.dtype)
    return dense


def to_transition_matrix(affinity_dense, beta, times):
    scaled = torch.pow(affinity_dense, beta)
    trans = scaled / torch.sum(scaled, dim=0, keepdim=True)
    for _ in range(times):
        trans = torch.matmul(trans, trans)
    return trans


def propagate_to_edge(x, edge, radius=5, beta=10, exp_times=8, dtype=torch.float32):
    """x (K,h,w), edge (1,h,w) torch -> rw (K,1,h,w).  dtype=float32 is what the reference computes; float64
    gives the exact value of the same expression (the 8 fp32 squarings themselves carry ~1e-5 of round-off)."""
    x = x.to(dtype)
    edge = edge.to(dtype)
    height, width = x.shape[-2:]
    hor_padded = width + radius * 2
    vert_padded = height + radius
    pi = PathIndex(radius=radius, default_size=(vert_padded, hor_padded))
    edge_padded = F.pad(edge, (radius, radius, 0, radius), mode="constant", value=1.0)
    sparse_aff = edge_to_affinity(torch.unsqueeze(edge_padded, 0), pi.path_indices)
    dense = affinity_sparse2dense(sparse_aff, pi.src_indices, pi.dst_indices, vert_padded * hor_padded)
    dense = dense.view(vert_padded, hor_padded, vert_padded, hor_padded)
    dense = dense[:-radius, radius:-radius, :-radius, radius:-radius]
    dense = dense.reshape(height * width, height * width)
    trans = to_transition_matrix(dense, beta=beta, times=exp_times)
    xm = x.view(-1, height, width) * (1 - edge)
    rw = torch.matmul(xm.view(-1, height * width), trans)
    return rw.view(rw.size(0), 1, height, width)
