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


def propagate_to_edge_sparse(x, edge, radius=5, beta=10, exp_times=8):
    """The same expression at sizes where the dense (hw x hw) matrix does not fit a test (94 x 125: 11 750^2 entries, 26 TFLOP
    of squarings): the transition matrix built as a SPARSE float64 matrix from the very same affinities, index tables, padding
    and crop -- dense[ind_from, ind_to] += a; dense[ind_to, ind_from] += a; + I; crop; pow beta; column-normalise -- and
    x (1 - edge) @ T^(2^exp_times) evaluated as 2^exp_times vector-matrix products (matrix powers commute with the product;
    float64 makes the evaluation order immaterial at the tolerances used: equal to the dense float64 form to 1e-12 on small
    grids, tests/test_irn_oracle.py).  x (K,h,w), edge (1,h,w) torch or numpy -> numpy float64 (K,1,h,w)."""
    import scipy.sparse as sp

    x = torch.as_tensor(np.asarray(x)).to(torch.float64)
    edge = torch.as_tensor(np.asarray(edge)).to(torch.float64)
    height, width = x.shape[-2:]
    hor_padded, vert_padded = width + radius * 2, height + radius
    pi = PathIndex(radius=radius, default_size=(vert_padded, hor_padded))
    edge_padded = F.pad(edge, (radius, radius, 0, radius), mode="constant", value=1.0)
    aff = edge_to_affinity(torch.unsqueeze(edge_padded, 0), pi.path_indices).reshape(-1).numpy()
    n = vert_padded * hor_padded
    ind_to = np.asarray(pi.dst_indices).reshape(-1)
    ind_from = np.tile(np.asarray(pi.src_indices), np.asarray(pi.dst_indices).shape[0])
    assert ind_from.shape == ind_to.shape == aff.shape
    dense = sp.coo_matrix((aff, (ind_from, ind_to)), shape=(n, n)).tocsr()
    dense = dense + dense.T + sp.identity(n, format="csr")
    keep = np.arange(n).reshape(vert_padded, hor_padded)[:-radius, radius:-radius].reshape(-1)  # dense[:-r, r:-r, :-r, r:-r]
    dense = dense[keep][:, keep].tocsc()
    dense.data = dense.data ** beta
    col = np.asarray(dense.sum(axis=0)).reshape(-1)
    trans = (dense @ sp.diags(1.0 / col)).tocsr()
    v = (x.reshape(-1, height, width) * (1 - edge)).reshape(-1, height * width).numpy()
    tt = trans.T.tocsr()
    for _ in range(2 ** exp_times):
        v = (tt @ v.T).T
    return v.reshape(v.shape[0], 1, height, width)
