"""Algebraic overlapping decomposition of a global sparse matrix: what the reference's examples/generateFromFile.cpp:47-139 does
before it calls the solver (there with METIS for the initial partition; METIS is not in this image, so the partition is either given
by the caller or taken as equal chunks of a reverse Cuthill-McKee ordering -- contiguous strips of the graph).

Semantics restated from the reference: every subdomain is its own part plus `overlap` layers of graph neighbours (pattern of A);
the local matrix is the restriction R_p A R_p^T; unknowns are numbered by increasing global index, so that the shared lists of two
neighbours come out in the same order on both sides; the weights handed to multiplicityScaling are 1 on the own part, 1 - l / overlap
on layer l and 0 on the last layer (generateFromFile.cpp:114-118).
"""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee


def strip_partition(A, parts):
    """part[i] in [0, parts): equal chunks of the reverse Cuthill-McKee ordering of the pattern of A + A^T"""
    n = A.shape[0]
    pattern = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape)
    perm = reverse_cuthill_mckee((pattern + pattern.T).tocsr(), symmetric_mode=True)
    part = np.empty(n, dtype=np.int32)
    part[perm] = (np.arange(n) * parts) // n
    return part


def decompose(A, parts, overlap=1, part=None, rhs=None):
    """A: scipy sparse (n x n, general storage).  Returns the list of subdomain dicts the rest of the package works with (n, ia, ja, a,
    sym, numbering, neighbors, connectivity, d, f) plus "idx", the global numbers of the local unknowns (increasing)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    assert A.shape[0] == A.shape[1] and parts >= 1 and overlap >= 1
    part = strip_partition(A, parts) if part is None else np.asarray(part, dtype=np.int32)
    assert part.shape == (n,) and part.min() >= 0 and part.max() < parts
    pattern = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape)
    pattern = ((pattern + pattern.T) > 0).astype(np.float64).tocsr()
    # indicator[p, k]: 0 outside, 1 on the own part, 1 + l on layer l (generateFromFile.cpp:71-83)
    indicator = np.zeros((parts, n))
    indicator[part, np.arange(n)] = 1.0
    for layer in range(overlap):
        reached = (pattern @ (indicator > 0.5).T.astype(np.float64)).T > 0.5
        new = reached & ~(indicator > 0.5)
        indicator[new] = layer + 2.0
    f_global = np.ones(n) if rhs is None else np.asarray(rhs, dtype=np.float64)
    idx = [np.flatnonzero(indicator[p] > 0.0) for p in range(parts)]
    subs = []
    for p in range(parts):
        loc = A[idx[p]][:, idx[p]].tocsr()
        loc.sort_indices()
        ind = indicator[p, idx[p]]
        d = np.where(np.abs(ind - (1.0 + overlap)) < 0.5, 0.0, 1.0 - (ind - 1.0) / float(overlap))
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[idx[p]] = np.arange(idx[p].size)
        neighbors, connectivity = [], []
        for q in range(parts):
            if q == p:
                continue
            shared = np.intersect1d(idx[p], idx[q], assume_unique=True)
            if shared.size:
                neighbors.append(q)
                connectivity.append(g2l[shared].astype(np.int32))
        subs.append(dict(n=int(idx[p].size), ia=loc.indptr.astype(np.int32), ja=loc.indices.astype(np.int32), a=loc.data.astype(np.float64),
                         sym=False, numbering="C", neighbors=np.array(neighbors, dtype=np.int32), connectivity=connectivity, d=d,
                         f=f_global[idx[p]].copy(), idx=idx[p]))
    return subs


def gather(subs, xs, n):
    """global vector from the (consistent) local ones"""
    x = np.zeros((n,) + np.shape(xs[0])[1:])
    for sd, v in zip(subs, xs):
        x[sd["idx"]] = v
    return x
