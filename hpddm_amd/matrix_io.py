"""Text format of the reference's matrix dumps (MatrixCSR::dump / the parsing constructor,
include/HPDDM_matrix.hpp:121-135 and :173-244; written by -hpddm_dump_matrices=<prefix> as <prefix>_<rank>_<size>.txt,
include/HPDDM_subdomain.hpp:379-386):

    # First line: n m (is symmetric) nnz indexing
    # For each nonzero coefficient: i j a_ij such that (i, j) \\in  {1, ..., n} x {1, ..., m}
    n m sym  nnz N
    <i> <j> <a_ij>        one line per stored entry, ALWAYS 1-based, rows ascending, %.44e

``N`` ('C' or 'F') is the numbering the matrix had in memory, not the numbering of the file.  With ``sym`` set only the
lower triangle is stored (diagonal last in its row).
"""
import numpy as np


def read_matrix(path):
    """-> dict(n, m, sym, nnz, numbering, ia, ja, a) with 0-based CSR arrays (int32 / float64)."""
    with open(path) as fh:
        line = fh.readline()
        while line.startswith("#"):
            line = fh.readline()
        head = line.split()
        if len(head) != 5:
            raise ValueError(f"{path}: malformed header {line!r}")
        n, m, sym, nnz = (int(v) for v in head[:4])
        numbering = head[4]
        if numbering not in ("C", "F"):
            raise ValueError(f"{path}: unknown numbering {numbering!r}")
        data = np.loadtxt(fh, dtype=np.float64, ndmin=2) if nnz else np.zeros((0, 3))
    if data.shape != (nnz, 3):
        raise ValueError(f"{path}: expected {nnz} entries, found {data.shape[0]}")
    rows = data[:, 0].astype(np.int64) - 1
    cols = data[:, 1].astype(np.int64) - 1
    if nnz and (rows.min() < 0 or rows.max() >= n or cols.min() < 0 or cols.max() >= m):
        raise ValueError(f"{path}: index out of range")
    if np.any(np.diff(rows) < 0):
        raise ValueError(f"{path}: rows are not in ascending order")
    ia = np.zeros(n + 1, dtype=np.int32)
    np.add.at(ia, rows + 1, 1)
    np.cumsum(ia, out=ia)
    return {"n": n, "m": m, "sym": bool(sym), "nnz": nnz, "numbering": numbering, "ia": ia, "ja": cols.astype(np.int32),
            "a": np.ascontiguousarray(data[:, 2])}


def write_matrix(path, n, ia, ja, a, sym=False, m=None, numbering="C"):
    """Write 0-based CSR arrays in the reference's format (byte-identical to MatrixCSR::dump for real scalars)."""
    m = n if m is None else m
    ia = np.asarray(ia)
    ja = np.asarray(ja)
    a = np.asarray(a, dtype=np.float64)
    with open(path, "w") as fh:
        fh.write("# First line: n m (is symmetric) nnz indexing\n")
        fh.write("# For each nonzero coefficient: i j a_ij such that (i, j) \\in  {1, ..., n} x {1, ..., m}\n")
        fh.write(f"{n} {m} {int(bool(sym))}  {int(ia[n] - ia[0])} {numbering}\n")
        for i in range(n):
            for k in range(int(ia[i] - ia[0]), int(ia[i + 1] - ia[0])):
                fh.write(f"{i + 1:9d} {int(ja[k]) + 1:9d} {a[k]:.44e}\n")


def csrmv(mat, x):
    """y = A x for a matrix returned by read_matrix (expands the symmetric storage), host side, for residual checks."""
    y = np.zeros(mat["n"])
    ia, ja, a = mat["ia"], mat["ja"], mat["a"]
    rows = np.repeat(np.arange(mat["n"]), np.diff(ia))
    np.add.at(y, rows, a * x[ja])
    if mat["sym"]:
        off = rows != ja
        np.add.at(y, ja[off], a[off] * x[rows[off]])
    return y
