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


def _is_int(word):
    try:
        int(word)
        return True
    except ValueError:
        return False


def read_matrix(path):
    """-> dict(n, m, sym, nnz, numbering, ia, ja, a) with 0-based CSR arrays (int32 / float64 or complex128).

    Accepts what the reference's parsing constructor accepts (include/HPDDM_matrix.hpp:173-244): comment lines starting with
    ``#`` or ``%`` and blank lines anywhere; a header of one field (``n`` on one line, ``nnz`` on a later one: square, not
    symmetric), three fields (``n m nnz``) or four and more (``n m sym nnz [numbering]``); entries ``i j a`` or, when the first word
    of the first entry is not an integer, ``a i j`` (:218-222); complex values written ``(re,im)`` (:105-112).  Entries are 1-based."""
    n = m = nnz = 0
    sym, numbering = False, "C"
    entries = []
    with open(path) as fh:
        lines = iter(fh)
        for line in lines:                       # header: until nnz is known (:182-207)
            if not line.strip() or line[0] in "#%":
                continue
            head = line.split()
            if len(head) == 1:
                if n == 0:
                    n = m = int(head[0])
                else:
                    nnz = int(head[0])
            elif len(head) == 3:
                n, m, nnz = (int(v) for v in head)
            elif len(head) > 3:
                n, m, sym, nnz = int(head[0]), int(head[1]), bool(int(head[2])), int(head[3])
                if len(head) > 4:
                    numbering = head[4]
                    if numbering not in ("C", "F"):
                        raise ValueError(f"{path}: unknown numbering {numbering!r}")
            else:
                raise ValueError(f"{path}: malformed header {line!r}")
            if nnz:
                break
        for line in lines:
            if line.strip() and line[0] not in "#%":
                entries.append(line)
    if n <= 0 or m <= 0:
        raise ValueError(f"{path}: no header found")
    if len(entries) != nnz:
        raise ValueError(f"{path}: expected {nnz} entries, found {len(entries)}")
    index_first = _is_int(entries[0].split()[0]) if entries else True
    cplx = any("(" in e for e in entries[:1])
    rows = np.empty(nnz, dtype=np.int64)
    cols = np.empty(nnz, dtype=np.int64)
    vals = np.empty(nnz, dtype=np.complex128 if cplx else np.float64)
    for k, e in enumerate(entries):
        w = e.split()
        if len(w) != 3:
            raise ValueError(f"{path}: malformed entry {e!r}")
        iw, jw, vw = (w[0], w[1], w[2]) if index_first else (w[1], w[2], w[0])
        rows[k], cols[k] = int(iw) - 1, int(jw) - 1
        if cplx:
            re, im = vw.strip("()").split(",")
            vals[k] = complex(float(re), float(im))
        else:
            vals[k] = float(vw)
    if nnz and (rows.min() < 0 or rows.max() >= n or cols.min() < 0 or cols.max() >= m):
        raise ValueError(f"{path}: index out of range")
    if np.any(np.diff(rows) < 0):                # the reference trusts the file to be row-ordered; a stable sort is a superset
        order = np.argsort(rows, kind="stable")
        rows, cols, vals = rows[order], cols[order], vals[order]
    ia = np.zeros(n + 1, dtype=np.int32)
    np.add.at(ia, rows + 1, 1)
    np.cumsum(ia, out=ia)
    return {"n": n, "m": m, "sym": bool(sym), "nnz": nnz, "numbering": numbering, "ia": ia, "ja": cols.astype(np.int32),
            "a": np.ascontiguousarray(vals)}


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
