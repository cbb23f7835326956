"""Problem generators for the RAS path (host logic, numpy only).

* :func:`generate2d` restates the reference generator ``examples/generate.cpp:43-311`` (same as ``generate.c`` /
  ``generate.py``): 2-D Poisson on [0,10]^2, cell-centred 5-point stencil, box partition with ``overlap`` layers,
  neighbour lists, shared-dof lists, partition-of-unity ramp and the analytic right-hand side -- including the
  reference's quirk that the vertical stencil offset is ``k +- Nx/xGrid`` (generate.cpp:202,219,233), which only
  equals the local row width when the subdomain does not own an overlap layer.
* :func:`generate3d` is the 3-D extension used by configs 2-3 of BASELINE.json (7-point Laplacian on N^3 cells of the
  unit cube, px x py x pz boxes grown by ``overlap`` layers, product-of-ramps partition of unity), for which the
  reference has no generator.

Every function returns one dict per subdomain with the arguments of ``HpddmSchwarzCreate`` (interface/HPDDM.h:101):
``n, ia, ja, a, sym, neighbors, connectivity, d`` (weights BEFORE multiplicityScaling), ``f`` and ``box``.
"""
import math

import numpy as np

PI = 3.141592653589793238463


def _rhs2d(xs, ys):
    """analytic right-hand side of examples/generate.cpp:71-86"""
    xsc, ysc, rsc, asc = (6.5, 2.0, 7.0), (8.0, 7.0, 3.0), (0.3, 0.3, 0.4), (0.3, 0.2, -0.1)
    frs = np.ones((len(ys), len(xs)))
    X, Y = np.meshgrid(xs, ys)
    for n in range(3):
        xd, yd = X - xsc[n], Y - ysc[n]
        inside = np.sqrt(xd * xd + yd * yd) <= rsc[n]
        frs = np.where(inside, frs - asc[n] * np.cos(0.5 * PI * xd / rsc[n]) * np.cos(0.5 * PI * yd / rsc[n]), frs)
    return frs.reshape(-1)


def generate2d(Nx, Ny, size, overlap=1, sym=False, numbering="C"):
    """All ``size`` subdomains of the reference's 2-D example (one per MPI rank there)."""
    F = 1 if numbering == "F" else 0
    xGrid = int(math.sqrt(size))
    while size % xGrid != 0:
        xGrid -= 1
    yGrid = size // xGrid
    dx, dy = 10.0 / Nx, 10.0 / Ny
    subs = []
    for rank in range(size):
        y, x = divmod(rank, xGrid)
        iStart, iEnd = max(x * Nx // xGrid - overlap, 0), min((x + 1) * Nx // xGrid + overlap, Nx)
        jStart, jEnd = max(y * Ny // yGrid - overlap, 0), min((y + 1) * Ny // yGrid + overlap, Ny)
        w, h = iEnd - iStart, jEnd - jStart
        ndof = w * h
        f = _rhs2d(dx * (np.arange(iStart, iEnd) + 0.5), dy * (np.arange(jStart, jEnd) + 0.5))
        d = np.ones(ndof)
        o, mapping = [], []
        ov = float(overlap)
        if jStart != 0:
            if iStart != 0:
                o.append(rank - xGrid - 1)
                mapping.append([i - iStart + w * j for j in range(2 * overlap) for i in range(iStart, iStart + 2 * overlap)])
                for j in range(overlap):
                    for i in range(overlap - j):
                        d[i + j + j * w] = j / ov
                    for i in range(j):
                        d[i + j * w] = i / ov
            else:
                for j in range(overlap):
                    for i in range(overlap):
                        d[i + j * w] = j / ov
            o.append(rank - xGrid)
            mapping.append([i - iStart + w * j for j in range(2 * overlap) for i in range(iStart, iEnd)])
            for j in range(overlap):
                for i in range(iStart + overlap, iEnd - overlap):
                    d[i - iStart + w * j] = j / ov
            if iEnd != Nx:
                o.append(rank - xGrid + 1)
                mapping.append([w * (i + 1) - 2 * overlap + j for i in range(2 * overlap) for j in range(2 * overlap)])
                for j in range(overlap):
                    for i in range(overlap - j):
                        d[w * (j + 1) - overlap + i] = j / ov
                    for i in range(j):
                        d[w * (j + 1) - i - 1] = i / ov
            else:
                for j in range(overlap):
                    for i in range(overlap):
                        d[w * (j + 1) - overlap + i] = j / ov
        if iStart != 0:
            o.append(rank - 1)
            mapping.append([j + (i - jStart) * w for i in range(jStart, jEnd) for j in range(2 * overlap)])
            for i in range(jStart + (jStart != 0) * overlap, jEnd - (jEnd != Ny) * overlap):
                for j in range(overlap):
                    d[j + (i - jStart) * w] = j / ov
        if iEnd != Nx:
            o.append(rank + 1)
            mapping.append([w * (i + 1 - jStart) - 2 * overlap + j for i in range(jStart, jEnd) for j in range(2 * overlap)])
            for i in range(jStart + (jStart != 0) * overlap, jEnd - (jEnd != Ny) * overlap):
                for j in range(overlap):
                    d[w * (i + 1 - jStart) - j - 1] = j / ov
        if jEnd != Ny:
            base = ndof - overlap * w
            if iStart != 0:
                o.append(rank + xGrid - 1)
                mapping.append([ndof - 2 * overlap * w + i - iStart + w * j for j in range(2 * overlap) for i in range(iStart, iStart + 2 * overlap)])
                for j in range(overlap):
                    for i in range(overlap - j):
                        d[base + i + w * j] = i / ov
                    for i in range(overlap - j, overlap):
                        d[base + i + w * j] = (overlap - 1 - j) / ov
            else:
                for j in range(overlap):
                    for i in range(overlap):
                        d[base + w * j + i] = (overlap - j - 1) / ov
            o.append(rank + xGrid)
            mapping.append([ndof - 2 * overlap * w + i - iStart + w * j for j in range(2 * overlap) for i in range(iStart, iEnd)])
            for j in range(overlap):
                for i in range(iStart + overlap, iEnd - overlap):
                    d[base + i - iStart + w * j] = (overlap - 1 - j) / ov
            if iEnd != Nx:
                o.append(rank + xGrid + 1)
                mapping.append([ndof - 2 * overlap * w + i - iStart + w * j + (w - 2 * overlap) for j in range(2 * overlap) for i in range(iStart, iStart + 2 * overlap)])
                for j in range(overlap):
                    for i in range(j, overlap):
                        d[base + i + w * (j + 1) - overlap] = (overlap - 1 - i) / ov
                    for i in range(j):
                        d[base + i + w * (j + 1) - overlap] = (overlap - 1 - j) / ov
            else:
                for j in range(overlap):
                    for i in range(overlap):
                        d[base + i + w * (j + 1) - overlap] = (overlap - j - 1) / ov
        # ---- matrix (examples/generate.cpp:188-241): note the +-Nx/xGrid vertical offset of the reference ----
        voff = Nx // xGrid
        ia, ja, a = [F], [], []
        k = 0
        for j in range(jStart, jEnd):
            for i in range(iStart, iEnd):
                if j > jStart:
                    a.append(-1 / (dy * dy)); ja.append(k - voff + F)
                if i > iStart:
                    a.append(-1 / (dx * dx)); ja.append(k - 1 + F)
                a.append(2 / (dx * dx) + 2 / (dy * dy)); ja.append(k + F)
                if not sym:
                    if i < iEnd - 1:
                        a.append(-1 / (dx * dx)); ja.append(k + 1 + F)
                    if j < jEnd - 1:
                        a.append(-1 / (dy * dy)); ja.append(k + voff + F)
                k += 1
                ia.append(len(a) + F)
        subs.append(dict(n=ndof, ia=np.array(ia, dtype=np.int32), ja=np.array(ja, dtype=np.int32), a=np.array(a), sym=bool(sym),
                         numbering=numbering, neighbors=np.array(o, dtype=np.int32), connectivity=[np.array(m, dtype=np.int32) for m in mapping],
                         d=d, f=f, box=(iStart, iEnd, jStart, jEnd)))
    return subs


def _factor3(p):
    """px*py*pz = p as cubic as possible"""
    best = None
    for a in range(1, p + 1):
        if p % a:
            continue
        for b in range(1, p // a + 1):
            if (p // a) % b:
                continue
            c = p // a // b
            key = max(a, b, c) - min(a, b, c)
            if best is None or key < best[0]:
                best = (key, (a, b, c))
    return best[1]


def _numbering(P, brick):
    """Subdomain numbering on a px x py x pz grid of boxes: (coords(r), index(cx, cy, cz)).  Without ``brick`` it is the
    lexicographic order (x fastest).  With ``brick = (bx, by, bz)`` the boxes are numbered brick by brick (bricks in lexicographic
    order, boxes inside a brick too), so that the contiguous ranges of HpddmHipSchwarzSetPartition are bx x by x bz bricks: the
    box-contiguous mapping of SURVEY 8(e) -- configs[3] is 4 x 4 x 4 boxes in 2 x 2 x 2 bricks, one brick per GPU, every GPU a
    neighbour of the 7 others."""
    px, py, pz = P
    if brick is None or tuple(brick) == (px, py, pz):
        def coords(r):
            z, rem = divmod(r, px * py)
            y, x = divmod(rem, px)
            return (x, y, z)

        def index(cx, cy, cz):
            return cx + px * (cy + py * cz)
        return coords, index
    bx, by, bz = brick
    assert px % bx == 0 and py % by == 0 and pz % bz == 0, "the brick must tile the grid of boxes"
    gx, gy = px // bx, py // by
    per = bx * by * bz

    def coords(r):
        b, w = divmod(r, per)
        Bz, rem = divmod(b, gx * gy)
        By, Bx = divmod(rem, gx)
        wz, rem = divmod(w, bx * by)
        wy, wx = divmod(rem, bx)
        return (Bx * bx + wx, By * by + wy, Bz * bz + wz)

    def index(cx, cy, cz):
        Bx, wx = divmod(cx, bx)
        By, wy = divmod(cy, by)
        Bz, wz = divmod(cz, bz)
        return (Bx + gx * (By + gy * Bz)) * per + wx + bx * (wy + by * wz)
    return coords, index


def gpu_grid(ngpus, per_gpu=(1, 1, 1)):
    """the grid of 2 x 2 x 2 bricks (one per GPU) bench.py and the tests use for ``ngpus`` GPUs: the factorisation that makes the
    global domain ``per_gpu * grid`` as cubic as possible, ties broken towards z (1: 1x1x1, 2: 1x1x2, 4: 1x2x2, 8: 2x2x2 for a
    cubic share; a share of 64 x 64 x 128 cells on 4 GPUs gives 2x2x1 = the 128^3 cube of configs[4])"""
    best = None
    for a in range(1, ngpus + 1):
        if ngpus % a:
            continue
        for b in range(1, ngpus // a + 1):
            if (ngpus // a) % b:
                continue
            g = (a, b, ngpus // a // b)
            dims = [per_gpu[k] * g[k] for k in range(3)]
            key = (max(dims) / min(dims), g[0], g[1])
            if best is None or key < best[0]:
                best = (key, g)
    return best[1]


def generate3d(N, parts, overlap=1, sym=True, numbering="C", rhs="ones", first=0, count=None, grid=None, normalize=False, neumann=False, brick=None):
    """7-point Laplacian on N^3 cells of the unit cube (h = 1/N, homogeneous Dirichlet through the stencil, like the
    2-D reference problem), split into ``parts`` boxes grown by ``overlap`` layers.  Returns the subdomains
    ``first .. first+count-1`` (default: all).  ``d`` is the product of the 1-D ramps of the reference generator
    (0, 1/overlap, ..., on the layers owned by a neighbour) -- multiplicityScaling turns it into a partition of unity.
    ``sym`` selects HPDDM's symmetric storage (lower triangle, diagonal last in row).  ``normalize`` returns in ``d``
    the final partition of unity w_i / sum_j w_j instead of the raw weights (what multiplicityScaling computes; needed
    when the neighbours of a subdomain live on another GPU)."""
    F = 1 if numbering == "F" else 0
    px, py, pz = grid if grid is not None else _factor3(parts)
    assert px * py * pz == parts
    dims, P = (N, N, N) if np.isscalar(N) else tuple(N), (px, py, pz)
    h2 = [float(dims[a]) ** 2 for a in range(3)]  # 1/h^2 per direction on the unit cube
    count = parts - first if count is None else count

    coords, index = _numbering(P, brick)

    boxes = {}
    for r in range(parts):
        c = coords(r)
        boxes[r] = [(max(c[a] * dims[a] // P[a] - overlap, 0), min((c[a] + 1) * dims[a] // P[a] + overlap, dims[a])) for a in range(3)]
    def weights(r):
        ramps = []
        for a, (s, e) in enumerate(boxes[r]):
            t = np.ones(e - s)
            if overlap > 0:
                if s != 0:  # a neighbour owns the first layers: 0, 1/ov, ...
                    t[:overlap] = np.arange(overlap) / float(overlap)
                if e != dims[a]:
                    t[-overlap:] = np.arange(overlap)[::-1] / float(overlap)
            ramps.append(t)
        return ramps[2][:, None, None] * ramps[1][None, :, None] * ramps[0][None, None, :]

    wsum = None
    if normalize:
        wsum = np.zeros((dims[2], dims[1], dims[0]))
        for r in range(parts):
            (a0, a1), (b0, b1), (c0, c1) = boxes[r]
            wsum[c0:c1, b0:b1, a0:a1] += weights(r)
    subs = []
    for r in range(first, first + count):
        c = coords(r)
        (i0, i1), (j0, j1), (k0, k1) = boxes[r]
        nx, ny, nz = i1 - i0, j1 - j0, k1 - k0
        n = nx * ny * nz
        idx = np.arange(n, dtype=np.int64).reshape(nz, ny, nx)
        # ---- matrix ----
        rows, cols, vals = [], [], []
        diag = 2.0 * (h2[0] + h2[1] + h2[2])
        rows.append(idx.ravel()); cols.append(idx.ravel()); vals.append(np.full(n, diag))
        for axis, hh in ((2, h2[0]), (1, h2[1]), (0, h2[2])):
            lo = np.take(idx, np.arange(0, idx.shape[axis] - 1), axis=axis).ravel()
            hi = np.take(idx, np.arange(1, idx.shape[axis]), axis=axis).ravel()
            rows.append(hi); cols.append(lo); vals.append(np.full(lo.size, -hh))      # lower triangle
            if not sym:
                rows.append(lo); cols.append(hi); vals.append(np.full(lo.size, -hh))
        rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
        order = np.lexsort((cols, rows))  # row-major, ascending columns => symmetric storage has the diagonal last in row
        rows, cols, vals = rows[order], cols[order], vals[order]
        ia = np.zeros(n + 1, dtype=np.int64)
        np.add.at(ia, rows + 1, 1)
        ia = np.cumsum(ia)
        a_neumann = None
        if neumann:
            # Neumann matrix of the subdomain (MatNeumann of examples/generate.cpp:243-291 in spirit): same pattern, no
            # coupling through the artificial interfaces -- the diagonal loses 1/h^2 per missing in-domain neighbour
            miss = np.zeros((nz, ny, nx))
            for axis, (lo, hi), hh, dim in ((2, (i0, i1), h2[0], dims[0]), (1, (j0, j1), h2[1], dims[1]), (0, (k0, k1), h2[2], dims[2])):
                sl_lo = [slice(None)] * 3
                sl_hi = [slice(None)] * 3
                sl_lo[axis], sl_hi[axis] = 0, -1
                if lo != 0:
                    miss[tuple(sl_lo)] += hh
                if hi != dim:
                    miss[tuple(sl_hi)] += hh
            a_neumann = vals.astype(np.float64).copy()
            a_neumann[rows == cols] -= miss.reshape(-1)
        # ---- partition-of-unity weights: product of the 1-D ramps ----
        d = weights(r)
        if normalize:
            d = d / wsum[k0:k1, j0:j1, i0:i1]
        d = d.reshape(-1)
        # ---- neighbours and shared dofs (intersection of the grown boxes, lexicographic order on both sides) ----
        neigh, conn = [], []
        for dz in (-1, 0, 1):
            for dy_ in (-1, 0, 1):
                for dx_ in (-1, 0, 1):
                    if dx_ == dy_ == dz == 0:
                        continue
                    cx, cy, cz = c[0] + dx_, c[1] + dy_, c[2] + dz
                    if not (0 <= cx < px and 0 <= cy < py and 0 <= cz < pz):
                        continue
                    q = index(cx, cy, cz)
                    inter = [(max(boxes[r][a][0], boxes[q][a][0]), min(boxes[r][a][1], boxes[q][a][1])) for a in range(3)]
                    if any(lo >= hi for lo, hi in inter):
                        continue
                    sl = idx[inter[2][0] - k0:inter[2][1] - k0, inter[1][0] - j0:inter[1][1] - j0, inter[0][0] - i0:inter[0][1] - i0]
                    neigh.append(q)
                    conn.append(sl.reshape(-1).astype(np.int32))
        order = sorted(range(len(neigh)), key=neigh.__getitem__)   # increasing neighbour number (brick numberings are not lexicographic)
        neigh, conn = [neigh[k] for k in order], [conn[k] for k in order]
        if rhs == "ones":
            f = np.ones(n)
        else:  # smooth analytic right-hand side
            X = (np.arange(i0, i1) + 0.5) / dims[0]
            Y = (np.arange(j0, j1) + 0.5) / dims[1]
            Z = (np.arange(k0, k1) + 0.5) / dims[2]
            f = (1.0 + np.sin(PI * Z)[:, None, None] * np.sin(2 * PI * Y)[None, :, None] * np.cos(PI * X)[None, None, :]).reshape(-1)
        subs.append(dict(n=n, ia=(ia + F).astype(np.int32), ja=(cols + F).astype(np.int32), a=vals.astype(np.float64), sym=bool(sym),
                         numbering=numbering, neighbors=np.array(neigh, dtype=np.int32), connectivity=conn, d=d, f=f,
                         box=(i0, i1, j0, j1, k0, k1)))
        if neumann:
            subs[-1]["a_neumann"] = a_neumann
    return subs


def _q1_elasticity_stiffness(h, E, nu):
    """24 x 24 stiffness matrix of a trilinear hexahedron of edge h (isotropic Hooke law, 2x2x2 Gauss points); local
    node a = (ax, ay, az) in {0,1}^3 numbered ax + 2*ay + 4*az, dof 3*a + component."""
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    C = np.zeros((6, 6))
    C[:3, :3] = lam
    C[np.arange(3), np.arange(3)] += 2 * mu
    C[np.arange(3, 6), np.arange(3, 6)] = mu
    gp = np.array([-1.0, 1.0]) / np.sqrt(3.0)
    corners = np.array([[(a & 1), (a >> 1) & 1, (a >> 2) & 1] for a in range(8)]) * 2.0 - 1.0
    Ke = np.zeros((24, 24))
    for gx in gp:
        for gy in gp:
            for gz in gp:
                g = np.array([gx, gy, gz])
                # derivatives of N_a = prod (1 + s_ad xi_d) / 8 w.r.t. physical coordinates (x = h (xi + 1) / 2)
                dN = np.zeros((8, 3))
                for a in range(8):
                    for dd in range(3):
                        t = corners[a, dd] / 8.0
                        for o in range(3):
                            if o != dd:
                                t *= 1 + corners[a, o] * g[o]
                        dN[a, dd] = t * 2.0 / h
                B = np.zeros((6, 24))
                for a in range(8):
                    bx, by, bz = dN[a]
                    B[:, 3 * a:3 * a + 3] = [[bx, 0, 0], [0, by, 0], [0, 0, bz], [by, bx, 0], [0, bz, by], [bz, 0, bx]]
                Ke += B.T @ C @ B * (h / 2.0) ** 3
    return 0.5 * (Ke + Ke.T)


def generate_elasticity3d(N, parts, overlap=1, sym=True, first=0, count=None, grid=None, normalize=True, neumann=False, E=1.0, nu=0.3, brick=None):
    """Linear elasticity (3 dofs per node, trilinear hexahedra, h = 1/N) on N^3 nodes of a cube clamped on the face x = 0
    (a layer of elements between the clamped plane and the first nodes), the other faces free; node boxes grown by
    ``overlap`` layers -- the block-3 workload of BASELINE.json configs[3].  Same conventions as generate3d: subdomain
    matrices R_i A R_i^T (``a_neumann``: the unassembled local matrix, elements inside the box only), ``d`` the
    partition of unity on the nodes repeated on the 3 components, neighbours / shared dofs in lexicographic order."""
    import scipy.sparse as sp
    px, py, pz = grid if grid is not None else _factor3(parts)
    assert px * py * pz == parts
    dims, P = (N, N, N) if np.isscalar(N) else tuple(N), (px, py, pz)
    h = 1.0 / dims[0]
    Ke = _q1_elasticity_stiffness(h, E, nu)
    count = parts - first if count is None else count

    coords, index = _numbering(P, brick)

    boxes = {}
    for r in range(parts):
        c = coords(r)
        boxes[r] = [(max(c[a] * dims[a] // P[a] - overlap, 0), min((c[a] + 1) * dims[a] // P[a] + overlap, dims[a])) for a in range(3)]

    def weights(r):
        ramps = []
        for a, (s, e) in enumerate(boxes[r]):
            t = np.ones(e - s)
            if overlap > 0:
                if s != 0:
                    t[:overlap] = np.arange(overlap) / float(overlap)
                if e != dims[a]:
                    t[-overlap:] = np.arange(overlap)[::-1] / float(overlap)
            ramps.append(t)
        return ramps[2][:, None, None] * ramps[1][None, :, None] * ramps[0][None, None, :]

    wsum = None
    if normalize:
        wsum = np.zeros((dims[2], dims[1], dims[0]))
        for r in range(parts):
            (a0, a1), (b0, b1), (c0, c1) = boxes[r]
            wsum[c0:c1, b0:b1, a0:a1] += weights(r)

    def assemble(box, inside_only):
        (i0, i1), (j0, j1), (k0, k1) = box
        nx, ny, nz = i1 - i0, j1 - j0, k1 - k0
        nn = nx * ny * nz
        if inside_only:  # elements whose free nodes all lie in the box (the clamped layer counts when the box touches x = 0)
            ex = np.arange(-1 if i0 == 0 else i0, i1 - 1)
            ey, ez = np.arange(j0, j1 - 1), np.arange(k0, k1 - 1)
        else:            # every element touching a node of the box
            ex = np.arange(max(i0 - 1, -1), min(i1, dims[0] - 1))
            ey, ez = np.arange(max(j0 - 1, 0), min(j1, dims[1] - 1)), np.arange(max(k0 - 1, 0), min(k1, dims[2] - 1))
        EZ, EY, EX = np.meshgrid(ez, ey, ex, indexing="ij")
        EX, EY, EZ = EX.ravel(), EY.ravel(), EZ.ravel()
        rows, cols, vals = [], [], []

        def local(a):
            x, y, z = EX + (a & 1), EY + ((a >> 1) & 1), EZ + ((a >> 2) & 1)
            ok = (x >= i0) & (x < i1) & (y >= j0) & (y < j1) & (z >= k0) & (z < k1)
            return ok, (x - i0) + nx * ((y - j0) + ny * (z - k0))

        loc = [local(a) for a in range(8)]
        for a in range(8):
            oka, na = loc[a]
            for b in range(8):
                okb, nb = loc[b]
                ok = oka & okb
                if not ok.any():
                    continue
                ra, cb = na[ok], nb[ok]
                for ca in range(3):
                    for cc in range(3):
                        # full 3 x 3 blocks (block-3 CSR): couplings that vanish on this regular grid stay as stored zeros
                        rows.append(3 * ra + ca)
                        cols.append(3 * cb + cc)
                        vals.append(np.full(ra.size, Ke[3 * a + ca, 3 * b + cc]))
        M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(3 * nn, 3 * nn)).tocsr()
        M.sum_duplicates()
        M.sort_indices()
        return M

    def to_storage(M):
        M = sp.tril(M, format="csr") if sym else M
        M.sort_indices()
        return M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)

    subs = []
    for r in range(first, first + count):
        c = coords(r)
        (i0, i1), (j0, j1), (k0, k1) = boxes[r]
        nx, ny, nz = i1 - i0, j1 - j0, k1 - k0
        nn = nx * ny * nz
        idx = np.arange(nn, dtype=np.int64).reshape(nz, ny, nx)
        A = assemble(boxes[r], False)
        ia, ja, a = to_storage(A)
        d = weights(r)
        if normalize:
            d = d / wsum[k0:k1, j0:j1, i0:i1]
        d = np.repeat(d.reshape(-1), 3)
        neigh, conn = [], []
        for dz in (-1, 0, 1):
            for dy_ in (-1, 0, 1):
                for dx_ in (-1, 0, 1):
                    if dx_ == dy_ == dz == 0:
                        continue
                    cx, cy, cz = c[0] + dx_, c[1] + dy_, c[2] + dz
                    if not (0 <= cx < px and 0 <= cy < py and 0 <= cz < pz):
                        continue
                    q = index(cx, cy, cz)
                    inter = [(max(boxes[r][t][0], boxes[q][t][0]), min(boxes[r][t][1], boxes[q][t][1])) for t in range(3)]
                    if any(lo >= hi for lo, hi in inter):
                        continue
                    nodes = idx[inter[2][0] - k0:inter[2][1] - k0, inter[1][0] - j0:inter[1][1] - j0, inter[0][0] - i0:inter[0][1] - i0].reshape(-1)
                    neigh.append(q)
                    conn.append((3 * nodes[:, None] + np.arange(3)[None, :]).reshape(-1).astype(np.int32))
        order = sorted(range(len(neigh)), key=neigh.__getitem__)
        neigh, conn = [neigh[k] for k in order], [conn[k] for k in order]
        f = np.zeros(3 * nn)
        f[2::3] = -h ** 3  # gravity along -z, lumped
        sub = dict(n=3 * nn, ia=ia, ja=ja, a=a, sym=bool(sym), numbering="C", neighbors=np.array(neigh, dtype=np.int32), connectivity=conn, d=d, f=f,
                   box=(i0, i1, j0, j1, k0, k1), block=3)
        if neumann:
            ian, jan, an = to_storage(assemble(boxes[r], True))
            sub["ia_neumann"], sub["ja_neumann"], sub["a_neumann"] = ian, jan, an
        subs.append(sub)
    return subs


def generate_helmholtz3d(dims, parts, wavenumber=2.0 * PI * 8.0, h=None, overlap=1, first=0, count=None, grid=None, brick=None, normalize=True, numbering="C"):
    """BASELINE.json configs[4]: ``-Laplace(u) - k^2 u`` with k = 2 pi 8 on ``dims`` cells of spacing ``h`` (default
    1 / max(dims): the unit cube for a cubic grid), FIRST-ORDER ABSORBING condition ``du/dn = i k u`` on the boundary of the domain --
    complex symmetric, indefinite, no volumetric damping.  Cell-centred 7-point stencil like :func:`generate3d`; the ghost value of a
    boundary face is eliminated through ``u_g = (1 + i k h) u_b``, i.e. the diagonal loses ``(1 + i k h) / h^2`` per boundary face.

    Per subdomain, beside the keys of :func:`generate3d` (``a`` complex, full storage, ``sym`` False):

    * ``a`` -- the restriction of the global matrix to the grown box (what GMV and the coarse operator use: truncated stencil on the
      artificial interfaces, absorbing term on the physical boundary);
    * ``a_opt`` -- the optimised local matrix of ORAS (``callNumfact(A_opt)``, include/HPDDM_schwarz.hpp:337-366): the same first-order
      impedance condition on the ARTIFICIAL interfaces of the box (diagonal minus ``(1 + i k h) / h^2`` per interface face);
    * ``a_neumann`` -- the local Neumann matrix of the DtN eigenproblem: homogeneous Neumann on the artificial interfaces (diagonal minus
      ``1 / h^2`` per interface face), absorbing term on the physical boundary;
    * ``b_dtn = (ia, ja, a)`` -- the right-hand side matrix of ``solveGEVP(A, B)`` (include/HPDDM_schwarz.hpp:665-666: the slot a DtN coarse
      space fills; the standalone reference has no DtN code): the lumped mass matrix of the artificial interface in the scaling of the
      finite-difference operator, ``faces / h`` on the cells that touch it, nothing elsewhere.  Eigenvalues of ``(a_neumann, b_dtn)``
      are then Dirichlet-to-Neumann eigenvalues in units of the wavenumber (the classical criterion keeps ``Re(lambda) < k``);
    * ``f`` is NOT set (configs[4] draws 8 random right-hand sides); ``wavenumber``, ``h``.
    """
    dims = (dims, dims, dims) if np.isscalar(dims) else tuple(dims)
    h = 1.0 / max(dims) if h is None else float(h)
    k = float(wavenumber)
    ih2 = 1.0 / (h * h)
    out = []
    for sd in generate3d(dims, parts, overlap=overlap, sym=False, numbering=numbering, rhs="ones", first=first, count=count, grid=grid, normalize=normalize, brick=brick):
        sd = dict(sd)
        F = 1 if numbering == "F" else 0
        n = sd["n"]
        i0, i1, j0, j1, k0, k1 = sd["box"]
        nx, ny, nz = i1 - i0, j1 - j0, k1 - k0
        phys = np.zeros((nz, ny, nx))    # faces of a cell on the physical boundary
        artf = np.zeros((nz, ny, nx))    # faces of a cell on an artificial interface of the grown box
        for axis, (lo, hi), dim in ((2, (i0, i1), dims[0]), (1, (j0, j1), dims[1]), (0, (k0, k1), dims[2])):
            sl_lo, sl_hi = [slice(None)] * 3, [slice(None)] * 3
            sl_lo[axis], sl_hi[axis] = 0, -1
            (phys if lo == 0 else artf)[tuple(sl_lo)] += 1.0
            (phys if hi == dim else artf)[tuple(sl_hi)] += 1.0
        phys, artf = phys.reshape(-1), artf.reshape(-1)
        rows = np.repeat(np.arange(n), np.diff(sd["ia"]))
        diag = rows == (sd["ja"] - F)
        a = np.full(sd["ja"].size, -ih2, dtype=np.complex128)
        absorb = (1.0 + 1j * k * h) * ih2
        a[diag] = 6.0 * ih2 - k * k - absorb * phys
        a_opt, a_neu = a.copy(), a.copy()
        a_opt[diag] -= absorb * artf
        a_neu[diag] -= ih2 * artf
        on = np.flatnonzero(artf > 0)
        bia = np.zeros(n + 1, dtype=np.int64)
        bia[on + 1] = 1
        sd.update(a=a, a_opt=a_opt, a_neumann=a_neu, sym=False, wavenumber=k, h=h,
                  b_dtn=((np.cumsum(bia) + F).astype(np.int32), (on + F).astype(np.int32), (artf[on] / h).astype(np.complex128)))
        sd.pop("f", None)
        out.append(sd)
    return out
