"""Synthetic input matrices of BASELINE.json's configs, as the lower triangle in CSC (what a
MatrixMarket 'symmetric' file holds and what SparseFrame_compress produces, SparseFrame.c:526-587).

All generators are vectorised numpy and deterministic.  Node numbering of a grid: id = x + nx*(y + ny*z).
"""
import numpy as np


def _csc_from_coo(n, rows, cols, vals):
    """column-major compress; entry order inside a column = input order (stable), like the
    reference's counting sort (SparseFrame.c:560-576)."""
    order = np.argsort(cols, kind="stable")
    rows, cols, vals = rows[order], cols[order], vals[order]
    Cp = np.zeros(n + 1, dtype=np.int64)
    Cp[1:] = np.cumsum(np.bincount(cols, minlength=n))          # (np.add.at is an order of magnitude slower at 10^8 entries)
    return Cp, rows.astype(np.int64), vals.astype(np.float64)


def laplacian_lower(nx, ny=1, nz=1, diag=None):
    """(2*dims)-point Laplacian on an nx*ny*nz grid (5-point in 2-D, 7-point in 3-D): lower triangle.
    diag defaults to 2*dims (4 in 2-D, 6 in 3-D), off-diagonals are -1."""
    n = nx * ny * nz
    dims = (nx > 1) + (ny > 1) + (nz > 1)
    if diag is None:
        diag = 2.0 * max(dims, 1)
    # built column by column without a sort (a stable sort of 6.7e7 entries is most of the 28 s the COO route took at 256^3):
    # column j holds its diagonal, then j+1, j+nx, j+nx*ny where they exist -- the order the stable counting sort of the
    # reference's compress step (SparseFrame.c:560-576) gives the same entries listed stencil arm by stencil arm
    idx = np.arange(n, dtype=np.int64)
    x = idx % nx
    y = (idx // nx) % ny
    m1 = x + 1 < nx
    m2 = y + 1 < ny
    m3 = idx < n - nx * ny if nz > 1 else np.zeros(n, dtype=bool)
    del x, y
    Cp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(1 + m1.astype(np.int64) + m2 + m3, out=Cp[1:])
    Ci = np.empty(Cp[-1], dtype=np.int64)
    Cx = np.full(Cp[-1], -1.0)
    pos = Cp[:-1].copy()
    Ci[pos] = idx
    Cx[pos] = float(diag)
    for mask, off in ((m1, 1), (m2, nx), (m3, nx * ny)):
        pos += 1
        Ci[pos[mask]] = idx[mask] + off
        pos[~mask] -= 1
    return n, Cp, Ci, Cx


def stencil_spd_lower(nx, ny, radius=2, seed=12345):
    """BASELINE config 3 stand-in: 2-D grid, 21-point stencil (|dx|,|dy| <= radius without
    the 4 far corners), off-diagonals U(-1,0), diagonal 1 + sum|offdiag| (strictly diagonally
    dominant => SPD)."""
    n = nx * ny
    rng = np.random.default_rng(seed)
    idx = np.arange(n, dtype=np.int64)
    x = idx % nx
    y = idx // nx
    rows, cols, vals = [], [], []
    for dy in range(0, radius + 1):
        for dx in range(-radius, radius + 1):
            if dy == 0 and dx <= 0:
                continue
            if abs(dx) == radius and abs(dy) == radius:
                continue
            mask = (x + dx >= 0) & (x + dx < nx) & (y + dy < ny)
            j = idx[mask]
            i = j + dx + dy * nx
            rows.append(i)
            cols.append(j)
            vals.append(-rng.random(j.size))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    dsum = np.ones(n)
    np.add.at(dsum, rows, np.abs(vals))
    np.add.at(dsum, cols, np.abs(vals))
    rows = np.concatenate([idx, rows])
    cols = np.concatenate([idx, cols])
    vals = np.concatenate([dsum, vals])
    Cp, Ci, Cx = _csc_from_coo(n, rows, cols, vals)
    return n, Cp, Ci, Cx


def random_spd_lower(n, nnz_per_col=4, seed=0, bandwidth=None):
    """small random SPD test matrix: random lower pattern (optionally banded), values U(-1,1),
    diagonal = 1 + sum|offdiag| over the row/column."""
    rng = np.random.default_rng(seed)
    cols = np.repeat(np.arange(n, dtype=np.int64), nnz_per_col)
    if bandwidth is None:
        rows = rng.integers(0, n, size=cols.size)
    else:
        rows = np.minimum(cols + rng.integers(1, bandwidth + 1, size=cols.size), n - 1)
    lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
    keep = lo != hi
    lo, hi = lo[keep], hi[keep]
    key = np.unique(hi * n + lo)
    hi, lo = key // n, key % n
    vals = rng.uniform(-1, 1, size=hi.size)
    dsum = np.ones(n)
    np.add.at(dsum, hi, np.abs(vals))
    np.add.at(dsum, lo, np.abs(vals))
    idx = np.arange(n, dtype=np.int64)
    rows = np.concatenate([idx, hi])
    cc = np.concatenate([idx, lo])
    vv = np.concatenate([dsum, vals])
    Cp, Ci, Cx = _csc_from_coo(n, rows, cc, vv)
    return n, Cp, Ci, Cx


def arrow_spd_lower(n, width=1):
    """arrow-head: dense last `width` rows + diagonal (a relaxed-amalgamation edge case)."""
    idx = np.arange(n, dtype=np.int64)
    rows, cols, vals = [idx], [idx], [np.full(n, float(n + 1))]
    for w in range(width):
        r = n - 1 - w
        j = idx[idx < r]
        rows.append(np.full(j.size, r, dtype=np.int64))
        cols.append(j)
        vals.append(np.full(j.size, -0.5 / (w + 1)))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    Cp, Ci, Cx = _csc_from_coo(n, rows, cols, vals)
    return n, Cp, Ci, Cx


def unsymmetric_stencil(nx, ny=1, nz=1, extra_per_row=1, seed=2024, reach=1, drop=0.0):
    """BASELINE config 5 generator: 3-D stencil of faces + edges within `reach` (19-point for
    reach=1), made structurally unsymmetric by `extra_per_row` random off-pattern entries per row and/or by
    dropping a fraction `drop` of the off-diagonal entries (each (i,j) independently of (j,i)); independent
    values U(-1,-0.5) for (i,j) and (j,i), diagonal 1 + sum of |row| + |column| so that NO-PIVOT LU is stable
    (the reference never pivots).  Returns the whole matrix in CSC (n, Cp, Ci, Cx).
    Note: random long-range extras destroy the grid separators (fill explodes under a geometric ordering);
    the bench workload therefore uses drop > 0 and extra_per_row = 0."""
    n = nx * ny * nz
    rng = np.random.default_rng(seed)
    idx = np.arange(n, dtype=np.int64)
    x = idx % nx
    y = (idx // nx) % ny
    z = idx // (nx * ny)
    rows, cols = [], []
    for dz in range(-reach, reach + 1):
        for dy in range(-reach, reach + 1):
            for dx in range(-reach, reach + 1):
                if (dx, dy, dz) == (0, 0, 0) or abs(dx) + abs(dy) + abs(dz) > 2 * reach:
                    continue
                if nz == 1 and dz or ny == 1 and dy:
                    continue
                m = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz >= 0) & (z + dz < nz)
                j = idx[m]
                rows.append(j + dx + dy * nx + dz * nx * ny)
                cols.append(j)
    if extra_per_row:
        r = np.repeat(idx, extra_per_row)
        c = rng.integers(0, n, size=r.size)
        keep = r != c
        rows.append(r[keep])
        cols.append(c[keep])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    key = np.unique(cols * n + rows)
    cols, rows = key // n, key % n
    if drop > 0:
        keep = rng.random(rows.size) >= drop
        rows, cols = rows[keep], cols[keep]
    vals = rng.uniform(-1.0, -0.5, size=rows.size)
    d = np.ones(n)
    np.add.at(d, rows, np.abs(vals))
    np.add.at(d, cols, np.abs(vals))
    rows = np.concatenate([rows, idx])
    cols = np.concatenate([cols, idx])
    vals = np.concatenate([vals, d])
    Cp, Ci, Cx = _csc_from_coo(n, rows, cols, vals)
    return n, Cp, Ci, Cx


def unsymmetric_general(nx, ny=1, nz=1, seed=7, reach=1, diag_scale=0.3):
    """the stencil pattern of `unsymmetric_stencil`, NOT diagonally dominant: off-diagonals U(-1, 1), diagonal
    diag_scale * U(-1, 1) * (number of entries in the row)^(1/2) -- a matrix that needs pivoting.  Whole matrix in CSC."""
    n, Cp, Ci, Cx = unsymmetric_stencil(nx, ny, nz, extra_per_row=0, seed=seed, reach=reach, drop=0.1)
    rng = np.random.default_rng(seed + 1)
    Cx = rng.uniform(-1.0, 1.0, size=Cx.size)
    cols = np.repeat(np.arange(n), np.diff(Cp))
    isd = Ci == cols
    cnt = np.bincount(Ci, minlength=n).astype(float)
    Cx[isd] = diag_scale * rng.uniform(-1.0, 1.0, size=int(isd.sum())) * np.sqrt(cnt[Ci[isd]])
    return n, Cp, Ci, Cx


def weaken_diagonal(n, Cp, Ci, Cx, fraction=0.2, factor=0.02, seed=77):
    """a copy of the whole-matrix CSC (n, Cp, Ci, Cx) in which a random `fraction` of the diagonal entries is multiplied by
    `factor`: those rows lose their diagonal dominance, a threshold-pivoting LU has to interchange rows there (BASELINE config 5
    'with partial pivoting' exercised for real; the matrix stays far from singular: the other entries are untouched)"""
    rng = np.random.default_rng(seed)
    Cx = np.array(Cx, dtype=np.float64, copy=True)
    cols = np.repeat(np.arange(n), np.diff(Cp))
    dpos = np.flatnonzero(np.asarray(Ci) == cols)
    pick = dpos[rng.random(dpos.size) < fraction]
    Cx[pick] *= factor
    return n, Cp, Ci, Cx


def dense_from_csc(n, Cp, Ci, Cx):
    A = np.zeros((n, n))
    for j in range(n):
        A[Ci[Cp[j]:Cp[j + 1]], j] = Cx[Cp[j]:Cp[j + 1]]
    return A


def dense_from_lower(n, Cp, Ci, Cx):
    """dense symmetric matrix from a stored triangle (small n only)."""
    A = np.zeros((n, n))
    for j in range(n):
        for p in range(Cp[j], Cp[j + 1]):
            A[Ci[p], j] = Cx[p]
            A[j, Ci[p]] = Cx[p]
    return A


def write_matrix_market(path, n, Cp, Ci, Cx, symmetric=True):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real " + ("symmetric" if symmetric else "general") + "\n")
        f.write("%d %d %d\n" % (n, n, len(Ci)))
        for j in range(n):
            for p in range(Cp[j], Cp[j + 1]):
                f.write("%d %d %.17g\n" % (Ci[p] + 1, j + 1, Cx[p]))
