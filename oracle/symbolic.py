"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Pure-Python restatement ("oracle") of the reference's host-side symbolic analysis for Cholesky,
function by function, in the reference's own order and with its own tie-breaking:

    perm                <- SparseFrame_perm                Cholesky/Source/SparseFrame.c:956-1066
    etree               <- SparseFrame_etree               :1068-1127
    postorder           <- SparseFrame_postorder           :1129-1236
    colcount            <- SparseFrame_colcount            :1238-1352
    analyze_supernodal  <- SparseFrame_analyze_supernodal  :1354-1914
    analyze             <- SparseFrame_analyze             :1916-1978  (ordering supplied by the caller:
                           the reference calls third-party METIS_NodeND at :1937, unpinned)
    should_relax        <- Include/parameter.h:31-46

The LU variant (LU/Source/SparseFrame.c, "L:") is the same pipeline over the pattern of L + U^T with
(2*nsrow - nscol) x nscol panels; `analyze(..., lu=True, symmetric=...)` follows
    perm L:1068-1288, etree L:1290-1390, colcount L:1501-1645, analyze_supernodal L:1647-2231.

Plain loops over Python lists: use it for n up to a few 10^4.  The product's C++ analysis
(sparse-matrix-factorization-library_amd/csrc/sf_symbolic.cpp) must reproduce every integer array of
this file bit for bit; tests/test_symbolic_parity.py checks that.

Pins against the real reference: SURVEY.md Appendix C records (nfsuper, nsuper, nstage) of reference
runs made by the survey session -- 2-D 5-pt Laplacian 100x100, identity ordering, 1 GiB slot:
9900 / 155 / 1; 3-D 7-pt Laplacian 32^3 and 48^3 with geometric ND, 8 GiB slot: 21931 / 655 and
73773 / 2471.  tests/test_oracle_pins.py checks this file against those numbers.
"""

SIZEOF_FLOAT = 8
SIZEOF_LONG = 8


def should_relax(col, rate):
    cols = (16, 64, 256)
    rates = (0.8, 0.1, 0.05)
    for k in (2, 1, 0):
        if col > cols[k] and rate > rates[k]:
            return False
    return True


def perm_lu(n, Cp, Ci, Cx, Perm):
    """L:1068-1288, unsymmetric input: L by column (rows i >= j), U by ROW (columns j >= i), their
    transposes; the diagonal goes to both.  Returns (Lp, Li, Lx, LTp, LTi, LTx, Up, Ui, Ux, UTp, UTi, UTx)."""
    Pinv = [-1] * n
    for j in range(n):
        if Perm[j] >= 0:
            Pinv[Perm[j]] = j
    Lp = [0] * (n + 1)
    LTp = [0] * (n + 1)
    Up = [0] * (n + 1)
    UTp = [0] * (n + 1)
    for j in range(n):
        jold = Perm[j]
        if jold >= 0:
            for pold in range(Cp[jold], Cp[jold + 1]):
                i = Pinv[Ci[pold]]
                if j <= i:                                  # L:1114-1118
                    Lp[j + 1] += 1
                    LTp[i + 1] += 1
                if j >= i:                                  # L:1119-1123
                    Up[i + 1] += 1
                    UTp[j + 1] += 1
    for j in range(n):
        Lp[j + 1] += Lp[j]
        LTp[j + 1] += LTp[j]
        Up[j + 1] += Up[j]
        UTp[j + 1] += UTp[j]
    Li = [0] * Lp[n]
    Lx = [0.0] * Lp[n]
    LTi = [0] * Lp[n]
    LTx = [0.0] * Lp[n]
    Ui = [0] * Up[n]
    Ux = [0.0] * Up[n]
    UTi = [0] * Up[n]
    UTx = [0.0] * Up[n]
    Lw, LTw, Uw, UTw = Lp[:n], LTp[:n], Up[:n], UTp[:n]
    for j in range(n):
        jold = Perm[j]
        if jold >= 0:
            for pold in range(Cp[jold], Cp[jold + 1]):
                i = Pinv[Ci[pold]]
                if j <= i:                                  # L:1178-1196
                    lp = Lw[j]
                    Lw[j] += 1
                    Li[lp] = i
                    Lx[lp] = Cx[pold]
                    ltp = LTw[i]
                    LTw[i] += 1
                    LTi[ltp] = j
                    LTx[ltp] = Cx[pold]
                if j >= i:                                  # L:1198-1215
                    up = Uw[i]
                    Uw[i] += 1
                    Ui[up] = j
                    Ux[up] = Cx[pold]
                    utp = UTw[j]
                    UTw[j] += 1
                    UTi[utp] = i
                    UTx[utp] = Cx[pold]
    return Lp, Li, Lx, LTp, LTi, LTx, Up, Ui, Ux, UTp, UTi, UTx


def perm(n, Cp, Ci, Cx, Perm):
    """:956-1066 -- returns (Lp, Li, Lx, LTp, LTi, LTx)."""
    Pinv = [-1] * n
    for j in range(n):
        if Perm[j] >= 0:
            Pinv[Perm[j]] = j
    Lp = [0] * (n + 1)
    LTp = [0] * (n + 1)
    for j in range(n):
        jold = Perm[j]
        if jold >= 0:
            for pold in range(Cp[jold], Cp[jold + 1]):
                i = Pinv[Ci[pold]]
                Lp[min(i, j) + 1] += 1
                LTp[max(i, j) + 1] += 1
    for j in range(n):
        Lp[j + 1] += Lp[j]
        LTp[j + 1] += LTp[j]
    nz = Lp[n]
    Li = [0] * nz
    Lx = [0.0] * nz
    LTi = [0] * nz
    LTx = [0.0] * nz
    Lw = Lp[:n]
    LTw = LTp[:n]
    for j in range(n):
        jold = Perm[j]
        if jold >= 0:
            for pold in range(Cp[jold], Cp[jold + 1]):
                i = Pinv[Ci[pold]]
                lp = Lw[min(i, j)]
                Lw[min(i, j)] += 1
                Li[lp] = max(i, j)
                Lx[lp] = Cx[pold]
                ltp = LTw[max(i, j)]
                LTw[max(i, j)] += 1
                LTi[ltp] = min(i, j)
                LTx[ltp] = Cx[pold]
    return Lp, Li, Lx, LTp, LTi, LTx


def etree(n, LTp, LTi, UTp=None, UTi=None):
    """:1068-1127 ; LU also walks the rows of U^T (L:1358-1383)"""
    Parent = [-1] * n
    Ancestor = [-1] * n

    def walk(i, j):
        if i < j:
            while True:
                ancestor = Ancestor[i]
                if ancestor < 0:
                    Parent[i] = j
                    Ancestor[i] = j
                elif ancestor != j:
                    Ancestor[i] = j
                    i = ancestor
                else:
                    ancestor = -1
                if not ancestor >= 0:
                    break

    for j in range(n):
        for p in range(LTp[j], LTp[j + 1]):
            walk(LTi[p], j)
        if UTp is not None:
            for p in range(UTp[j], UTp[j + 1]):
                walk(UTi[p], j)
    return Parent


def postorder(n, Parent, ColCount=None):
    """:1129-1236"""
    Head = [-1] * n
    Next = [-1] * n
    if ColCount is None:
        for j in range(n - 1, -1, -1):
            p = Parent[j]
            if 0 <= p < n:
                Next[j] = Head[p]
                Head[p] = j
    else:
        # the reference's Whead has n usable buckets (weights 0..n-1, :1175-1199); a non-root column
        # of weight n would be lost there.  One extra bucket keeps it (documented deviation).
        Whead = [-1] * (n + 1)
        for j in range(n):
            if Parent[j] >= 0:
                w = ColCount[j]
                Next[j] = Whead[w]
                Whead[w] = j
        for w in range(n, -1, -1):
            j = Whead[w]
            while j >= 0:
                jnext = Next[j]
                p = Parent[j]
                Next[j] = Head[p]
                Head[p] = j
                j = jnext
    Stack = []
    for j in range(n - 1, -1, -1):
        if Parent[j] < 0:
            Stack.append(j)
    Post = [0] * n
    k = 0
    while Stack:
        j = Stack[-1]
        child = Head[j]
        if 0 <= child < n:
            Stack.append(child)
            Head[j] = Next[child]
        else:
            Stack.pop()
            Post[k] = j
            k += 1
    return Post


def colcount(n, Lp, Li, Post, Parent, Up=None, Ui=None):
    """:1238-1352 ; LU also visits row j of U (L:1601-1625)"""
    First = [-1] * n
    for k in range(n):
        p = Post[k]
        while p >= 0 and First[p] < 0:
            First[p] = k
            p = Parent[p]
    ColCount = [0] * n
    SetParent = list(range(n))
    PrevLeaf = list(range(n))
    PrevNbr = [-1] * n
    for k in range(n):
        j = Post[k]
        PrevNbr[j] = k
        cols = [Li[p] for p in range(Lp[j], Lp[j + 1])]
        if Up is not None:
            cols += [Ui[p] for p in range(Up[j], Up[j + 1])]
        for i in cols:
            if i > j:
                if First[j] > PrevNbr[i]:
                    prevleaf = PrevLeaf[i]
                    r = prevleaf
                    while r != SetParent[r]:
                        r = SetParent[r]
                    s = prevleaf
                    while s != r:           # the reference's loop exits after one step (:1323-1326)
                        SetParent[s] = r
                        s = SetParent[s]
                    ColCount[j] += 1
                    ColCount[r] -= 1
                    PrevLeaf[i] = j
                PrevNbr[i] = k
        SetParent[j] = Parent[j]
    for k in range(n):
        j = Post[k]
        p = Parent[j]
        if p >= 0:
            ColCount[p] += ColCount[j]
    for k in range(n):
        ColCount[Post[k]] += 1
    return ColCount


def analyze(n, Cp, Ci, Cx, Perm_in=None, devSlotSize=1 << 30, lu=False, symmetric=True):
    """:1916-1978 followed by :1354-1914 (lu=True: L:2233-2458 and L:1647-2231).  Returns a dict with every
    array the reference leaves in matrix_info (plus nfsuper and the pre-supernodal Parent0/Post/ColCount0)."""
    Cp = [int(v) for v in Cp]
    Ci = [int(v) for v in Ci]
    Cx = [float(v) for v in Cx]
    Perm = list(range(n)) if Perm_in is None else [int(v) for v in Perm_in]
    both = lu and not symmetric
    Up = Ui = Ux = UTp = UTi = UTx = None

    def panel_values(ncol, nrow):                                   # C:1641 / L:1946
        return ncol * (2 * nrow - ncol) if lu else ncol * nrow

    if both:
        Lp, Li, Lx, LTp, LTi, LTx, Up, Ui, Ux, UTp, UTi, UTx = perm_lu(n, Cp, Ci, Cx, Perm)
    else:
        Lp, Li, Lx, LTp, LTi, LTx = perm(n, Cp, Ci, Cx, Perm)      # :1953
    Parent = etree(n, LTp, LTi, UTp, UTi)                           # :1957
    Post = postorder(n, Parent, None)                               # :1961
    ColCount = colcount(n, Lp, Li, Post, Parent, Up, Ui)            # :1965
    Post = postorder(n, Parent, ColCount)                           # :1967
    out = {"Post": list(Post), "Parent0": list(Parent), "ColCount0": list(ColCount)}

    # ---- analyze_supernodal ----
    InvPost = [0] * n                                               # :1429-1447
    for k in range(n):
        InvPost[Post[k]] = k
    Bperm = [0] * n
    Bparent = [0] * n
    Bcolcount = [0] * n
    for k in range(n):
        parent = Parent[Post[k]]
        Bperm[k] = Perm[Post[k]]
        Bparent[k] = -1 if parent < 0 else InvPost[parent]
        Bcolcount[k] = ColCount[Post[k]]
    Perm, Parent, ColCount = Bperm, Bparent, Bcolcount
    if both:
        Lp, Li, Lx, LTp, LTi, LTx, Up, Ui, Ux, UTp, UTi, UTx = perm_lu(n, Cp, Ci, Cx, Perm)
    else:
        Lp, Li, Lx, LTp, LTi, LTx = perm(n, Cp, Ci, Cx, Perm)

    Nchild = [0] * n                                                # :1462-1469
    for j in range(n):
        parent = Parent[j]
        if 0 <= parent < n:
            Nchild[parent] += 1

    Super = [0] * (n + 1)                                           # :1471-1502
    nfsuper = 1 if n > 0 else 0
    Super[0] = 0
    for j in range(1, n):
        first = Super[nfsuper - 1]
        if (Parent[j - 1] != j or ColCount[j - 1] != ColCount[j] + 1 or Nchild[j] > 1) or \
           (panel_values(j - first + 1, ColCount[first]) * SIZEOF_FLOAT + ColCount[first] * SIZEOF_LONG > devSlotSize):
            Super[nfsuper] = j
            nfsuper += 1
    Super[nfsuper] = n

    Nscol = [0] * max(n, 1)
    Scolcount = [0] * max(n, 1)
    SuperMap = [0] * n
    Sparent = [0] * max(n, 1)
    for s in range(nfsuper):                                        # :1504-1522
        Nscol[s] = Super[s + 1] - Super[s]
        Scolcount[s] = ColCount[Super[s]]
    for s in range(nfsuper):
        for j in range(Super[s], Super[s + 1]):
            SuperMap[j] = s
    for s in range(nfsuper):
        parent = Parent[Super[s + 1] - 1]
        Sparent[s] = -1 if parent < 0 else SuperMap[parent]

    Merge = list(range(nfsuper))                                    # :1524-1591
    Nsz = [0] * nfsuper
    for s in range(nfsuper - 2, -1, -1):
        sparent = Sparent[s]
        if 0 <= sparent < nfsuper and Merge[s + 1] == Merge[sparent]:
            smerge = Merge[sparent]
            s_ncol, p_ncol = Nscol[s], Nscol[smerge]
            s_colcount, p_colcount = Scolcount[s], Scolcount[smerge]
            # C:1560 (s_ncol+p_ncol)*(s_ncol+p_colcount) ; L:1866 (s_ncol+p_ncol)*(s_ncol-p_ncol+2*p_colcount)
            if panel_values(s_ncol + p_ncol, s_ncol + p_colcount) * SIZEOF_FLOAT + (s_ncol + p_colcount) * SIZEOF_LONG <= devSlotSize:
                s_zero, p_zero = Nsz[s], Nsz[smerge]
                new_zero = s_ncol * (s_ncol + p_colcount - s_colcount)
                total_zero = s_zero + p_zero + new_zero
                tot = s_ncol + p_ncol
                denom = tot * (tot + 1) // 2 + tot * (p_colcount - p_ncol)
                if should_relax(tot, float(total_zero) / denom):
                    Nscol[smerge] = tot
                    Scolcount[smerge] = s_ncol + p_colcount
                    Nsz[smerge] = total_zero
                    Merge[s] = smerge

    nsuper = 0                                                      # :1593-1622
    Super[0] = 0
    for s in range(nfsuper):
        if Merge[s] == s:
            Super[nsuper + 1] = Super[s + 1]
            Nscol[nsuper] = Nscol[s]
            Scolcount[nsuper] = Scolcount[s]
            nsuper += 1
    Super[nsuper] = n
    for s in range(nsuper):
        for j in range(Super[s], Super[s + 1]):
            SuperMap[j] = s
    for s in range(nsuper):
        parent = Parent[Super[s + 1] - 1]
        Sparent[s] = -1 if parent < 0 else SuperMap[parent]

    Lsip = [0] * (nsuper + 1)                                       # :1632-1645
    Lsxp = [0] * (nsuper + 1)
    for s in range(nsuper):
        Lsip[s + 1] = Lsip[s] + Scolcount[s]
        Lsxp[s + 1] = Lsxp[s] + panel_values(Nscol[s], Scolcount[s])
    isize, xsize = Lsip[nsuper], Lsxp[nsuper]

    Lsi = [-1] * isize                                              # :1660-1692
    Lsip_copy = Lsip[:nsuper]
    Marker = [Super[s + 1] for s in range(nsuper)]
    for s in range(nsuper):
        for k in range(Super[s], Super[s + 1]):
            Lsi[Lsip_copy[s]] = k
            Lsip_copy[s] += 1
    for s in range(nsuper):
        for j in range(Super[s], Super[s + 1]):
            srcs = [LTi[p] for p in range(LTp[j], LTp[j + 1])]
            if both:                                                # L:1996-2007
                srcs += [UTi[p] for p in range(UTp[j], UTp[j + 1])]
            for i in srcs:
                sd = SuperMap[i]
                while sd >= 0 and Marker[sd] <= j:
                    Lsi[Lsip_copy[sd]] = j
                    Lsip_copy[sd] += 1
                    Marker[sd] = j + 1
                    sd = Sparent[sd]

    csize = 0                                                       # :1694-1719
    for s in range(nsuper):
        nscol = Super[s + 1] - Super[s]
        nsrow = Lsip[s + 1] - Lsip[s]
        if nscol < nsrow:
            si_last = nscol
            sparent_last = SuperMap[Lsi[Lsip[s] + nscol]]
            for si in range(nscol, nsrow):
                sparent = SuperMap[Lsi[Lsip[s] + si]]
                if sparent != sparent_last:
                    # C:1711 ; L:2028 (si - si_last) * (2*nsrow - si - si_last)
                    csize = max(csize, (si - si_last) * ((2 * nsrow - si - si_last) if lu else (nsrow - si_last)))
                    si_last = si
                    sparent_last = sparent
            csize = max(csize, (nsrow - si_last) * (nsrow - si_last))

    ST_Head = [-1] * max(nsuper, 1)                                 # :1721-1825
    ST_Next = [-1] * max(nsuper, 1)
    ST_Asize = [0] * max(nsuper, 1)
    ST_Msize = [0] * max(nsuper, 1)
    ST_Map = [-1] * nsuper
    nstage = 1 if nsuper > 0 else 0

    def fits(st, s):
        a = panel_values(Super[s + 1] - Super[s], Lsip[s + 1] - Lsip[s])
        m = Lsip[s + 1] - Lsip[s]
        return (ST_Asize[st] + a) * SIZEOF_FLOAT + (ST_Msize[st] + m) * SIZEOF_LONG <= devSlotSize

    def put(st, s):
        ST_Map[s] = st
        ST_Asize[st] += panel_values(Super[s + 1] - Super[s], Lsip[s + 1] - Lsip[s])
        ST_Msize[st] += Lsip[s + 1] - Lsip[s]

    for s in range(nsuper - 1, -1, -1):
        if Sparent[s] >= 0:
            st = ST_Map[Sparent[s]]
            if fits(st, s):
                put(st, s)
                continue
            st = ST_Head[ST_Map[Sparent[s]]]
        else:
            st = 0
        while st >= 0:
            if fits(st, s):
                put(st, s)
                break
            st = ST_Next[st]
        if st < 0:
            ST_Map[s] = nstage
            ST_Asize[nstage] = panel_values(Super[s + 1] - Super[s], Lsip[s + 1] - Lsip[s])
            ST_Msize[nstage] = Lsip[s + 1] - Lsip[s]
            if Sparent[s] >= 0:
                ST_Next[nstage] = ST_Head[ST_Map[Sparent[s]]]
                ST_Head[ST_Map[Sparent[s]]] = nstage
            else:
                ST_Next[nstage] = ST_Next[0]
                ST_Next[0] = nstage
            nstage += 1
    for s in range(nsuper):
        ST_Map[s] = nstage - 1 - ST_Map[s]

    ST_Pointer = [0] * (nstage + 1)                                 # :1827-1841
    ST_Index = [0] * nsuper
    for s in range(nsuper):
        ST_Pointer[ST_Map[s] + 1] += 1
    for st in range(nstage):
        ST_Pointer[st + 1] += ST_Pointer[st]
    w = ST_Pointer[:nstage]
    for s in range(nsuper):
        ST_Index[w[ST_Map[s]]] = s
        w[ST_Map[s]] += 1

    Nschild = [0] * max(nsuper, 1)                                  # :1848-1873
    for s in range(nsuper):
        nscol = Super[s + 1] - Super[s]
        nsrow = Lsip[s + 1] - Lsip[s]
        if nscol < nsrow:
            Nschild[SuperMap[Lsi[Lsip[s] + nscol]]] = 1
    LeafQueue = [-1] * nsuper
    nsleaf = 0
    for sp in range(nsuper):
        s = ST_Index[sp]
        if Nschild[s] == 0:
            LeafQueue[nsleaf] = s
            nsleaf += 1

    Aoffset = [0] * nsuper                                          # :1875-1904
    Moffset = [0] * nsuper
    for st in range(nstage):
        Asize = 0
        Msize = 0
        for pt in range(ST_Pointer[st], ST_Pointer[st + 1]):
            s = ST_Index[pt]
            nscol = Super[s + 1] - Super[s]
            nsrow = Lsip[s + 1] - Lsip[s]
            Aoffset[s] = Asize
            Asize += panel_values(nscol, nsrow) * SIZEOF_FLOAT
            Moffset[s] = Msize
            Msize += nsrow * SIZEOF_LONG
        for pt in range(ST_Pointer[st], ST_Pointer[st + 1]):
            Moffset[ST_Index[pt]] += Asize

    out.update(dict(
        n=n, nnz=Lp[n], Perm=Perm, Parent=Parent, ColCount=ColCount, lu=int(lu), symmetric=int(symmetric),
        Lp=Lp, Li=Li, Lx=Lx, LTp=LTp, LTi=LTi, LTx=LTx,
        Up=Up if both else [], Ui=Ui if both else [], Ux=Ux if both else [],
        UTp=UTp if both else [], UTi=UTi if both else [], UTx=UTx if both else [], unz=Up[n] if both else 0,
        nfsuper=nfsuper, nsuper=nsuper, Super=Super[:nsuper + 1], SuperMap=SuperMap, Sparent=Sparent[:nsuper],
        Lsip=Lsip, Lsxp=Lsxp, Lsi=Lsi, isize=isize, xsize=xsize, csize=csize,
        nstage=nstage, ST_Map=ST_Map, ST_Pointer=ST_Pointer, ST_Index=ST_Index,
        nsleaf=nsleaf, LeafQueue=LeafQueue, Aoffset=Aoffset, Moffset=Moffset))
    return out
