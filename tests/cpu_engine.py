"""TEST-ONLY numpy engine with the interface ShardedCholesky expects from an engine (set_values, factorize_phase,
top_tensor, num_segments / segment_tensors / factorize_segment, get_factor).  A plain right-looking supernodal
Cholesky with the same phases and the same compact storage (owned panels, then the top panels, contiguous) as the HIP
plan, so that the multi-rank orchestration (partition, sum all-reduces, replicated or distributed top) can be exercised
under gloo without a GPU.  Distributed top: one segment per top supernode (reduce its panel, factor it on every
rank, then apply only the columns k = rank (mod world) of its Schur updates -- a split of the additive update)."""
import numpy as np


class NumpyEngine:
    def __init__(self, sym, phase, load_top, rank=0, world=1, distributed=False):
        self.S, self.phase, self.load_top = sym, np.asarray(phase), bool(load_top)
        self.rank, self.world, self.distributed = rank, world, bool(distributed and world > 1)
        self.top_list = [int(s) for s in np.flatnonzero(self.phase == 1)]
        S = sym
        self.ncol = np.diff(S.Super)
        self.nrow = np.diff(S.Lsip)
        self.off = np.full(S.nsuper, -1, dtype=np.int64)
        run = 0
        for ph in (0, 1):
            if ph == 1:
                self.top_off = run
            for s in range(S.nsuper):
                if self.phase[s] == ph:
                    self.off[s] = run
                    run += int(self.ncol[s] * self.nrow[s])
        self.buf = np.zeros(max(run, 1))
        self.total = run
        self.Lx = None

    def panel(self, s):
        o = self.off[s]
        return self.buf[o:o + self.ncol[s] * self.nrow[s]].reshape(self.ncol[s], self.nrow[s]).T   # nsrow x nscol view

    def set_values(self, Lx):
        self.Lx = np.asarray(Lx, dtype=np.float64)

    def _assemble(self):
        S = self.S
        self.buf[:] = 0
        for s in range(S.nsuper):
            if self.phase[s] == 0 or (self.phase[s] == 1 and self.load_top):
                rows = S.Lsi[S.Lsip[s]:S.Lsip[s + 1]]
                pos = {int(g): k for k, g in enumerate(rows)}
                A = self.panel(s)
                for j in range(S.Super[s], S.Super[s + 1]):
                    for p in range(S.Lp[j], S.Lp[j + 1]):
                        A[pos[int(S.Li[p])], j - S.Super[s]] = self.Lx[p]

    def num_segments(self):
        return len(self.top_list) if self.distributed else 0

    def segment_tensors(self, k):
        import torch
        s = self.top_list[k]
        o = self.off[s]
        return [torch.from_numpy(self.buf[o:o + self.ncol[s] * self.nrow[s]])]

    def factorize_segment(self, k):
        self._factor_supernodes([self.top_list[k]], split=True)

    def finish(self):
        pass

    def factorize_phase(self, which):
        S = self.S
        if which == 0:
            self._assemble()
        assert not (self.distributed and which == 1)
        self._factor_supernodes([s for s in range(S.nsuper) if self.phase[s] == which], split=False)

    def _factor_supernodes(self, which, split):
        S = self.S
        for s in which:
            n, r = int(self.ncol[s]), int(self.nrow[s])
            A = self.panel(s)
            L11 = np.linalg.cholesky(np.tril(A[:n, :n]) + np.tril(A[:n, :n], -1).T)
            A[:n, :n] = L11
            if r > n:
                A[n:, :] = np.linalg.solve(L11, A[n:, :].T).T
            rows = S.Lsi[S.Lsip[s]:S.Lsip[s + 1]]
            i = n
            while i < r:
                a = int(S.SuperMap[rows[i]])
                e = i
                while e < r and S.SuperMap[rows[e]] == a:
                    e += 1
                cols = slice(self.rank, None, self.world) if split else slice(None)
                C = A[i:, cols] @ A[i:e, cols].T                 # (r-i) x (e-i)
                arows = S.Lsi[S.Lsip[a]:S.Lsip[a + 1]]
                rm = np.searchsorted(arows, rows[i:])
                T = self.panel(a)
                for cj in range(e - i):
                    T[rm[cj:], rm[cj]] -= C[cj:, cj]             # lower trapezoid only
                i = e

    def top_tensor(self):
        import torch
        return torch.from_numpy(self.buf[self.top_off:self.total])

    def get_factor(self, out=None):
        S = self.S
        if out is None:
            out = np.zeros(S.xsize)
        for s in range(S.nsuper):
            if self.off[s] >= 0:
                out[S.Lsxp[s]:S.Lsxp[s + 1]] = self.buf[self.off[s]:self.off[s] + self.ncol[s] * self.nrow[s]]
        return out

    def close(self):
        pass
