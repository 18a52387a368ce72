// Host solve with the factor in the reference layout, on several threads.
//
// The reference's solve is a scalar sweep over Lsx on the host (Cholesky C:3036-3139, LU L:3592-3700); sf_host.cpp / sf_lu_host.cpp
// keep that sweep as it is for small systems.  The struct path normally solves on the factor that is still resident on the device
// (sf_handlers_solve_resident_sym); it comes here when there is none -- an out-of-core factorization (the factor only
// exists on the host, and two sweeps over host memory at a few hundred GB/s beat two trips over PCIe), several handlers without a
// gather, an evicted plan, an Lsx the caller changed.  At 128^3 the scalar sweep takes 3.8 s for the 30 GB factor; this one is
// bound by the host's memory bandwidth.
//
//   subtrees   sf_subtree_partition cuts the supernodal tree into T sets of subtrees + the supernodes above them ("top").  Every
//              thread sweeps its own subtrees; what they add to rows of top supernodes goes to a private array per thread and is
//              summed afterwards (forward), and the backward sweep of a subtree only reads rows that are final by then.
//   top        all threads work on one supernode at a time, 64 columns per step: thread 0 solves the 64 x 64 triangle (and applies
//              the LU row interchanges of the block), the rows below are cut into one contiguous chunk per thread.
//
// Results differ from the scalar sweep's in the last bits (the order of the additions into top rows); tests compare both.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <time.h>
#include <thread>
#include <vector>

namespace sf_host_solve {

typedef int64_t Long;

struct SpinBarrier {
    std::atomic<int> count{0}, sense{0};
    int n;
    explicit SpinBarrier(int n_) : n(n_) {}
    void wait() {
        const int s = sense.load(std::memory_order_relaxed);
        if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            count.store(0, std::memory_order_relaxed);
            sense.store(s ^ 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (sense.load(std::memory_order_acquire) == s)
                if (++spins > 2000) { std::this_thread::yield(); spins = 0; }
        }
    }
};

// U(c, r), r > c, of supernode panel P: Cholesky L(r, c); LU: U11 stored transposed in the diagonal block's upper part, U12^T below L21
template <bool LU>
static inline double upper(const double* P, Long lda, Long nsrow, Long nscol, Long c, Long r) {
    if (!LU) return P[c * lda + r];
    return r < nscol ? P[r * lda + c] : P[(nsrow - nscol) + c * lda + r];
}

// owner[s]: thread of the subtree holding supernode s, -1 = top (sf_subtree_partition).  PivInv: LU row interchanges or nullptr.
template <bool LU>
void solve_parallel(Long n, Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi, const Long* Lsxp,
                    const double* Lsx, const Long* PivInv, const int32_t* owner, int T, double* x) {
    // compact index of the columns of top supernodes
    std::vector<int32_t> topidx((size_t)std::max<Long>(n, 1), -1);
    std::vector<Long> top;                          // top supernodes, ascending
    Long ntop = 0;
    for (Long s = 0; s < nsuper; ++s)
        if (owner[s] < 0) {
            top.push_back(s);
            for (Long j = Super[s]; j < Super[s + 1]; ++j) topidx[(size_t)j] = (int32_t)ntop++;
        }
    std::vector<std::vector<double>> upd((size_t)T, std::vector<double>((size_t)std::max<Long>(ntop, 1), 0.0));
    std::vector<double> part((size_t)T * 64, 0.0);
    double xb[64];
    SpinBarrier bar(T);

    auto interchange = [&](Long s, Long c) {        // LU: the block's rows go to their pivot positions before the block's columns are used
        if (!LU || !PivInv) return;
        const Long nscol = Super[s + 1] - Super[s];
        const Long c0 = Super[s] + c, bw = std::min<Long>(64, nscol - c);
        bool moved = false;
        for (Long k = 0; k < bw; ++k) moved = moved || PivInv[c0 + k] != c0 + k;
        if (!moved) return;
        double tmp[64];
        for (Long k = 0; k < bw; ++k) tmp[PivInv[c0 + k] - c0] = x[c0 + k];
        for (Long k = 0; k < bw; ++k) x[c0 + k] = tmp[k];
    };

    Long max_nsrow = 1;
    for (Long s = 0; s < nsuper; ++s) max_nsrow = std::max(max_nsrow, Lsip[s + 1] - Lsip[s]);

    // Every supernode is handled as a dense block: the triangle on the (contiguous) entries of its own columns, the rows below as a
    // matrix-vector product into / out of a dense temporary, with ONE gather or scatter through the row indices per supernode (per
    // 64-column step in the top part) -- the column loops then stream the panel and vectorise.
    const bool trace = getenv("SF_TRACE") != nullptr;
    double stamps[6] = {0, 0, 0, 0, 0, 0};
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    auto worker = [&](int tid) {
        double* mine = upd[(size_t)tid].data();
        std::vector<double> tbuf((size_t)max_nsrow, 0.0);
        double* t = tbuf.data();
        if (tid == 0) stamps[0] = now();
        // ---- forward, own subtrees
        for (Long s = 0; s < nsuper; ++s) {
            if (owner[s] != tid) continue;
            const Long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s], lda = LU ? 2 * nsrow - nscol : nsrow;
            const Long* rows = Lsi + Lsip[s];
            const double* P = Lsx + Lsxp[s];
            double* y = x + Super[s];                       // the supernode's own columns: rows[c] = Super[s] + c
            const Long nb = nsrow - nscol;
            for (Long r = 0; r < nb; ++r) t[r] = 0.0;
            for (Long c = 0; c < nscol; ++c) {
                if (c % 64 == 0) interchange(s, c);
                const double* col = P + c * lda;
                const double xj = LU ? y[c] : (y[c] /= col[c]);
                for (Long r = c + 1; r < nscol; ++r) y[r] -= col[r] * xj;
                const double* cb = col + nscol;
                for (Long r = 0; r < nb; ++r) t[r] += cb[r] * xj;
            }
            for (Long r = 0; r < nb; ++r) {
                const Long g = rows[nscol + r];
                const int32_t ti = topidx[(size_t)g];
                if (ti < 0) x[g] -= t[r];
                else mine[ti] -= t[r];
            }
        }
        bar.wait();
        if (tid == 0) stamps[1] = now();
        // the subtrees' contributions to top rows (every thread a slice of the top columns)
        {
            const Long lo = ntop * tid / T, hi = ntop * (tid + 1) / T;
            for (Long s : top)
                for (Long j = Super[s]; j < Super[s + 1]; ++j) {
                    const Long ti = topidx[(size_t)j];
                    if (ti < lo || ti >= hi) continue;
                    double a = 0;
                    for (int q = 0; q < T; ++q) a += upd[(size_t)q][(size_t)ti];
                    x[j] += a;
                }
        }
        bar.wait();
        // ---- forward, top supernodes: all threads on one supernode
        for (Long s : top) {
            const Long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s], lda = LU ? 2 * nsrow - nscol : nsrow;
            const Long* rows = Lsi + Lsip[s];
            const double* P = Lsx + Lsxp[s];
            for (Long c0 = 0; c0 < nscol; c0 += 64) {
                const Long bw = std::min<Long>(64, nscol - c0), below = c0 + bw;
                if (tid == 0) {
                    interchange(s, c0);
                    for (Long c = c0; c < below; ++c) {
                        const double* col = P + c * lda;
                        const double xj = LU ? x[rows[c]] : (x[rows[c]] /= col[c]);
                        xb[c - c0] = xj;
                        for (Long r = c + 1; r < below; ++r) x[rows[r]] -= col[r] * xj;
                    }
                }
                bar.wait();
                const Long m = nsrow - below, lo = below + m * tid / T, hi = below + m * (tid + 1) / T, len = hi - lo;
                for (Long r = 0; r < len; ++r) t[r] = 0.0;
                for (Long c = c0; c < below; ++c) {
                    const double* col = P + c * lda + lo;
                    const double xj = xb[c - c0];
                    for (Long r = 0; r < len; ++r) t[r] += col[r] * xj;
                }
                for (Long r = 0; r < len; ++r) x[rows[lo + r]] -= t[r];
                bar.wait();
            }
        }
        if (tid == 0) stamps[2] = now();
        // ---- backward, top supernodes
        for (size_t k = top.size(); k-- > 0;) {
            const Long s = top[k];
            const Long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s], lda = LU ? 2 * nsrow - nscol : nsrow;
            const Long* rows = Lsi + Lsip[s];
            const double* P = Lsx + Lsxp[s];
            const Long nblk = (nscol + 63) / 64;
            for (Long b = nblk - 1; b >= 0; --b) {
                const Long c0 = b * 64, bw = std::min<Long>(64, nscol - c0), below = c0 + bw;
                const Long m = nsrow - below, lo = below + m * tid / T, hi = below + m * (tid + 1) / T, len = hi - lo;
                double* pp = part.data() + (size_t)tid * 64;
                for (Long r = 0; r < len; ++r) t[r] = x[rows[lo + r]];          // one gather for the 64 columns
                for (Long c = c0; c < below; ++c) pp[c - c0] = 0.0;
                if (!LU) {
                    for (Long c = c0; c < below; ++c) {
                        const double* col = P + c * lda + lo;
                        double a = 0;
                        for (Long r = 0; r < len; ++r) a += col[r] * t[r];
                        pp[c - c0] = a;
                    }
                } else {
                    // rows inside the diagonal block (r < nscol): U11(c, r) = P[r * lda + c], contiguous in c; below it: U12^T, contiguous in r
                    const Long mid = std::min(std::max(lo, nscol), hi);
                    for (Long r = lo; r < std::min(hi, nscol); ++r) {
                        const double xr = t[r - lo];
                        const double* row = P + r * lda;
                        for (Long c = c0; c < below; ++c) pp[c - c0] += row[c] * xr;
                    }
                    for (Long c = c0; c < below; ++c) {
                        const double* col = P + (nsrow - nscol) + c * lda;
                        double a = 0;
                        for (Long r = mid; r < hi; ++r) a += col[r] * t[r - lo];
                        pp[c - c0] += a;
                    }
                }
                bar.wait();
                if (tid == 0) {
                    for (Long c = below - 1; c >= c0; --c) {
                        double acc = x[rows[c]];
                        for (int q = 0; q < T; ++q) acc -= part[(size_t)q * 64 + (size_t)(c - c0)];
                        for (Long r = c + 1; r < below; ++r) acc -= upper<LU>(P, lda, nsrow, nscol, c, r) * x[rows[r]];
                        x[rows[c]] = acc / P[c * lda + c];
                    }
                }
                bar.wait();
            }
        }
        if (tid == 0) stamps[3] = now();
        // ---- backward, own subtrees (every row they read is final: ancestors in the subtree were done before, top rows above)
        for (Long s = nsuper - 1; s >= 0; --s) {
            if (owner[s] != tid) continue;
            const Long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s], lda = LU ? 2 * nsrow - nscol : nsrow;
            const Long* rows = Lsi + Lsip[s];
            const double* P = Lsx + Lsxp[s];
            double* y = x + Super[s];
            const Long nb = nsrow - nscol;
            for (Long r = 0; r < nb; ++r) t[r] = x[rows[nscol + r]];
            for (Long c = nscol - 1; c >= 0; --c) {
                double acc = y[c];
                if (!LU) {
                    const double* col = P + c * lda;
                    for (Long r = c + 1; r < nscol; ++r) acc -= col[r] * y[r];
                    const double* cb = col + nscol;
                    double a = 0;
                    for (Long r = 0; r < nb; ++r) a += cb[r] * t[r];
                    acc -= a;
                } else {
                    for (Long r = c + 1; r < nscol; ++r) acc -= P[r * lda + c] * y[r];
                    const double* cb = P + nsrow + c * lda;                 // U12^T of column c: rows [nscol, nsrow) at offset (nsrow - nscol) + r
                    double a = 0;
                    for (Long r = 0; r < nb; ++r) a += cb[r] * t[r];
                    acc -= a;
                }
                y[c] = acc / P[c * lda + c];
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(worker, t);
    worker(0);
    for (std::thread& t : th) t.join();
    if (trace)
        fprintf(stderr, "[sparseframe-hip] host solve, %d threads, %lld top supernodes (%lld columns): forward subtrees %.0f ms, top %.0f ms; backward top %.0f ms, "
                        "subtrees %.0f ms\n", T, (long long)top.size(), (long long)ntop, 1e3 * (stamps[1] - stamps[0]), 1e3 * (stamps[2] - stamps[1]),
                1e3 * (stamps[3] - stamps[2]), 1e3 * (now() - stamps[3]));
    (void)SuperMap;
}

// number of threads for a system of this size: 1 = keep the scalar sweep.  SF_HOST_SOLVE_THREADS (0 / 1: scalar), SF_HOST_SOLVE_MIN
// (factor entries below which the scalar sweep stays; default 2^24) -- read per call.
static inline int threads_for(Long xsize) {
    Long min_entries = (Long)1 << 24;
    if (const char* env = getenv("SF_HOST_SOLVE_MIN")) min_entries = atoll(env);
    if (xsize < min_entries) return 1;
    int T = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* env = getenv("SF_HOST_SOLVE_THREADS")) T = std::max(1, std::min(64, atoi(env)));
    return T;
}

}  // namespace sf_host_solve
