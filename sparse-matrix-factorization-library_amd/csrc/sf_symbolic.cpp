// Host-side symbolic analysis for the supernodal Cholesky path.
//
// This is the caller side of the hot path: it produces every integer input of the numeric
// factorization.  The semantics (and therefore every integer output, bit for bit, for a given
// Perm and devSlotSize) follow the reference's SparseFrame_analyze pipeline:
//   perm            Cholesky/Source/SparseFrame.c:956-1066
//   etree           :1068-1127
//   postorder       :1129-1236   (plain, then weighted by ColCount :1967)
//   colcount        :1238-1352
//   analyze_supernodal :1354-1914 (fundamental supernodes, relaxed amalgamation,
//                    Lsi row structure, csize, stages, leaf queue, slot offsets)
// The code is organised differently (separate passes over std::vector, no shared workspace
// aliasing) but each pass is written to reproduce the reference's tie-breaking exactly.
#include <sched.h>
#include <mutex>
#include "sf_symbolic.h"
#include <atomic>
#include <chrono>
#include <thread>
#include <time.h>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>

namespace sf {

// For the duration of an analysis call the calling thread -- and with it every helper thread it starts: new threads inherit the
// mask -- stays on the CPUs of the NUMA node it is running on.  The passes walk n-entry arrays at random; on the two-socket hosts of
// the MI355X boxes half of those accesses were remote: the built-in ordering of the 128^3 matrix takes 0.34 s with the threads
// spread over both sockets and 0.26 s on one node (round 4, `taskset` experiment).  The caller's mask is restored on the way out;
// SF_ANALYZE_PIN=0 turns this off.  Nothing happens when the mask already lies inside one node or the topology cannot be read.
struct NodeAffinity {
    cpu_set_t saved;
    bool active = false;
    NodeAffinity() {
        if (const char* e = getenv("SF_ANALYZE_PIN")) if (atoi(e) == 0) return;
        if (sched_getaffinity(0, sizeof saved, &saved) != 0) return;
        const int cpu = sched_getcpu();
        if (cpu < 0) return;
        for (int node = 0; node < 64; ++node) {
            char path[96];
            snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
            FILE* f = fopen(path, "r");
            if (!f) { if (node == 0) return; break; }
            char buf[4096];
            const bool ok = fgets(buf, sizeof buf, f) != nullptr;
            fclose(f);
            if (!ok) continue;
            cpu_set_t mine;
            CPU_ZERO(&mine);
            bool has = false;
            for (char* q = buf; *q;) {                      // "0-63,128-191"
                char* end;
                const long a = strtol(q, &end, 10);
                if (end == q) break;
                long b = a;
                if (*end == '-') { q = end + 1; b = strtol(q, &end, 10); }
                for (long k = a; k <= b && k < CPU_SETSIZE; ++k) { if (CPU_ISSET(k, &saved)) CPU_SET(k, &mine); if (k == cpu) has = true; }
                q = (*end == ',') ? end + 1 : end;
                if (*end != ',') break;
            }
            if (!has) continue;
            const int cnt = CPU_COUNT(&mine);
            if (cnt > 0 && cnt < CPU_COUNT(&saved) && sched_setaffinity(0, sizeof mine, &mine) == 0) active = true;
            return;
        }
    }
    ~NodeAffinity() { if (active) (void)sched_setaffinity(0, sizeof saved, &saved); }
    NodeAffinity(const NodeAffinity&) = delete;
    NodeAffinity& operator=(const NodeAffinity&) = delete;
};

// buffers of RawVec vectors that were handed to a caller who will free() them (sf_symbolic.h)
static std::mutex g_stolen_mu;
static std::vector<void*> g_stolen;
void raw_mark_stolen(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(g_stolen_mu);
    g_stolen.push_back(p);
}
bool raw_take_if_stolen(void* p) {
    std::lock_guard<std::mutex> g(g_stolen_mu);
    for (size_t k = 0; k < g_stolen.size(); ++k)
        if (g_stolen[k] == p) { g_stolen[k] = g_stolen.back(); g_stolen.pop_back(); return true; }
    return false;
}

namespace {

// reference parameter.h:31-46
bool relax_allowed(Long ncol, double zero_rate) {
    static const Long col_thr[3] = {16, 64, 256};
    static const double rate_thr[3] = {0.8, 0.1, 0.05};
    for (int k = 2; k >= 0; --k)
        if (ncol > col_thr[k] && zero_rate > rate_thr[k]) return false;
    return true;
}

// host threads of the analysis (the bucket sorts below; SF_ANALYZE_THREADS overrides, 1 = the sequential code path of round 1)
int analysis_threads() {
    static const int T = [] {
        if (const char* e = getenv("SF_ANALYZE_THREADS")) return std::max(1, atoi(e));
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(hc ? hc : 1u, 16u);
    }();
    return T;
}

template <class F>
void parallel_ranges(int T, Long n, F&& body) {      // body(t, begin, end) on T host threads over [0, n)
    if (T <= 1 || n < ((Long)1 << 15)) { body(0, (Long)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { body(t, n * t / T, n * (t + 1) / T); });
    body(0, (Long)0, n / T);
    for (auto& x : th) x.join();
}

// The entry stream of P A P^T in the traversal order of C:1036-1063 (new column j = 0 .. n-1, the entries of the original column
// Perm[j] in their stored order): e -> (row i, column j, value) in the NEW numbering.  Built once, in parallel over column ranges;
// every triangle below is a stable bucket sort of it.
struct EntryStream {
    RawVec<Long> row, col;
    RawVec<double> val;
    Long E = 0;
};
void build_stream(Long n, const Long* Cp, const Long* Ci, const double* Cx, const std::vector<Long>& Perm, EntryStream& st) {
    std::vector<Long> Pinv(n, -1);
    for (Long j = 0; j < n; ++j)
        if (Perm[j] >= 0) Pinv[Perm[j]] = j;
    std::vector<Long> off(n + 1, 0);
    for (Long j = 0; j < n; ++j) off[j + 1] = off[j] + (Perm[j] >= 0 ? Cp[Perm[j] + 1] - Cp[Perm[j]] : 0);
    st.E = off[n];
    st.row.resize(st.E); st.col.resize(st.E); st.val.resize(st.E);
    parallel_ranges(analysis_threads(), n, [&](int, Long j0, Long j1) {
        for (Long j = j0; j < j1; ++j) {
            const Long jold = Perm[j];
            if (jold < 0) continue;
            Long e = off[j];
            for (Long p = Cp[jold]; p < Cp[jold + 1]; ++p, ++e) {
                st.row[e] = Pinv[Ci[p]];
                st.col[e] = j;
                st.val[e] = Cx ? Cx[p] : 0.0;
            }
        }
    });
}

// Stable bucket sort of the stream: entry e goes to bucket key(e) (-1: not in this part) carrying idx(e) and its value; inside a
// bucket the entries keep the stream order (= the sequential fill of C:1036-1063).  Thread t owns the buckets [n t/T, n (t+1)/T): it
// scans the whole key stream (sequential reads) and handles the keys of its range -- no atomics, no per-thread histograms, and its
// scattered writes stay inside its own slice of the output.  (Round 4 tried the textbook alternative -- every thread routes the entries
// of its CHUNK of the stream into T per-range lists, then each range is counted and placed from its lists: two passes over the data
// instead of 2 T -- and it lost on the 2-socket host: triangles 50 - 73 -> 80 - 119 ms, renumber + triangles 76 - 90 -> 104 - 122 ms; the
// lists are 200 MB of fresh pages written by 16 threads at once.)
template <class K, class Ix>
void bucket_stream(Long n, const EntryStream& st, int T, K&& key, Ix&& idx, std::vector<Long>& P, RawVec<Long>& I, RawVec<double>& X) {
    P.assign(n + 1, 0);
    parallel_ranges(T, n, [&](int, Long b0, Long b1) {
        for (Long e = 0; e < st.E; ++e) {
            const Long k = key(e);
            if (k >= b0 && k < b1) P[k + 1]++;
        }
    });
    for (Long j = 0; j < n; ++j) P[j + 1] += P[j];
    I.resize(P[n]); X.resize(P[n]);
    parallel_ranges(T, n, [&](int, Long b0, Long b1) {
        std::vector<Long> next(P.begin() + b0, P.begin() + b1);
        for (Long e = 0; e < st.E; ++e) {
            const Long k = key(e);
            if (k >= b0 && k < b1) {
                const Long q = next[k - b0]++;
                I[q] = idx(e);
                X[q] = st.val[e];
            }
        }
    });
}

// lower(P A P^T) by column (column = min(i,j), row = max(i,j)) and its transpose.
// Entry order inside a column follows the traversal order of C:1036-1063.
// LU with an unsymmetric input (L:1179-1282): entry (i,j) of P A P^T goes to L (by column j, rows i >= j) when
// j <= i and to U (by ROW i, columns j >= i) when j >= i; the diagonal is in both.
void build_lu_parts(Long n, const EntryStream& st, Symbolic& S) {
    const int T = std::max(1, analysis_threads() / 4);
    const Long* R = st.row.data();
    const Long* Cc = st.col.data();
    std::thread t1([&] { bucket_stream(n, st, T, [=](Long e) { return Cc[e] <= R[e] ? R[e] : (Long)-1; }, [=](Long e) { return Cc[e]; }, S.LTp, S.LTi, S.LTx); });
    std::thread t2([&] { bucket_stream(n, st, T, [=](Long e) { return Cc[e] >= R[e] ? R[e] : (Long)-1; }, [=](Long e) { return Cc[e]; }, S.Up, S.Ui, S.Ux); });
    std::thread t3([&] { bucket_stream(n, st, T, [=](Long e) { return Cc[e] >= R[e] ? Cc[e] : (Long)-1; }, [=](Long e) { return R[e]; }, S.UTp, S.UTi, S.UTx); });
    bucket_stream(n, st, T, [=](Long e) { return Cc[e] <= R[e] ? Cc[e] : (Long)-1; }, [=](Long e) { return R[e]; }, S.Lp, S.Li, S.Lx);
    t1.join(); t2.join(); t3.join();
}

void build_triangles(Long n, const Long* Cp, const Long* Ci, const double* Cx,
                     const std::vector<Long>& Perm, Symbolic& S) {
    EntryStream st;
    const bool tr_ = getenv("SF_TRACE") != nullptr;
    auto now_ = [] { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; };
    const double t0_ = now_();
    build_stream(n, Cp, Ci, Cx, Perm, st);
    if (tr_) fprintf(stderr, "[sparseframe-hip]   entry stream %.1f ms (%d threads)\n", now_() - t0_, analysis_threads());
    if (S.lu && !S.symmetric) { build_lu_parts(n, st, S); return; }
    // L (by column) and L^T (by row) are two independent stable bucket sorts of the same stream
    const int T = std::max(1, analysis_threads() / 2);
    const Long* R = st.row.data();
    const Long* Cc = st.col.data();
    std::thread t([&] { bucket_stream(n, st, T, [=](Long e) { return std::max(R[e], Cc[e]); }, [=](Long e) { return std::min(R[e], Cc[e]); }, S.LTp, S.LTi, S.LTx); });
    bucket_stream(n, st, T, [=](Long e) { return std::min(R[e], Cc[e]); }, [=](Long e) { return std::max(R[e], Cc[e]); }, S.Lp, S.Li, S.Lx);
    t.join();
}

// The three graph passes below walk arrays of n entries at random (ancestor chains, child lists, disjoint sets): at n = 2M they are
// bound by cache misses, so their WORK arrays are 32-bit whenever n allows (round 4; the results are the same integers, the reference's
// Long arrays at the interface).  I = int32_t for n < 2^31, Long otherwise.

// The decomposition the parallel elimination tree and the parallel row-structure pass share: row j of L only concerns indices in
// [lo(j), j], lo(j) = its leftmost column (from L^T, for the LU pattern also U^T).  tasks: the maximal CLOSED ranges (no row of the
// range reaches left of it; one stack pass finds the closed range ending at every row, they are nested or disjoint) of at most
// n / 8T rows, longest first; top[j] = 1 for the rows outside them (the top separators of a dissection ordering), to be handled
// after all tasks in ascending order.  A row of a task never touches an index of another task or of a top row below it.
void closed_range_tasks(const Symbolic& S, int T, std::vector<std::pair<Long, Long>>& tasks, std::vector<char>& top) {
    const Long n = S.n;
    const bool both = S.lu && !S.symmetric;
    std::vector<Long> start(n), lo(n);
    parallel_ranges(T, n, [&](int, Long j0, Long j1) {
        for (Long j = j0; j < j1; ++j) {
            Long m = j;
            for (Long p = S.LTp[j]; p < S.LTp[j + 1]; ++p) m = std::min(m, S.LTi[p]);
            if (both)
                for (Long p = S.UTp[j]; p < S.UTp[j + 1]; ++p) m = std::min(m, S.UTi[p]);
            lo[j] = m;
        }
    });
    {
        std::vector<std::pair<Long, Long>> st;      // (start, smallest lo) of the maximal closed ranges that tile [0, j)
        st.reserve(1024);
        for (Long j = 0; j < n; ++j) {
            Long a = j, m = lo[j];
            while (m < a) {                         // the row reaches left of its range: swallow the range before it
                a = st.back().first;
                m = std::min(m, st.back().second);
                st.pop_back();
            }
            st.emplace_back(a, m);
            start[j] = a;
        }
    }
    const Long smax = std::max<Long>(1024, n / (8 * (Long)std::max(T, 1)));
    tasks.clear();
    top.assign(n, 0);
    for (Long j = n - 1; j >= 0;) {
        if (j - start[j] + 1 > smax) { top[j] = 1; --j; continue; }
        tasks.emplace_back(start[j], j);
        j = start[j] - 1;
    }
    std::sort(tasks.begin(), tasks.end(), [](const std::pair<Long, Long>& x, const std::pair<Long, Long>& y) {
        const Long sx = x.second - x.first, sy = y.second - y.first;
        return sx != sy ? sx > sy : x.first < y.first;
    });
}

// Liu's elimination tree with path compression over the rows of L (columns of L^T).
// In parallel over CLOSED RANGES (round 4).  Row j climbs from the columns of its entries, all inside [lo(j), j] with lo(j) its
// leftmost column, and only writes Parent / Anc of indices in that interval.  A range [a, b] is closed when lo(j) >= a for every row
// j in it: its rows never leave it.  The closed range ending at every row comes from one stack pass (ranges are nested or disjoint);
// the maximal ones of at most n / (8 T) rows are independent tasks for the T analysis threads, the rows whose range is longer -- the
// top separators of a dissection ordering, a few per cent of the matrix -- follow sequentially in ascending order.  Every row still
// sees every lower row it shares an index with already processed, so the result is the sequential one (the tree is unique anyway).
template <class I>
void elimination_tree_t(const Symbolic& S, std::vector<Long>& Parent) {
    const Long n = S.n;
    std::vector<I> Par(n, (I)-1), Anc(n, (I)-1);
    auto climb = [&](Long i0, Long j) {
        I i = (I)i0;
        const I jj = (I)j;
        while (i >= 0 && i < jj) {
            const I a = Anc[i];
            Anc[i] = jj;
            if (a < 0) { Par[i] = jj; break; }
            if (a == jj) break;
            i = a;
        }
    };
    const bool both = S.lu && !S.symmetric;     // LU: tree of the pattern of L + U^T (L:1358-1383)
    auto row = [&](Long j) {
        for (Long p = S.LTp[j]; p < S.LTp[j + 1]; ++p) climb(S.LTi[p], j);
        if (both)
            for (Long p = S.UTp[j]; p < S.UTp[j + 1]; ++p) climb(S.UTi[p], j);
    };
    const int T = analysis_threads();
    if (T <= 1 || n < 20000) {
        for (Long j = 0; j < n; ++j) row(j);
    } else {
        std::vector<std::pair<Long, Long>> tasks;
        std::vector<char> top;
        closed_range_tasks(S, T, tasks, top);
        std::atomic<size_t> next{0};
        auto worker = [&] {
            for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1))
                for (Long j = tasks[k].first; j <= tasks[k].second; ++j) row(j);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(worker);
        worker();
        for (std::thread& x : th) x.join();
        for (Long j = 0; j < n; ++j)
            if (top[j]) row(j);
    }
    Parent.resize(n);
    for (Long j = 0; j < n; ++j) Parent[j] = Par[j];
}
void elimination_tree(const Symbolic& S, std::vector<Long>& Parent) {
    if (S.n < (Long)0x7fffffff) elimination_tree_t<int32_t>(S, Parent); else elimination_tree_t<Long>(S, Parent);
}

// Postorder of the forest.  With weights (ColCount), children are visited in ascending weight,
// ties in ascending index; without, in ascending index.  Roots in ascending index.
// (The reference threads the children through linked lists filled from weight buckets, C:1175-1199; the visiting order that produces
// is reproduced here with the children of every node laid out contiguously -- a counting sort by weight, then a stable distribution
// to the parents -- so the depth-first walk reads child lists sequentially instead of chasing two pointers per child.)
template <class I>
void postorder_t(const std::vector<Long>& Parent, const std::vector<Long>* Weight, std::vector<Long>& Post) {
    const Long n = (Long)Parent.size();
    std::vector<I> order;                   // the non-root nodes in visiting order of siblings: ascending (weight, index)
    order.reserve(n);
    if (!Weight) {
        for (Long j = 0; j < n; ++j)
            if (Parent[j] >= 0 && Parent[j] < n) order.push_back((I)j);
    } else {
        // weights lie in [0, n] (the reference's buckets are [0, n), C:1175-1199; a weight of n -- first column dense, not a root --
        // would fall outside them there; a bucket is kept for it)
        std::vector<I> start(n + 2, 0);
        for (Long j = 0; j < n; ++j)
            if (Parent[j] >= 0) start[(*Weight)[j] + 1]++;
        for (Long w = 0; w <= n; ++w) start[w + 1] += start[w];
        order.resize((size_t)start[n + 1]);
        for (Long j = 0; j < n; ++j)
            if (Parent[j] >= 0) order[(size_t)start[(*Weight)[j]]++] = (I)j;
    }
    std::vector<I> cptr(n + 1, 0), child(order.size());
    for (I j : order) cptr[Parent[j] + 1]++;
    for (Long p = 0; p < n; ++p) cptr[p + 1] += cptr[p];
    {
        std::vector<I> fill(cptr.begin(), cptr.end() - 1);
        for (I j : order) child[(size_t)fill[Parent[j]]++] = j;
    }
    // No depth-first walk: with the subtree sizes (one ascending pass: parents follow their children in the elimination tree's
    // numbering) every node's position is start + size - 1, where a node's children get consecutive start positions in visiting
    // order (one descending pass over contiguous child lists).  Forests that are not numbered that way (a parent before a child: only
    // a caller's hand-made Parent array could do that) take the stack walk.
    bool topo = true;
    for (Long j = 0; j < n && topo; ++j) topo = Parent[j] < 0 || Parent[j] > j;
    Post.assign(n, -1);
    if (topo) {
        std::vector<I> size(n, 1), start(n, 0);
        for (Long j = 0; j < n; ++j)
            if (Parent[j] >= 0) size[Parent[j]] += size[j];
        I run = 0;
        for (Long r = 0; r < n; ++r)
            if (Parent[r] < 0) { start[r] = run; run += size[r]; }          // roots in ascending index
        for (Long j = n - 1; j >= 0; --j) {
            I sp = start[j];
            for (I q = cptr[j]; q < cptr[j + 1]; ++q) { const I ch = child[(size_t)q]; start[ch] = sp; sp += size[ch]; }
        }
        for (Long j = 0; j < n; ++j) Post[(size_t)start[j] + size[j] - 1] = j;
        return;
    }
    std::vector<I> next(cptr.begin(), cptr.end() - 1);      // next child of a node on the stack
    std::vector<I> Stack;
    Stack.reserve(1024);
    Long k = 0;
    for (Long r = 0; r < n; ++r) {
        if (Parent[r] >= 0) continue;           // roots in ascending index
        Stack.push_back((I)r);
        while (!Stack.empty()) {
            const I j = Stack.back();
            if (next[j] < cptr[j + 1]) Stack.push_back(child[(size_t)next[j]++]);
            else { Stack.pop_back(); Post[k++] = j; }
        }
    }
}
void postorder(const std::vector<Long>& Parent, const std::vector<Long>* Weight, std::vector<Long>& Post) {
    if ((Long)Parent.size() < (Long)0x7ffffff0) postorder_t<int32_t>(Parent, Weight, Post); else postorder_t<Long>(Parent, Weight, Post);
}

// Column counts of L by the skeleton-leaf / disjoint-set method (C:1238-1352).
// In parallel over SUBTREES (round 4).  The maximal subtrees of the elimination tree with at most n / 8T nodes are contiguous in the
// postorder and independent up to the rows ABOVE them: a row inside a subtree only has entries in columns of the subtree, so its
// skeleton is found with the subtree's own part of the arrays; for a row i above the subtree (a proper ancestor of its root r) the
// FIRST column of the subtree with an entry in row i is always a leaf of row i's skeleton (everything before it in the postorder
// lies before the subtree), the later ones are decided inside the subtree (previous neighbour and previous leaf are columns of the
// subtree, their common ancestor lies inside it).  What cannot be decided inside is the common ancestor of that first leaf and the
// row's previous leaf OUTSIDE the subtree, whose count is decremented: it is the lowest ancestor of the previous leaf that has not
// been processed when the subtree starts -- a node above all subtrees -- and is found in the sequential stitch, which walks the
// postorder, treats every subtree as one step (its recorded rows above: one find, one decrement, new previous leaf / neighbour) and
// the nodes above the subtrees as the sequential algorithm does.  A find never enters a later subtree: the parent of a subtree's root
// is such a top node, linked only when the stitch reaches it.  Same counts as the sequential pass, whatever the number of threads.
template <class I>
void column_counts_t(const Symbolic& S, const std::vector<Long>& Parent, const std::vector<Long>& Post,
                     std::vector<Long>& Count) {
    const Long n = S.n;
    std::vector<I> Par(n), First(n, (I)-1), Set(n), PrevLeaf(n), PrevNbr(n, (I)-1), Cnt(n, 0);
    for (Long j = 0; j < n; ++j) Par[j] = (I)Parent[j];
    for (Long k = 0; k < n; ++k) {
        for (I p = (I)Post[k]; p >= 0 && First[p] < 0; p = Par[p]) First[p] = (I)k;
    }
    for (Long j = 0; j < n; ++j) { Set[j] = (I)j; PrevLeaf[j] = (I)j; }
    const bool both = S.lu && !S.symmetric;     // LU: counts of the pattern of L + U^T (L:1601-1625)
    auto find = [&](I x) {
        I r = x;
        while (r != Set[r]) r = Set[r];
        for (I s = x; s != r;) { const I t = Set[s]; Set[s] = r; s = t; }
        return r;
    };
    // the sequential step for column j = Post[k]: all of its entries (C:1300-1340)
    auto column = [&](I j, I k) {
        PrevNbr[j] = k;
        auto visit = [&](Long i0) {
            const I i = (I)i0;
            if (i <= j) return;
            if (First[j] > PrevNbr[i]) {
                const I r = find(PrevLeaf[i]);
                Cnt[j]++;
                Cnt[r]--;
                PrevLeaf[i] = j;
            }
            PrevNbr[i] = k;
        };
        for (Long p = S.Lp[j]; p < S.Lp[j + 1]; ++p) visit(S.Li[p]);
        if (both)
            for (Long p = S.Up[j]; p < S.Up[j + 1]; ++p) visit(S.Ui[p]);
        Set[j] = Par[j];
    };
    const int T = analysis_threads();
    if (T <= 1 || n < 20000) {
        for (Long k = 0; k < n; ++k) column((I)Post[k], (I)k);
    } else {
        std::vector<I> ipost(n), depth(n);
        for (Long k = 0; k < n; ++k) ipost[Post[k]] = (I)k;
        I maxdepth = 0;
        for (Long j = n - 1; j >= 0; --j) {          // parents have larger indices than their children
            depth[j] = Par[j] < 0 ? (I)0 : (I)(depth[Par[j]] + 1);
            maxdepth = std::max(maxdepth, depth[j]);
        }
        const Long smax = std::max<Long>(1024, n / (8 * (Long)T));
        auto size_of = [&](I j) { return (Long)ipost[j] - First[j] + 1; };
        struct Task { I root, k0, k1; };
        std::vector<Task> tasks;                   // in postorder
        for (Long k = 0; k < n; ++k) {
            const I j = (I)Post[k];
            if (size_of(j) <= smax && (Par[j] < 0 || size_of(Par[j]) > smax)) tasks.push_back(Task{j, First[j], (I)k});
        }
        struct Above { I row, last_leaf, last_nbr; };
        std::vector<std::vector<Above>> above(tasks.size());
        std::vector<size_t> by_size(tasks.size());
        for (size_t t = 0; t < tasks.size(); ++t) by_size[t] = t;
        std::sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) {
            const I sx = tasks[x].k1 - tasks[x].k0, sy = tasks[y].k1 - tasks[y].k0;
            return sx != sy ? sx > sy : x < y;
        });
        std::atomic<size_t> next{0};
        auto worker = [&] {
            // rows above the running subtree, by their distance from its root: stamp == task number + 1 marks a live entry
            std::vector<I> stamp((size_t)maxdepth + 1, (I)0), lnbr((size_t)maxdepth + 1), lleaf((size_t)maxdepth + 1);
            for (size_t q = next.fetch_add(1); q < by_size.size(); q = next.fetch_add(1)) {
                const size_t t = by_size[q];
                const Task tk = tasks[t];
                const I dr = depth[tk.root], tag = (I)(t + 1);
                std::vector<Above>& out = above[t];
                for (I k = tk.k0; k <= tk.k1; ++k) {
                    const I j = (I)Post[k];
                    PrevNbr[j] = k;
                    auto visit = [&](Long i0) {
                        const I i = (I)i0;
                        if (i <= j) return;
                        if (ipost[i] <= tk.k1) {           // a row of the subtree: the sequential step
                            if (First[j] > PrevNbr[i]) {
                                const I r = find(PrevLeaf[i]);
                                Cnt[j]++;
                                Cnt[r]--;
                                PrevLeaf[i] = j;
                            }
                            PrevNbr[i] = k;
                            return;
                        }
                        const size_t d = (size_t)(dr - depth[i] - 1);       // i is a proper ancestor of the root
                        if (stamp[d] != tag) {             // first column of the subtree in row i: a leaf, its partner is found by the stitch
                            stamp[d] = tag;
                            Cnt[j]++;
                            lleaf[d] = j;
                            out.push_back(Above{i, 0, 0});
                        } else if (First[j] > lnbr[d]) {
                            const I r = find(lleaf[d]);
                            Cnt[j]++;
                            Cnt[r]--;
                            lleaf[d] = j;
                        }
                        lnbr[d] = k;
                    };
                    for (Long p = S.Lp[j]; p < S.Lp[j + 1]; ++p) visit(S.Li[p]);
                    if (both)
                        for (Long p = S.Up[j]; p < S.Up[j + 1]; ++p) visit(S.Ui[p]);
                    Set[j] = Par[j];
                }
                for (Above& a : out) {
                    const size_t d = (size_t)(dr - depth[a.row] - 1);
                    a.last_leaf = lleaf[d];
                    a.last_nbr = lnbr[d];
                }
            }
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(worker);
            worker();
            for (std::thread& x : th) x.join();
        }
        // the stitch: the postorder with every subtree as one step
        size_t nt = 0;
        for (Long k = 0; k < n;) {
            if (nt < tasks.size() && tasks[nt].k0 == (I)k) {
                for (const Above& a : above[nt]) {
                    const I r = find(PrevLeaf[a.row]);
                    Cnt[r]--;
                    PrevLeaf[a.row] = a.last_leaf;
                    PrevNbr[a.row] = a.last_nbr;
                }
                k = (Long)tasks[nt].k1 + 1;
                ++nt;
            } else {
                column((I)Post[k], (I)k);
                ++k;
            }
        }
    }
    Count.assign(n, 0);
    for (Long k = 0; k < n; ++k) {
        const I j = (I)Post[k];
        if (Par[j] >= 0) Cnt[Par[j]] += Cnt[j];
    }
    for (Long j = 0; j < n; ++j) Count[j] = (Long)Cnt[j] + 1;
}
void column_counts(const Symbolic& S, const std::vector<Long>& Parent, const std::vector<Long>& Post,
                   std::vector<Long>& Count) {
    // (32-bit work arrays: the intermediate counts stay within +-n and the final ones are at most n)
    if (S.n < (Long)0x7ffffff0) column_counts_t<int32_t>(S, Parent, Post, Count); else column_counts_t<Long>(S, Parent, Post, Count);
}

// number of values of a panel with ncol columns and nrow rows in its row list:
// Cholesky nrow*ncol (C:1641), LU (2*nrow - ncol)*ncol (L:1946)
inline Long panel_values(bool lu, Long ncol, Long nrow) { return lu ? ncol * (2 * nrow - ncol) : ncol * nrow; }

inline bool fits_slot(bool lu, Long ncol, Long nrow, size_t slot) {
    // panel values * sizeof(Float) + nrow * sizeof(Long) <= devSlotSize, in size_t arithmetic (C:1482-1484, L:1787-1789)
    return (size_t)panel_values(lu, ncol, nrow) * sizeof(double) + (size_t)nrow * sizeof(Long) <= slot;
}

}  // namespace

static int analyze_any(Long n, const Long* Cp, const Long* Ci, const double* Cx,
                       const Long* perm, size_t devSlotSize, bool lu, bool symmetric, Symbolic& S);

int analyze_cholesky(Long n, const Long* Cp, const Long* Ci, const double* Cx,
                     const Long* perm, size_t devSlotSize, Symbolic& S) {
    return analyze_any(n, Cp, Ci, Cx, perm, devSlotSize, false, true, S);
}

int analyze_lu(Long n, const Long* Cp, const Long* Ci, const double* Cx,
               const Long* perm, size_t devSlotSize, bool symmetric, Symbolic& S) {
    return analyze_any(n, Cp, Ci, Cx, perm, devSlotSize, true, symmetric, S);
}

static int analyze_any(Long n, const Long* Cp, const Long* Ci, const double* Cx,
                       const Long* perm, size_t devSlotSize, bool lu, bool symmetric, Symbolic& S) {
    if (n < 0 || !Cp || (n > 0 && !Ci)) return 1;
    NodeAffinity on_one_node;
    S = Symbolic();
    S.n = n;
    S.devSlotSize = devSlotSize;
    S.lu = lu;
    S.symmetric = symmetric;

    std::vector<Long> Perm(n);
    for (Long j = 0; j < n; ++j) Perm[j] = perm ? perm[j] : j;
    if (perm) {  // must be a permutation
        std::vector<char> seen(n, 0);
        for (Long j = 0; j < n; ++j) {
            if (Perm[j] < 0 || Perm[j] >= n || seen[Perm[j]]) return 1;
            seen[Perm[j]] = 1;
        }
    }
    for (Long p = 0; p < Cp[n]; ++p)
        if (Ci[p] < 0 || Ci[p] >= n) return 1;

    const bool tr_ = getenv("SF_TRACE") != nullptr;
    auto now_ = [] { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; };
    double tm_[8]; tm_[0] = now_();
    build_triangles(n, Cp, Ci, Cx, Perm, S);
    tm_[1] = now_();

    std::vector<Long> Parent, Post, Count;
    double te_[5]; te_[0] = now_();
    elimination_tree(S, Parent);
    te_[1] = now_();
    postorder(Parent, nullptr, Post);
    te_[2] = now_();
    column_counts(S, Parent, Post, Count);
    te_[3] = now_();
    postorder(Parent, &Count, Post);
    te_[4] = now_();
    if (tr_) fprintf(stderr, "[sparseframe-hip]   etree %.1f ms, postorder %.1f ms, column counts %.1f ms, weighted postorder %.1f ms\n",
                     te_[1] - te_[0], te_[2] - te_[1], te_[3] - te_[2], te_[4] - te_[3]);

    tm_[2] = now_();
    S.Post = Post;
    S.Parent0 = Parent;
    S.ColCount0 = Count;

    // ---- renumber into the weighted postorder and rebuild the triangles (C:1429-1447) ----
    std::vector<Long> InvPost(n);
    for (Long k = 0; k < n; ++k) InvPost[Post[k]] = k;
    S.Perm.resize(n); S.Parent.resize(n); S.ColCount.resize(n);
    for (Long k = 0; k < n; ++k) {
        const Long old = Post[k];
        S.Perm[k] = Perm[old];
        S.Parent[k] = Parent[old] < 0 ? -1 : InvPost[Parent[old]];
        S.ColCount[k] = Count[old];
    }
    // the triangles of the final numbering are needed by the row-structure pass only: they are rebuilt (by their own threads)
    // while this thread finds the supernodes, which read Parent and ColCount and nothing else (round 4)
    std::thread tri([&] { build_triangles(n, Cp, Ci, Cx, S.Perm, S); });
    struct JoinTri { std::thread& t; ~JoinTri() { if (t.joinable()) t.join(); } } join_tri{tri};
    tm_[3] = now_();

    const std::vector<Long>& Par = S.Parent;
    const std::vector<Long>& CC = S.ColCount;

    // ---- fundamental supernodes, capped by the slot size (C:1462-1522) ----
    std::vector<Long> Nchild(n, 0);
    for (Long j = 0; j < n; ++j)
        if (Par[j] >= 0 && Par[j] < n) Nchild[Par[j]]++;

    std::vector<Long> Fsuper;  // first column of each fundamental supernode
    if (n > 0) Fsuper.push_back(0);
    for (Long j = 1; j < n; ++j) {
        const Long first = Fsuper.back();
        const bool chain_breaks = (Par[j - 1] != j) || (CC[j - 1] != CC[j] + 1) || (Nchild[j] > 1);
        if (chain_breaks || !fits_slot(lu, j - first + 1, CC[first], devSlotSize)) Fsuper.push_back(j);
    }
    const Long nf = (Long)Fsuper.size();
    Fsuper.push_back(n);
    S.nfsuper = nf;

    std::vector<Long> Nscol(nf), Scc(nf), Fmap(n), Fparent(nf);
    for (Long s = 0; s < nf; ++s) {
        Nscol[s] = Fsuper[s + 1] - Fsuper[s];
        Scc[s] = CC[Fsuper[s]];
        for (Long j = Fsuper[s]; j < Fsuper[s + 1]; ++j) Fmap[j] = s;
    }
    for (Long s = 0; s < nf; ++s) {
        const Long p = Par[Fsuper[s + 1] - 1];
        Fparent[s] = p < 0 ? -1 : Fmap[p];
    }

    // ---- relaxed amalgamation of a last child into its parent (C:1524-1622) ----
    std::vector<Long> Merge(nf), Nsz(nf, 0);
    for (Long s = 0; s < nf; ++s) Merge[s] = s;
    for (Long s = nf - 2; s >= 0; --s) {
        const Long sp = Fparent[s];
        if (sp < 0 || sp >= nf || Merge[s + 1] != Merge[sp]) continue;
        const Long g = Merge[sp];
        const Long s_n = Nscol[s], p_n = Nscol[g], s_c = Scc[s], p_c = Scc[g];
        if (!fits_slot(lu, s_n + p_n, s_n + p_c, devSlotSize)) continue;
        const Long total_zero = Nsz[s] + Nsz[g] + s_n * (s_n + p_c - s_c);
        const Long tot = s_n + p_n;
        const Long denom = tot * (tot + 1) / 2 + tot * (p_c - p_n);
        if (relax_allowed(tot, (double)total_zero / (double)denom)) {
            Nscol[g] = tot;
            Scc[g] = s_n + p_c;
            Nsz[g] = total_zero;
            Merge[s] = g;
        }
    }

    // compact: a merged group is numbered by its representative, its first column is the first
    // column of its earliest member (members are consecutive, ending at the representative)
    S.Super.clear();
    std::vector<Long> Gncol, Gcc;
    S.Super.push_back(0);
    for (Long s = 0; s < nf; ++s) {
        if (Merge[s] == s) {
            S.Super.push_back(Fsuper[s + 1]);
            Gncol.push_back(Nscol[s]);
            Gcc.push_back(Scc[s]);
        }
    }
    const Long ns = (Long)Gncol.size();
    S.nsuper = ns;
    if (n == 0) { S.Super.assign(1, 0); }
    S.Super[ns] = n;

    S.SuperMap.assign(n, 0);
    for (Long s = 0; s < ns; ++s)
        for (Long j = S.Super[s]; j < S.Super[s + 1]; ++j) S.SuperMap[j] = s;
    S.Sparent.assign(ns, -1);
    for (Long s = 0; s < ns; ++s) {
        const Long p = Par[S.Super[s + 1] - 1];
        S.Sparent[s] = p < 0 ? -1 : S.SuperMap[p];
    }

    // ---- pointers (C:1632-1645) ----
    S.Lsip.assign(ns + 1, 0);
    S.Lsxp.assign(ns + 1, 0);
    for (Long s = 0; s < ns; ++s) {
        S.Lsip[s + 1] = S.Lsip[s] + Gcc[s];
        S.Lsxp[s + 1] = S.Lsxp[s] + panel_values(lu, Gncol[s], Gcc[s]);
    }
    S.isize = S.Lsip[ns];
    S.xsize = S.Lsxp[ns];

    tri.join();
    tm_[4] = now_();
    // ---- row structure: own columns, then every row j that reaches the supernode through
    //      the supernodal tree from the supernode of a nonzero (j,i), i<=j (C:1660-1692) ----
    S.Lsi.resize(S.isize);            // (uninitialised: every entry is written below -- the fill counts are checked against Lsip at the end)
    {
        std::vector<Long> fill(S.Lsip.begin(), S.Lsip.end() - 1), Marker(ns);
        for (Long s = 0; s < ns; ++s) {
            Marker[s] = S.Super[s + 1];
            for (Long k = S.Super[s]; k < S.Super[s + 1]; ++k) S.Lsi[fill[s]++] = k;
        }
        const bool both = lu && !symmetric;
        std::atomic<int> bad{0};
        auto row = [&](Long j) {
            for (int pass = 0; pass < (both ? 2 : 1); ++pass) {
                const std::vector<Long>& Tp = pass ? S.UTp : S.LTp;
                const RawVec<Long>& Ti = pass ? S.UTi : S.LTi;
                for (Long p = Tp[j]; p < Tp[j + 1]; ++p) {
                    for (Long d = S.SuperMap[Ti[p]]; d >= 0 && Marker[d] <= j; d = S.Sparent[d]) {
                        if (fill[d] >= S.Lsip[d + 1]) { bad.store(1); return; }  // count mismatch: symbolic inconsistency
                        S.Lsi[fill[d]++] = j;
                        Marker[d] = j + 1;
                    }
                }
            }
        };
        // Row j walks up the supernodal tree from the supernodes of its entries while their columns start at or before j: all
        // inside [lo(j), j].  With the closed-range tasks of the elimination tree (closed_range_tasks, here on the FINAL numbering) a
        // supernode only receives rows of its own task, in ascending order, and afterwards rows of the sequential top part: the same
        // lists as the sequential pass (round 4: 65 -> 20 ms at 128^3).
        const int T = analysis_threads();
        if (T <= 1 || n < 20000) {
            for (Long j = 0; j < n && !bad.load(std::memory_order_relaxed); ++j) row(j);
        } else {
            std::vector<std::pair<Long, Long>> tasks;
            std::vector<char> top;
            closed_range_tasks(S, T, tasks, top);
            std::atomic<size_t> next{0};
            auto worker = [&] {
                for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1))
                    for (Long j = tasks[k].first; j <= tasks[k].second; ++j) row(j);
            };
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(worker);
            worker();
            for (std::thread& x : th) x.join();
            for (Long j = 0; j < n; ++j)
                if (top[j]) row(j);
        }
        if (bad.load()) return 2;
        for (Long s = 0; s < ns; ++s)
            if (fill[s] != S.Lsip[s + 1]) return 2;
    }

    tm_[5] = now_();
    // ---- csize: largest update block any descendant->ancestor pair needs (C:1694-1719) ----
    S.csize = 0;
    for (Long s = 0; s < ns; ++s) {
        const Long nscol = S.Super[s + 1] - S.Super[s];
        const Long nsrow = S.Lsip[s + 1] - S.Lsip[s];
        if (nscol >= nsrow) continue;
        const Long* rows = &S.Lsi[S.Lsip[s]];
        Long start = nscol;
        Long owner = S.SuperMap[rows[nscol]];
        for (Long si = nscol; si < nsrow; ++si) {
            const Long o = S.SuperMap[rows[si]];
            if (o != owner) {
                // Cholesky (si-start)*(nsrow-start) (C:1711); LU adds the U^T rows: (si-start)*(2*nsrow-si-start) (L:2028)
                S.csize = std::max(S.csize, (si - start) * (lu ? (2 * nsrow - si - start) : (nsrow - start)));
                start = si;
                owner = o;
            }
        }
        S.csize = std::max(S.csize, (nsrow - start) * (nsrow - start));
    }

    // ---- stages: greedy root-to-leaf packing into slot-sized groups (C:1721-1846) ----
    {
        std::vector<Long> Head(std::max<Long>(ns, 1), -1), Next(std::max<Long>(ns, 1), -1);
        std::vector<Long> Asz(std::max<Long>(ns, 1), 0), Msz(std::max<Long>(ns, 1), 0);
        S.ST_Map.assign(ns, -1);
        Long nstage = ns > 0 ? 1 : 0;
        auto stage_fits = [&](Long st, Long a, Long m) {
            return (size_t)(Asz[st] + a) * sizeof(double) + (size_t)(Msz[st] + m) * sizeof(Long) <= devSlotSize;
        };
        for (Long s = ns - 1; s >= 0; --s) {
            const Long a = panel_values(lu, S.Super[s + 1] - S.Super[s], S.Lsip[s + 1] - S.Lsip[s]);
            const Long m = S.Lsip[s + 1] - S.Lsip[s];
            Long st;
            const Long sp = S.Sparent[s];
            if (sp >= 0) {
                st = S.ST_Map[sp];
                if (stage_fits(st, a, m)) {
                    S.ST_Map[s] = st; Asz[st] += a; Msz[st] += m;
                    continue;
                }
                st = Head[S.ST_Map[sp]];
            } else {
                st = 0;
            }
            while (st >= 0) {
                if (stage_fits(st, a, m)) {
                    S.ST_Map[s] = st; Asz[st] += a; Msz[st] += m;
                    break;
                }
                st = Next[st];
            }
            if (st < 0) {
                S.ST_Map[s] = nstage;
                Asz[nstage] = a;
                Msz[nstage] = m;
                if (sp >= 0) {
                    Next[nstage] = Head[S.ST_Map[sp]];
                    Head[S.ST_Map[sp]] = nstage;
                } else {
                    Next[nstage] = Next[0];
                    Next[0] = nstage;
                }
                nstage++;
            }
        }
        for (Long s = 0; s < ns; ++s) S.ST_Map[s] = nstage - 1 - S.ST_Map[s];
        S.nstage = nstage;
        S.ST_Pointer.assign(nstage + 1, 0);
        S.ST_Index.assign(ns, 0);
        for (Long s = 0; s < ns; ++s) S.ST_Pointer[S.ST_Map[s] + 1]++;
        for (Long st = 0; st < nstage; ++st) S.ST_Pointer[st + 1] += S.ST_Pointer[st];
        std::vector<Long> fill(S.ST_Pointer.begin(), S.ST_Pointer.end() - 1);
        for (Long s = 0; s < ns; ++s) S.ST_Index[fill[S.ST_Map[s]]++] = s;
    }

    // ---- leaf queue in stage order (C:1848-1873) ----
    {
        std::vector<char> has_child(std::max<Long>(ns, 1), 0);
        for (Long s = 0; s < ns; ++s) {
            const Long nscol = S.Super[s + 1] - S.Super[s];
            const Long nsrow = S.Lsip[s + 1] - S.Lsip[s];
            if (nscol < nsrow) has_child[S.SuperMap[S.Lsi[S.Lsip[s] + nscol]]] = 1;
        }
        S.LeafQueue.assign(ns, -1);
        S.nsleaf = 0;
        for (Long k = 0; k < ns; ++k) {
            const Long s = S.ST_Index[k];
            if (!has_child[s]) S.LeafQueue[S.nsleaf++] = s;
        }
    }

    // ---- byte offsets of each panel / relative map inside its stage's slot (C:1875-1904) ----
    S.Aoffset.assign(ns, 0);
    S.Moffset.assign(ns, 0);
    for (Long st = 0; st < S.nstage; ++st) {
        size_t asz = 0, msz = 0;
        for (Long k = S.ST_Pointer[st]; k < S.ST_Pointer[st + 1]; ++k) {
            const Long s = S.ST_Index[k];
            const Long nscol = S.Super[s + 1] - S.Super[s];
            const Long nsrow = S.Lsip[s + 1] - S.Lsip[s];
            S.Aoffset[s] = (Long)asz;
            asz += (size_t)panel_values(lu, nscol, nsrow) * sizeof(double);
            S.Moffset[s] = (Long)msz;
            msz += (size_t)nsrow * sizeof(Long);
        }
        for (Long k = S.ST_Pointer[st]; k < S.ST_Pointer[st + 1]; ++k) S.Moffset[S.ST_Index[k]] += (Long)asz;
    }
    if (tr_)
        fprintf(stderr, "[sparseframe-hip] analyze: triangles %.1f ms, etree + postorder + column counts %.1f ms, renumber %.1f ms, "
                        "supernodes beside the triangles' rebuild %.1f ms, row structure %.1f ms, csize + stages + offsets %.1f ms\n", tm_[1] - tm_[0], tm_[2] - tm_[1],
                tm_[3] - tm_[2], tm_[4] - tm_[3], tm_[5] - tm_[4], now_() - tm_[5]);
    return 0;
}

double flops_struct(const Symbolic& S) {
    double f = 0;
    // Cholesky: sum c^2 ; no-pivot LU on the symmetrised pattern: sum (c-1) + 2 (c-1)^2   
    for (Long c : S.ColCount0)
        f += S.lu ? ((double)(c - 1) + 2.0 * (double)(c - 1) * (double)(c - 1)) : (double)c * (double)c;
    return f;
}

double flops_exec(const Symbolic& S, double* update_flops, double* scatter_elems) {
    double fac = 0, upd = 0, sc = 0;
    for (Long s = 0; s < S.nsuper; ++s) {
        const double n = (double)(S.Super[s + 1] - S.Super[s]);
        const Long nsrow = S.Lsip[s + 1] - S.Lsip[s];
        const double m = (double)nsrow - n;
        // Cholesky n^3/3 + m n^2 ; LU r n^2 - n^3/3 + m n^2 with r = nsrow
        fac += S.lu ? ((double)nsrow * n * n - n * n * n / 3.0 + m * n * n) : (n * n * n / 3.0 + m * n * n);
        const Long* rows = &S.Lsi[S.Lsip[s]];
        Long i = (Long)n;
        while (i < nsrow) {
            const Long owner = S.SuperMap[rows[i]];
            Long e = i;
            while (e < nsrow && S.SuperMap[rows[e]] == owner) ++e;
            const double dn = (double)(e - i), dm = (double)(nsrow - e);
            // LU: two GEMMs per update, (dn+dm) x dn x dk and dm x dn x dk (L:2570-2577)
            upd += S.lu ? (2.0 * (dn + dm) * dn * n + 2.0 * dm * dn * n) : (dn * (dn + 1) * n + 2.0 * dm * dn * n);
            sc += S.lu ? ((dn + dm) * dn + dm * dn) : (dn * (dn + 1) / 2.0 + dm * dn);
            i = e;
        }
    }
    if (update_flops) *update_flops = upd;
    if (scatter_elems) *scatter_elems = sc;
    return fac + upd;
}

// ---------------------------------------------------------------------------------------------
// Subtree-to-rank mapping for multi-GPU sharding.  The supernodal tree is cut from the roots down:
// the heaviest remaining subtree is split (its root joins the replicated "top" set, its children become
// subtrees) while that lowers the estimate  t = flops(top) + max_rank(sum of its subtrees' flops)  with the
// subtrees placed longest-first on the least loaded rank.  owner[s] = rank of the subtree holding s, or -1 (top).
// Work of a supernode = its own factorization + every update it pushes to its ancestors (right-looking).
// ---------------------------------------------------------------------------------------------
int subtree_partition(Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi,
                      int nranks, int32_t* owner, double* top_fraction, double* max_load_fraction, double top_weight) {
    if (nsuper < 0 || nranks < 1 || !owner || !(top_weight > 0)) return 1;
    std::vector<Long> par(nsuper, -1);
    std::vector<double> fl(nsuper, 0.0), sub(nsuper, 0.0);
    std::vector<std::vector<Long>> kids(nsuper);
    double total = 0;
    for (Long s = 0; s < nsuper; ++s) {
        const double k = (double)(Super[s + 1] - Super[s]);
        const Long nsrow = Lsip[s + 1] - Lsip[s];
        const double m = (double)nsrow - k;
        double f = k * k * k / 3.0 + m * k * k;
        const Long* rows = Lsi + Lsip[s];
        Long i = (Long)k;
        if (i < nsrow) par[s] = SuperMap[rows[i]];
        while (i < nsrow) {
            const Long o = SuperMap[rows[i]];
            Long e = i;
            while (e < nsrow && SuperMap[rows[e]] == o) ++e;
            const double dn = (double)(e - i), dm = (double)(nsrow - e);
            f += dn * (dn + 1) * k + 2.0 * dm * dn * k;
            i = e;
        }
        fl[s] = f;
        total += f;
    }
    for (Long s = 0; s < nsuper; ++s) {
        sub[s] += fl[s];
        if (par[s] >= 0) { sub[par[s]] += sub[s]; kids[par[s]].push_back(s); }
    }
    std::vector<Long> roots;            // current subtree roots
    for (Long s = 0; s < nsuper; ++s)
        if (par[s] < 0) roots.push_back(s);
    std::vector<char> is_top(nsuper, 0);
    double top = 0;
    auto estimate = [&](const std::vector<Long>& rs, std::vector<int>* assign) {
        std::vector<Long> order(rs);
        std::sort(order.begin(), order.end(), [&](Long a, Long b) { return sub[a] != sub[b] ? sub[a] > sub[b] : a < b; });
        std::vector<double> load(nranks, 0.0);
        if (assign) assign->assign(nsuper, -1);
        for (Long r : order) {
            int best = 0;
            for (int q = 1; q < nranks; ++q)
                if (load[q] < load[best]) best = q;
            load[best] += sub[r];
            if (assign) (*assign)[r] = best;
        }
        return *std::max_element(load.begin(), load.end());
    };
    std::vector<Long> best_roots = roots;
    std::vector<char> best_top = is_top;
    double best_t = top_weight * top + estimate(roots, nullptr), best_topf = top;
    int stale = 0;
    while (nranks > 1 && stale < 4 * nranks) {
        // split the heaviest subtree that has children
        Long pick = -1;
        size_t pos = 0;
        for (size_t k = 0; k < roots.size(); ++k)
            if (!kids[roots[k]].empty() && (pick < 0 || sub[roots[k]] > sub[pick])) { pick = roots[k]; pos = k; }
        if (pick < 0) break;
        roots.erase(roots.begin() + pos);
        is_top[pick] = 1;
        top += fl[pick];
        for (Long c : kids[pick]) roots.push_back(c);
        const double t = top_weight * top + estimate(roots, nullptr);
        if (t < best_t * (1.0 - 1e-12)) { best_t = t; best_roots = roots; best_top = is_top; best_topf = top; stale = 0; }
        else ++stale;
    }
    std::vector<int> assign;
    const double maxload = estimate(best_roots, &assign);
    // propagate the owner of each subtree root to its descendants (children precede parents in the postorder)
    std::vector<int32_t> own(nsuper, -1);
    for (Long s = nsuper - 1; s >= 0; --s) {
        if (best_top[s]) { own[s] = -1; continue; }
        if (assign[s] >= 0) own[s] = assign[s];
        else own[s] = (par[s] >= 0) ? own[par[s]] : 0;
    }
    for (Long s = 0; s < nsuper; ++s) owner[s] = own[s];
    if (top_fraction) *top_fraction = total > 0 ? best_topf / total : 0;
    if (max_load_fraction) *max_load_fraction = total > 0 ? maxload / total : 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Grouping for a factor that does not fit the device (the reference's answer to the same question is its slot-sized "stages"
// streamed through the device, C:1721-1846 / C:2421-2467).  The supernodal tree is cut at a size S: every maximal
// subtree of at most S panel entries is a unit, consecutive units (postorder) are packed into GROUPS of at most S entries, every
// supernode above the cut is "top".  The groups stream through two alternating buffers of S entries: group g is factorized while
// group g - 1 travels to the host.  The top panels receive the Schur updates of everything below them and are
//   mode 0  resident for the whole factorization, factorized after the last group (device need = all top panels + 2 S), or
//   mode 1  resident only while they are ACTIVE: a top supernode is allocated when the first group below it starts, factorized
//           right after the last one, and its space is used again two groups later (by then its copy has left the device):
//           need = the largest set of simultaneously active top panels (laid out first-fit by ooc_top_layout) + 2 S.  Slower --
//           a top supernode no longer shares launches with its whole level -- so it is only chosen when mode 0 does not fit;
//   mode 2  as mode 1 with the space used again ONE group later: the device then waits for the copy of the panel that was there
//           (the only mode with such waits), need = the active path of top panels + 2 S.
// The cut is the LARGEST S of a geometric ladder that fits `budget` entries, mode 0 before mode 1 before mode 2.
// group[s] in [0, *ngroups) or -1 (top).  Returns 0; 2 when nothing fits (group[] then holds the cheapest cut, *need what it takes).
// ---------------------------------------------------------------------------------------------
namespace {
struct OocTree {
    std::vector<Long> par;
    std::vector<int64_t> sz, sub;
    int64_t total = 0;
};
int ooc_tree(Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi, OocTree& T) {
    T.par.assign((size_t)nsuper, -1);
    T.sz.assign((size_t)nsuper, 0);
    T.sub.assign((size_t)nsuper, 0);
    T.total = 0;
    for (Long s = 0; s < nsuper; ++s) {
        const Long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
        T.sz[s] = (int64_t)nscol * nsrow;
        T.total += T.sz[s];
        if (nscol < nsrow) {
            T.par[s] = SuperMap[Lsi[Lsip[s] + nscol]];
            if (T.par[s] <= s || T.par[s] >= nsuper) return 1;      // postordered
        }
    }
    for (Long s = 0; s < nsuper; ++s) {
        T.sub[s] += T.sz[s];
        if (T.par[s] >= 0) T.sub[T.par[s]] += T.sub[s];
    }
    return 0;
}
}  // namespace

// Mode 1 / 2 layout of the top panels of a grouping (mode 2: a place is given to the next panel ONE group after its panel's last
// group instead of two -- the copy of the old panel may then still be on its way and the new one's first launch waits for it:
// wait[s] = the group whose copies must be over first): first[s] / last[s] = the first and the last group below top supernode s (a top
// supernode without grouped descendants: the group before it), off[s] = its offset in the top arena (-1 for grouped supernodes),
// *arena = the arena's size.  A panel is live from the start of group first[s] until two groups after last[s]; panels whose lives
// overlap never share addresses (first fit, ancestors before descendants).
int ooc_top_layout(Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi, const int32_t* group, int ngroups,
                   int mode, int32_t* first, int32_t* last, int64_t* off, int32_t* wait, int64_t* arena) {
    if (nsuper < 0 || !group || ngroups < 1 || !first || !last || !off || !arena || (mode != 1 && mode != 2)) return 1;
    const int lag = mode == 1 ? 2 : 1;          // a place is free again `lag` groups after its panel's last group
    OocTree T;
    if (ooc_tree(nsuper, Super, SuperMap, Lsip, Lsi, T)) return 1;
    int32_t prev = 0;
    for (Long s = 0; s < nsuper; ++s) { first[s] = INT32_MAX; last[s] = -1; off[s] = -1; if (wait) wait[s] = -1; }
    for (Long s = 0; s < nsuper; ++s) {
        if (group[s] >= 0) { first[s] = last[s] = group[s]; prev = group[s]; }
        else if (last[s] < 0) first[s] = last[s] = prev;
        if (T.par[s] >= 0) {
            const Long p = T.par[s];
            first[p] = std::min(first[p], first[s]);
            last[p] = std::max(last[p], last[s]);
        }
    }
    // by unit: release what ended two groups ago, then place what starts now -- the larger index (the ancestor, which lives longest) first
    std::vector<std::vector<Long>> starts((size_t)ngroups);
    for (Long s = nsuper - 1; s >= 0; --s)
        if (group[s] < 0) starts[(size_t)first[s]].push_back(s);
    struct Live { int64_t off, size; int32_t until; };
    std::vector<Live> live, gone;       // live: sorted by offset; gone: every place that was given back, with the group its panel ended with
    int64_t top = 0;
    for (int u = 0; u < ngroups; ++u) {
        for (const Live& L : live)
            if (L.until + lag <= u) gone.push_back(L);
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& L) { return L.until + lag <= u; }), live.end());
        for (Long s : starts[(size_t)u]) {
            const int64_t size = T.sz[s];
            int64_t at = 0;
            size_t pos = 0;
            for (; pos < live.size(); ++pos) {
                if (live[pos].off - at >= size) break;
                at = live[pos].off + live[pos].size;
            }
            live.insert(live.begin() + (std::ptrdiff_t)pos, Live{at, size, last[s]});
            off[s] = at;
            top = std::max(top, at + size);
            if (wait) {         // the latest group whose copy to the host has to be over before this place may be written again
                int32_t w = -1;
                for (const Live& G : gone)
                    if (G.off < at + size && at < G.off + G.size) w = std::max(w, G.until);
                wait[s] = w;
            }
        }
    }
    *arena = top;
    return 0;
}

int ooc_partition(Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi, int64_t budget,
                  int32_t* group, int* ngroups, int64_t* group_entries, int64_t* top_entries, int64_t* need, int* top_mode) {
    if (nsuper < 0 || !group || !ngroups || (nsuper > 0 && (!Super || !SuperMap || !Lsip || !Lsi))) return 1;
    OocTree T;
    if (ooc_tree(nsuper, Super, SuperMap, Lsip, Lsi, T)) return 1;
    const std::vector<Long>& par = T.par;
    const std::vector<int64_t>& sz = T.sz;
    const std::vector<int64_t>& sub = T.sub;
    const int64_t total = T.total;
    constexpr int MAX_GROUPS = 4096;
    std::vector<int32_t> f((size_t)std::max<Long>(nsuper, 1)), l((size_t)std::max<Long>(nsuper, 1));
    std::vector<int64_t> o((size_t)std::max<Long>(nsuper, 1));
    // the cut at S: returns the entries that have to be resident; mode 1: the top's share of that is the arena of ooc_top_layout
    auto cut = [&](int64_t S, int mode, int32_t* out, int* ng, int64_t* gmax, int64_t* top) -> int64_t {
        int64_t t = 0, cur = 0, mx = 0;
        int g = 0;
        bool open = false;
        Long prev_top = -2;
        for (Long s = 0; s < nsuper; ++s) {
            if (sub[s] > S) { t += sz[s]; out[s] = -1; continue; }
            const bool unit_root = par[s] < 0 || sub[par[s]] > S;
            if (!unit_root) continue;
            // the unit = supernodes (s - its descendants .. s]: a contiguous range of the postorder that ends at s
            // (modes 1 / 2: a group never holds subtrees of two different top supernodes -- the lives of sibling top panels would
            // overlap in that group and the arena would have to hold both)
            const bool other_parent = mode >= 1 && open && par[s] != prev_top;
            if (open && (cur + sub[s] > S || other_parent)) { ++g; cur = 0; }
            prev_top = par[s];
            open = true;
            cur += sub[s];
            mx = std::max(mx, cur);
            out[s] = g;
        }
        // descendants take their unit root's group (parents come later in the postorder: walk down)
        for (Long s = nsuper - 1; s >= 0; --s)
            if (sub[s] <= S && par[s] >= 0 && sub[par[s]] <= S) out[s] = out[par[s]];
        const int n_g = open ? g + 1 : 0;
        if (mode >= 1 && n_g > 1 && n_g <= MAX_GROUPS) {
            int64_t arena = 0;
            if (ooc_top_layout(nsuper, Super, SuperMap, Lsip, Lsi, out, n_g, mode, f.data(), l.data(), o.data(), nullptr, &arena) == 0) t = arena;
        }
        if (ng) *ng = n_g;
        if (gmax) *gmax = mx;
        if (top) *top = t;
        return t + (n_g > 1 ? 2 : 1) * mx;
    };
    int64_t bestS = total, best_need = total;       // everything in one group: in core
    int best_mode = 0;
    bool fits = total <= budget;
    if (!fits) {
        best_need = INT64_MAX;
        // modes in the order of their speed, each with at most 64 groups (hundreds of tiny groups cost more than the next mode does);
        // then mode 2 with whatever number of groups it takes
        for (int pass = 0; pass < 4 && !fits; ++pass) {
            const int mode = std::min(pass, 2), max_groups = pass < 3 ? 64 : MAX_GROUPS;
            double S = (double)budget / 2.0;
            for (int it = 0; it < 96 && S >= 1.0; ++it, S *= 0.85) {
                int ng = 0;
                const int64_t nd = cut((int64_t)S, mode, group, &ng, nullptr, nullptr);
                if (ng > max_groups) break;
                if (nd < best_need) { best_need = nd; bestS = (int64_t)S; best_mode = mode; }
                if (nd <= budget) { fits = true; break; }
            }
        }
    }
    int ng = 0;
    int64_t gmax = 0, top = 0;
    const int64_t nd = cut(bestS, best_mode, group, &ng, &gmax, &top);
    *ngroups = ng;
    if (group_entries) *group_entries = gmax;
    if (top_entries) *top_entries = top;
    if (need) *need = nd;
    if (top_mode) *top_mode = (ng > 1) ? best_mode : 0;
    return fits ? 0 : 2;
}

// ---------------------------------------------------------------------------------------------
// Built-in fill-reducing ordering for general patterns: nested dissection by BFS level structures.
// (The reference calls METIS_NodeND, Cholesky/Source/SparseFrame.c:942, a third-party library; this is NOT a
// restatement of METIS -- it is a self-contained stand-in so that SparseFrame_analyze is usable without a caller-
// supplied ordering.  Orderings are not part of the parity contract: every downstream array is defined for a GIVEN Perm.)
//
// For a connected piece: BFS from a pseudo-peripheral vertex (two sweeps), the vertex separator is the BFS level
// that best balances the two sides; the two sides are ordered first (recursively), the separator last.  Pieces
// of at most `leaf` vertices are ordered by BFS (reverse Cuthill-McKee like), which keeps their fill banded.
// ---------------------------------------------------------------------------------------------
namespace {
// The dissection of two pieces is independent: they own disjoint vertices and -- because a piece's vertices are ordered A, then B,
// then the separator -- disjoint, known ranges of the output.  So every call carries its output offset, piece ids come from one
// atomic counter, and a big enough first half is dissected by another host thread (at most analysis_threads() at a time) while the
// caller does the second: same permutation as the sequential code, whatever the thread count.  mark[] is read across pieces (a
// neighbour in another piece) while its owner may be renumbering it: those accesses are relaxed atomics, and the only thing a
// reader asks is "is it MY id", which no other piece's id ever equals.
// vertex ids, piece ids and BFS levels are 32-bit inside the dissection (n < 2^30, checked by graph_nd_perm): the sweeps are bound by
// memory traffic over the adjacency, mark and level arrays, and halving them is a quarter of the ordering's time (round 4)
using Vx = int32_t;
// software prefetch in the sweeps: 1 = the team sweeps of the big pieces only (2M-vertex sweep 45 -> 25 ms); 2 = also the sequential sweeps of
// the small pieces, which are mostly cache-resident -- no measurable difference in the whole ordering (0.25 - 0.31 s either way, box to box)
static const int g_nd_prefetch = 1;
struct NdCtx {
    const std::vector<Long>& Ap;
    const std::vector<Vx>& Ai;
    std::vector<Vx> mark;      // mark[v] = id of the piece v currently belongs to
    std::vector<Vx> level;
    Long* out;
    Vx leaf;
    std::atomic<Vx> next_id{1};
    std::atomic<int> helpers{0};
    int max_helpers = 0;
    Vx mk(Vx v) const { return __atomic_load_n(&mark[v], __ATOMIC_RELAXED); }
    void set_mk(Vx v, Vx id) { __atomic_store_n(&mark[v], id, __ATOMIC_RELAXED); }
};

// BFS inside piece `id` from `root`; returns the vertices in BFS order and fills level[]
void nd_bfs(NdCtx& c, Vx id, Vx root, std::vector<Vx>& order) {
    order.clear();
    order.push_back(root);
    c.level[root] = 0;
    c.set_mk(root, -2 - id);        // visited: -2 - id (unique to the piece as well)
    for (size_t h = 0; h < order.size(); ++h) {
        const Vx v = order[h];
        if (g_nd_prefetch >= 2 && h + 8 < order.size()) __builtin_prefetch(&c.Ap[order[h + 8]]);          // (see nd_bfs_team)
        if (g_nd_prefetch >= 2 && h + 4 < order.size()) __builtin_prefetch(&c.Ai[c.Ap[order[h + 4]]]);
        if (g_nd_prefetch >= 2 && h + 2 < order.size()) {
            const Vx u = order[h + 2];
            for (Long p = c.Ap[u]; p < c.Ap[u + 1]; ++p) __builtin_prefetch(&c.mark[c.Ai[p]]);
        }
        for (Long p = c.Ap[v]; p < c.Ap[v + 1]; ++p) {
            const Vx w = c.Ai[p];
            if (c.mk(w) == id) {
                c.set_mk(w, -2 - id);
                c.level[w] = c.level[v] + 1;
                order.push_back(w);
            }
        }
    }
    for (Vx v : order) c.set_mk(v, id);    // restore
}

void nd_component(NdCtx& c, std::vector<Vx>& comp, Vx pos);
void nd_recurse(NdCtx& c, std::vector<Vx>& verts, Vx pos, int team);

// ---- big pieces: the BFS itself is shared by a TEAM of threads -------------------------------------------------------------------
// At the top of the dissection there are 1, 2, 4, ... pieces: one thread per piece leaves the other threads idle exactly where the
// pieces are largest (the whole graph is swept three times by one thread).  Pieces of at least ND_PAR_MIN vertices therefore run a
// level-synchronous BFS in which the threads of a team split every frontier; a vertex is claimed with one compare-and-swap on its
// mark.  WHICH thread claims a vertex is a race, so the order inside a level is not reproducible -- and nothing below uses it: the
// level of every vertex, the level sets and the smallest vertex of the last level are the same for any team size, the restart root
// is that smallest vertex, and the vertex lists handed down are kept sorted by vertex number (a filter of a sorted list).  A piece
// takes this path or the sequential one by its SIZE alone, so the permutation does not depend on the number of threads.
constexpr Vx ND_PAR_MIN = 100000;

struct SpinBarrier {
    std::atomic<int> cnt{0}, gen{0};
    int n = 1;
    void wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (cnt.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { cnt.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_release); }
        else {
            // a BFS level of a grid is a few thousand vertices: the wait is usually shorter than a reschedule -- spin first, yield later
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == g) {
                if (++spins < 512) __builtin_ia32_pause(); else std::this_thread::yield();
            }
        }
    }
};

// BFS of piece `id` from `root` by `team` threads.  order: the visited vertices, level by level (capacity `cap` >= piece size);
// level[] filled; the marks of the visited vertices end as `final_mark`.  Returns the number of levels; *last_min = smallest vertex
// of the last level.
// Two barriers per BFS level (round 4; three before, and yielding ones: a 128^3 grid has ~380 levels of ~5,000 vertices, the sweep
// was bound by its synchronisation -- 47 ms for 2M vertices on 16 threads): every thread keeps its own copy of the frontier bounds and
// derives the next ones from the published per-thread counts (double-buffered), so nobody waits for a coordinator.
Vx nd_bfs_team(NdCtx& c, Vx id, Vx root, int team, Vx cap, Vx final_mark, std::vector<Vx>& order, Vx* last_min) {
    team = std::max(1, team);
    order.assign((size_t)cap, 0);
    const Vx vis = -2 - id;
    order[0] = root;
    c.level[root] = 0;
    c.set_mk(root, vis);
    std::vector<std::vector<Vx>> local((size_t)team);
    std::vector<Vx> sizes[2] = {std::vector<Vx>((size_t)team, 0), std::vector<Vx>((size_t)team, 0)};
    SpinBarrier bar;
    bar.n = team;
    Vx out_lo = 0, out_hi = 1, out_lev = 0;             // written by thread 0 when it leaves the loop
    auto body = [&](int t) {
        std::vector<Vx>& mine = local[(size_t)t];
        Vx lo = 0, hi = 1, lev = 0;                     // frontier = order[lo, hi): every thread's own, identical copy
        for (int par = 0;; par ^= 1) {
            const Vx F = hi - lo;
            const Vx a = lo + (Vx)((int64_t)F * t / team), b = lo + (Vx)((int64_t)F * (t + 1) / team);
            mine.clear();
            for (Vx h = a; h < b; ++h) {
                const Vx v = order[(size_t)h];
                // the sweep is a chain of cache misses (offsets of v, its adjacency, the marks of its neighbours): ask for the
                // offsets 16 vertices ahead, the adjacency 8 ahead and the neighbours' marks 4 ahead
                if (g_nd_prefetch >= 1 && h + 16 < b) __builtin_prefetch(&c.Ap[order[(size_t)h + 16]]);
                if (g_nd_prefetch >= 1 && h + 8 < b) __builtin_prefetch(&c.Ai[c.Ap[order[(size_t)h + 8]]]);
                if (g_nd_prefetch >= 1 && h + 4 < b) {
                    const Vx u = order[(size_t)h + 4];
                    for (Long p = c.Ap[u]; p < c.Ap[u + 1]; ++p) __builtin_prefetch(&c.mark[c.Ai[p]]);
                }
                for (Long p = c.Ap[v]; p < c.Ap[v + 1]; ++p) {
                    const Vx w = c.Ai[p];
                    if (c.mk(w) != id) continue;
                    Vx expect = id;
                    if (__atomic_compare_exchange_n(&c.mark[w], &expect, vis, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                        c.level[w] = lev + 1;
                        mine.push_back(w);
                    }
                }
            }
            sizes[par][(size_t)t] = (Vx)mine.size();
            bar.wait();                                 // all counts of this level are published
            Vx off = hi, total = 0;
            for (int q = 0; q < team; ++q) { if (q < t) off += sizes[par][(size_t)q]; total += sizes[par][(size_t)q]; }
            for (size_t k = 0; k < mine.size(); ++k) order[(size_t)off + k] = mine[k];
            if (total == 0) { if (t == 0) { out_lo = lo; out_hi = hi; out_lev = lev; } return; }
            lo = hi; hi += total; ++lev;
            bar.wait();                                 // the next frontier is complete in order[]
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < team; ++t) th.emplace_back(body, t);
    body(0);
    for (std::thread& x : th) x.join();
    const Vx lo = out_lo, hi = out_hi, lev = out_lev;
    order.resize((size_t)hi);
    Vx mn = order[(size_t)lo];
    for (Vx h = lo; h < hi; ++h) mn = std::min(mn, order[(size_t)h]);
    *last_min = mn;
    // marks: visited -> final_mark (split among the team)
    {
        const Vx N = hi;
        auto fin = [&](int t) { for (Vx h = (Vx)((int64_t)N * t / team); h < (Vx)((int64_t)N * (t + 1) / team); ++h) c.set_mk(order[(size_t)h], final_mark); };
        std::vector<std::thread> th2;
        for (int t = 1; t < team; ++t) th2.emplace_back(fin, t);
        fin(0);
        for (std::thread& x : th2) x.join();
    }
    return lev + 1;
}

// one connected big piece: `comp` sorted by vertex number, all marked with one id; `root`: where the level structure starts (the
// smallest vertex of the last level of the sweep that found the component = the second sweep of a pseudo-peripheral search)
void nd_component_big(NdCtx& c, std::vector<Vx>& comp, Vx pos, int team, Vx root) {
    const Vx id = c.mk(comp[0]);
    std::vector<Vx> order;
    Vx last_min = 0;
    const bool tr_big = getenv("SF_TRACE_ND") != nullptr;
    auto now_b = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tb0 = now_b();
    const Vx nlev = nd_bfs_team(c, id, root, team, (Vx)comp.size(), id, order, &last_min);
    { std::vector<Vx>().swap(order); }
    const double tb1 = now_b();
    if (nlev < 3) {         // (nearly) complete graph: no useful separator
        for (Vx v : comp) { c.out[pos++] = v; c.set_mk(v, -1); }
        return;
    }
    std::vector<Vx> cnt((size_t)nlev, 0);
    for (Vx v : comp) cnt[(size_t)c.level[v]]++;
    Vx best = 1, best_cost = -1, below = cnt[0], below0 = cnt[0];
    const Vx total = (Vx)comp.size();
    for (Vx l = 1; l + 1 < nlev; ++l) {
        const Vx above = total - below - cnt[(size_t)l];
        const Vx cost = std::max(below, above) + cnt[(size_t)l];
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = l; below0 = below; }
        below += cnt[(size_t)l];
    }
    std::vector<Vx> A, B, Sep;
    const Vx ida = c.next_id++, idb = c.next_id++;
    A.reserve((size_t)below0);
    B.reserve((size_t)(total - below0 - cnt[(size_t)best]));
    Sep.reserve((size_t)cnt[(size_t)best]);
    for (Vx v : comp) {                                   // sorted in, sorted out
        if (c.level[v] < best) { A.push_back(v); c.set_mk(v, ida); }
        else if (c.level[v] > best) { B.push_back(v); c.set_mk(v, idb); }
        else Sep.push_back(v);
    }
    for (Vx v : Sep) c.set_mk(v, -1);
    if (tr_big) fprintf(stderr, "[sparseframe-hip]     big piece of %zu vertices, team %d: level structure %.1f ms, count + split %.1f ms\n",
                        comp.size(), team, tb1 - tb0, now_b() - tb1);
    { std::vector<Vx>().swap(comp); }
    const Vx posA = pos, posB = pos + (Vx)A.size(), posS = posB + (Vx)B.size();
    const int team_a = std::max(1, team / 2), team_b = std::max(1, team - team_a);
    std::thread helper;
    if (!A.empty() && !B.empty() && team >= 2) {
        c.helpers.fetch_add(1);
        helper = std::thread([&c, &A, posA, team_a] { nd_recurse(c, A, posA, team_a); });
    } else if (!A.empty()) nd_recurse(c, A, posA, team);
    if (!B.empty()) nd_recurse(c, B, posB, helper.joinable() ? team_b : team);
    if (helper.joinable()) { helper.join(); c.helpers.fetch_sub(1); }
    Vx q = posS;
    for (Vx v : Sep) c.out[q++] = v;
}


// `verts`: vertices carrying one common mark (sorted by vertex number when the piece is big), to be written to out[pos ..).  Its
// connected components are found and dissected one after the other (iteratively: a diagonal matrix has n components).
//   * a big piece is usually ONE component (a dissection half of a mesh): the team's BFS from its smallest vertex finds that out and
//     the piece goes to nd_component_big as it is;
//   * otherwise everything that is left is labelled in ONE sequential sweep -- a BFS per component from its smallest vertex, each
//     vertex visited once -- and only the components that are big themselves go back to the team.  (Round 3 ran the team's BFS, a
//     whole-piece filter and 2 (team - 1) thread starts PER COMPONENT: a piece of 10^5 vertices with many small components -- the
//     identity rows of a finite-element matrix, a diagonal matrix -- went quadratic: n = 104,000 diagonal 34 s)
// Which path a component takes depends on sizes and vertex numbers only, never on the number of threads.
void nd_recurse(NdCtx& c, std::vector<Vx>& verts, Vx pos, int team) {
    const Vx id = c.mk(verts[0]);
    if ((Vx)verts.size() >= ND_PAR_MIN) {
        std::vector<Vx> order;
        Vx last_min = 0;
        const Vx cid = c.next_id++;
        (void)nd_bfs_team(c, id, verts[0], team, (Vx)verts.size(), cid, order, &last_min);
        const Vx found = (Vx)order.size();
        { std::vector<Vx>().swap(order); }
        if (found == (Vx)verts.size()) { nd_component_big(c, verts, pos, team, last_min); return; }
        std::vector<Vx> comp;
        comp.reserve((size_t)found);
        for (Vx v : verts)
            if (c.mk(v) == cid) comp.push_back(v);              // sorted in, sorted out
        if (found >= ND_PAR_MIN) nd_component_big(c, comp, pos, team, last_min);
        else {
            // the sequential code wants the component in BFS order from some vertex of it (the team's order inside a level is a race)
            std::vector<Vx> o2;
            nd_bfs(c, cid, comp[0], o2);
            nd_component(c, o2, pos);
        }
        pos += found;
    }
    std::vector<Vx> comp;
    for (Vx v : verts) {
        if (c.mk(v) != id) continue;          // already ordered as part of an earlier component
        nd_bfs(c, id, v, comp);
        const Vx cid = c.next_id++;
        for (Vx w : comp) c.set_mk(w, cid);
        const Vx sz = (Vx)comp.size();
        if (sz >= ND_PAR_MIN) {
            // a big component behind smaller ones: level structure from the smallest vertex of the last level of this sweep (what the
            // team's sweep would have reported), vertex list sorted by a filter of the piece's
            const Vx last = c.level[comp.back()];
            Vx root = comp.back();
            for (size_t k = comp.size(); k-- > 0 && c.level[comp[k]] == last;) root = std::min(root, comp[k]);
            std::vector<Vx> sorted;
            sorted.reserve((size_t)sz);
            for (Vx w : verts)
                if (c.mk(w) == cid) sorted.push_back(w);
            { std::vector<Vx>().swap(comp); }
            nd_component_big(c, sorted, pos, team, root);
        } else {
            nd_component(c, comp, pos);
        }
        pos += sz;
    }
}

// one connected piece, all vertices marked with one id; `comp` is in BFS order from some vertex of it; output range out[pos, pos + |comp|)
void nd_component(NdCtx& c, std::vector<Vx>& comp, Vx pos) {
    std::vector<Vx>& verts = comp;
    const Vx id = c.mk(verts[0]);
    if ((Vx)verts.size() <= c.leaf) {
        // BFS order from a pseudo-peripheral vertex, reversed
        std::vector<Vx> o2;
        nd_bfs(c, id, comp.back(), o2);
        for (size_t k = o2.size(); k-- > 0;) { c.out[pos++] = o2[k]; c.set_mk(o2[k], -1); }
        return;
    }
    // pseudo-peripheral root: restart the BFS from the last vertex of the previous one
    std::vector<Vx> order;
    nd_bfs(c, id, comp.back(), order);
    const Vx nlev = c.level[order.back()] + 1;
    if (nlev < 3) {     // (nearly) complete graph: no useful separator
        for (size_t k = order.size(); k-- > 0;) { c.out[pos++] = order[k]; c.set_mk(order[k], -1); }
        return;
    }
    std::vector<Vx> cnt(nlev, 0);
    for (Vx v : order) cnt[c.level[v]]++;
    // separator level: minimise max(|below|, |above|) + |level| over interior levels
    Vx best = 1, best_cost = -1, below = cnt[0];
    const Vx total = (Vx)order.size();
    for (Vx l = 1; l + 1 < nlev; ++l) {
        const Vx above = total - below - cnt[l];
        const Vx cost = std::max(below, above) + cnt[l];
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = l; }
        below += cnt[l];
    }
    std::vector<Vx> A, B, Sep;
    const Vx ida = c.next_id++, idb = c.next_id++;
    for (Vx v : order) {
        if (c.level[v] < best) { A.push_back(v); c.set_mk(v, ida); }
        else if (c.level[v] > best) { B.push_back(v); c.set_mk(v, idb); }
        else Sep.push_back(v);
    }
    for (Vx v : Sep) c.set_mk(v, -1);       // removed from the graph for the recursion
    { std::vector<Vx>().swap(order); std::vector<Vx>().swap(comp); }
    const Vx posA = pos, posB = pos + (Vx)A.size(), posS = posB + (Vx)B.size();
    // A (the levels below the separator) is CONNECTED -- every vertex hangs on the BFS tree of the root -- and already listed in BFS order
    // from that root: exactly what nd_recurse's component sweep from A[0] would find and hand to nd_component, so it goes there at once
    // (the same permutation as before, one sweep over half of every piece less).  B may fall apart: it is swept.
    std::thread helper;
    if (!A.empty() && !B.empty() && (Vx)A.size() >= 20000 && c.helpers.fetch_add(1) < c.max_helpers)
        helper = std::thread([&c, &A, posA] { nd_component(c, A, posA); });
    else {
        if (!A.empty() && !B.empty() && (Vx)A.size() >= 20000) c.helpers.fetch_sub(1);      // no free helper: undo the claim
        if (!A.empty()) nd_component(c, A, posA);
    }
    if (!B.empty()) nd_recurse(c, B, posB, 1);
    if (helper.joinable()) { helper.join(); c.helpers.fetch_sub(1); }
    Vx q = posS;
    for (Vx v : Sep) c.out[q++] = v;
}
}  // namespace

// Cp/Ci: any triangle (or both) of the symmetric pattern; the pattern is symmetrised.  perm[new] = old.
int graph_nd_perm(Long n, const Long* Cp, const Long* Ci, Long leaf, Long* perm) {
    if (n < 0 || !Cp || (n > 0 && !Ci) || !perm || leaf < 1) return 1;
    NodeAffinity on_one_node;
    const bool tr_nd = getenv("SF_TRACE") != nullptr;
    auto now_nd = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_nd0 = now_nd();
    if (n >= ((Long)1 << 30)) return 1;          // 32-bit vertex and piece ids inside the dissection
    std::vector<Long> Ap(n + 1, 0);
    std::vector<Vx> Ai;
    for (Long j = 0; j < n; ++j)
        for (Long p = Cp[j]; p < Cp[j + 1]; ++p) {
            const Long i = Ci[p];
            if (i < 0 || i >= n) return 1;
            if (i != j) { Ap[i + 1]++; Ap[j + 1]++; }
        }
    for (Long j = 0; j < n; ++j) Ap[j + 1] += Ap[j];
    Ai.resize(Ap[n]);
    {
        std::vector<Long> fill(Ap.begin(), Ap.end() - 1);
        for (Long j = 0; j < n; ++j)
            for (Long p = Cp[j]; p < Cp[j + 1]; ++p) {
                const Long i = Ci[p];
                if (i != j) { Ai[fill[i]++] = (Vx)j; Ai[fill[j]++] = (Vx)i; }
            }
    }
    const double t_nd1 = now_nd();
    NdCtx c{Ap, Ai, std::vector<Vx>(n, 0), std::vector<Vx>(n, 0), perm, (Vx)std::min<Long>(leaf, n)};
    c.max_helpers = analysis_threads() - 1;
    if (n == 0) return 0;
    std::vector<Vx> all(n);
    for (Long v = 0; v < n; ++v) all[v] = (Vx)v;
    for (Long v = 0; v < n; ++v) perm[v] = -1;
    nd_recurse(c, all, 0, analysis_threads());    // recursion depth = dissection depth, O(log n) for balanced level separators
    if (tr_nd) fprintf(stderr, "[sparseframe-hip]   ordering: adjacency %.1f ms, dissection %.1f ms (%d threads)\n", t_nd1 - t_nd0, now_nd() - t_nd1, analysis_threads());
    for (Long v = 0; v < n; ++v)
        if (perm[v] < 0) return 2;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// geometric nested dissection on a regular grid
// ---------------------------------------------------------------------------------------------
namespace {
struct Box { Long lo[3], hi[3]; };

void nd_emit(const Box& b, Long nx, Long ny, Long*& out) {
    for (Long z = b.lo[2]; z < b.hi[2]; ++z)
        for (Long y = b.lo[1]; y < b.hi[1]; ++y)
            for (Long x = b.lo[0]; x < b.hi[0]; ++x) *out++ = x + nx * (y + ny * z);
}

void nd_rec(const Box& b, Long nx, Long ny, Long leaf, Long sepw, Long*& out) {
    Long len[3];
    for (int a = 0; a < 3; ++a) len[a] = b.hi[a] - b.lo[a];
    if (len[0] <= 0 || len[1] <= 0 || len[2] <= 0) return;
    int ax = 0;
    for (int a = 1; a < 3; ++a)
        if (len[a] > len[ax]) ax = a;  // longest axis, ties -> lowest axis index
    if (len[ax] <= leaf || len[ax] < sepw + 2) { nd_emit(b, nx, ny, out); return; }
    const Long mid = b.lo[ax] + (len[ax] - sepw + 1) / 2;
    Box l = b, r = b, s = b;
    l.hi[ax] = mid;
    s.lo[ax] = mid; s.hi[ax] = mid + sepw;
    r.lo[ax] = mid + sepw;
    nd_rec(l, nx, ny, leaf, sepw, out);
    nd_rec(r, nx, ny, leaf, sepw, out);
    nd_emit(s, nx, ny, out);
}
}  // namespace

int grid_nd_perm(Long nx, Long ny, Long nz, Long leaf, Long sepw, Long* perm) {
    if (nx <= 0 || ny <= 0 || nz <= 0 || leaf <= 0 || sepw <= 0 || !perm) return 1;
    Box b{{0, 0, 0}, {nx, ny, nz}};
    Long* out = perm;
    nd_rec(b, nx, ny, leaf, sepw, out);
    return (out - perm) == nx * ny * nz ? 0 : 2;
}

}  // namespace sf
