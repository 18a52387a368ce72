// Device handlers behind the reference's struct entry points (SparseFrame_allocate_gpu C:16-285, _free_gpu C:287-366,
// _factorize_supernodal C:2150-3017, _factorize C:3019-3034; the LU library forwards here too).
//
// The reference builds, per device handler, eight equal device slots plus pinned mirrors and streams, and stages every
// panel through them.  Here a handler owns
//   * a lock (the reference's gpuLock, C:2287: MATRIX_THREAD_NUM callers share the handler list),
//   * a small cache of device plans keyed by a hash of the symbolic pattern: the task tables, the relative maps and the
//     resident factor are built once per pattern and re-used by every later factorization of that pattern,
// and a factorization is: values H2D, the level-scheduled numeric phase, and the factor copied back into
// matrix_info->Lsx WHILE the upper levels still compute (sf_chol_plan_factorize_to_host; reference C:2888-2895).
#include <sparseframe_hip.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "sf_kernels.h"
#include "sf_plan_internal.h"
#include "sf_symbolic.h"

int sf_comm_create_all(sf_comm** comms, int nranks, const int* devices);     // sf_multi.hip

namespace {

struct PlanKey {
    uint64_t h = 0;
    int64_t n = 0, nsuper = 0, isize = 0, xsize = 0, nnz = 0, unz = 0;
    int lu = 0;
    bool operator==(const PlanKey& o) const {
        return h == o.h && n == o.n && nsuper == o.nsuper && isize == o.isize && xsize == o.xsize && nnz == o.nnz && unz == o.unz && lu == o.lu;
    }
};

// 64-bit multiply-xorshift hash over 8-byte words, four independent lanes (memory-bound: ~10 GB/s per thread)
uint64_t hash_words(const int64_t* a, int64_t n, uint64_t seed) {
    const uint64_t M = 0x9E3779B97F4A7C15ull;
    uint64_t h0 = seed ^ 0x1234567ull, h1 = seed + 0x9abcdefull, h2 = ~seed, h3 = seed * M;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        h0 = (h0 ^ (uint64_t)a[i]) * M;     h0 ^= h0 >> 29;
        h1 = (h1 ^ (uint64_t)a[i + 1]) * M; h1 ^= h1 >> 31;
        h2 = (h2 ^ (uint64_t)a[i + 2]) * M; h2 ^= h2 >> 27;
        h3 = (h3 ^ (uint64_t)a[i + 3]) * M; h3 ^= h3 >> 33;
    }
    for (; i < n; ++i) { h0 = (h0 ^ (uint64_t)a[i]) * M; h0 ^= h0 >> 29; }
    uint64_t h = h0;
    h = (h ^ h1) * M; h ^= h >> 32;
    h = (h ^ h2) * M; h ^= h >> 32;
    h = (h ^ h3) * M; h ^= h >> 32;
    return h;
}

// the arrays a plan depends on: Super, Lsip, Lsxp, Lsi (the symbolic factor) and Lp, Li (Up, Ui) (where the matrix entries go);
// the two long ones (Lsi, Li) are hashed by their own threads
PlanKey make_key(int lu, sf_long n, sf_long nsuper, const sf_long* Super, const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                 const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui) {
    PlanKey k;
    k.lu = lu; k.n = n; k.nsuper = nsuper; k.isize = Lsip[nsuper]; k.xsize = Lsxp[nsuper]; k.nnz = Lp[n];
    k.unz = (lu && Up) ? Up[n] : 0;
    uint64_t hl = 0, hi = 0, hu = 0;
    std::thread t1([&] { hl = hash_words(Lsi, k.isize, 11); });
    std::thread t2([&] { hi = hash_words(Li, k.nnz, 13); });
    std::thread t3([&] { if (k.unz) hu = hash_words(Ui, k.unz, 17) ^ hash_words(Up, n + 1, 19); });
    uint64_t h = hash_words(Super, nsuper + 1, 1);
    h ^= hash_words(Lsip, nsuper + 1, 3) * 3;
    h ^= hash_words(Lsxp, nsuper + 1, 5) * 5;
    h ^= hash_words(Lp, n + 1, 7) * 7;
    t1.join(); t2.join(); t3.join();
    k.h = h ^ (hl * 11) ^ (hi * 13) ^ (hu * 17);
    return k;
}

struct HandlerState;
// whole plan a distributed factor is gathered into for the struct path's solve (owned by a MultiState cache entry and by the
// registry entries that point at it)
struct SolverSlot {
    sf_chol_plan* plan = nullptr;
    std::vector<int> imported;          // epochs of the parts the plan currently holds
    ~SolverSlot() { if (plan) sf_chol_plan_destroy(plan); }
};
// host copies of factors that are still on the device(s): in a single handler's cached plan (plan), or distributed over the
// per-rank plans of a multi-handler factorization (parts), see sf_handlers_solve_resident
struct Resident {
    sf_chol_plan* plan = nullptr;
    int epoch = 0;
    std::vector<sf_chol_plan*> parts;
    std::vector<int> part_epochs;
    std::shared_ptr<SolverSlot> slot;
    std::vector<sf_comm*> comms;        // the ranks' communicators (owned by the handler list), for the distributed solve
    int device = 0;
};
// (leaked on purpose: a process that exits with entries still registered -- no SparseFrame_free_gpu -- must not run plan destructors,
// i.e. HIP calls, from a static destructor after the runtime has been torn down)
std::mutex& g_res_mu = *new std::mutex();
std::unordered_map<const void*, Resident>& g_resident = *new std::unordered_map<const void*, Resident>();
int64_t g_resident_solves = 0;      // solves served from a resident factor (tests)
std::atomic<int64_t> g_fingerprint_fallbacks{0};   // verified solves that fell back to the host sweep because a fingerprint differed (sf_handlers_fingerprint_fallbacks)
// LU pivoting policy of the struct entry points (sf_handlers_set_lu_pivoting): unset = whatever the plans were created with (the
// reference's behaviour, no pivoting, unless SF_LU_PIVOT_TOL said otherwise); g_perturbed: perturbed pivots of the factorization that filled a given host array
std::mutex& g_piv_mu = *new std::mutex();
bool g_piv_set = false;
double g_piv_tol = 0.0, g_piv_perturb = 0.0;
std::unordered_map<const void*, int64_t>& g_perturbed = *new std::unordered_map<const void*, int64_t>();
// the policy of ONE call (sf_handlers_set_lu_pivoting_next_call: the struct library passes a matrix_info's own setting this way);
// read by the thread that calls sf_handlers_factorize, before it hands the plans to the handlers' threads
thread_local bool t_piv_set = false;
thread_local double t_piv_tol = 0.0, t_piv_perturb = 0.0;
// Every LU factorization through the handlers states its policy: the call's own, else the process-wide one, else what the plan was
// created with -- a cached plan must not carry the previous matrix's setting into the next one's factorization.
int apply_pivot_policy(sf_chol_plan* plan) {
    if (!plan || !plan->lu) return SF_OK;
    if (t_piv_set) return sf_lu_plan_set_pivoting(plan, t_piv_tol, t_piv_perturb);
    std::lock_guard<std::mutex> g(g_piv_mu);
    return g_piv_set ? sf_lu_plan_set_pivoting(plan, g_piv_tol, g_piv_perturb)
                     : sf_lu_plan_set_pivoting(plan, plan->piv_tol0, plan->piv_perturb0);
}
void note_perturbed(const void* Lsx, int64_t count) {
    std::lock_guard<std::mutex> g(g_piv_mu);
    if (g_perturbed.size() > 4096) g_perturbed.clear();
    g_perturbed[Lsx] = count;
}
// resident-solve policy (sf_handlers_set_resident_solve): 0 never, 1 after a FULL comparison of fingerprints (default), 2 trusted
std::atomic<int> g_resident_mode{1};

// The fingerprint of k_factor_hash over the caller's host array: H[s] = sum of bits(v_e) * (2 e + 1) * K over the values of panel s.
// Worker threads pull 32 MiB chunks from a counter and walk them with the supernode of the current value (one multiply per word:
// memory-bound); the per-chunk partial sums are merged under a lock.
void host_panel_hashes(const double* Lsx, const int64_t* Lsxp, int64_t nsuper, std::vector<uint64_t>& out) {
    out.assign((size_t)std::max<int64_t>(nsuper, 1), 0);
    const int64_t total = nsuper > 0 ? Lsxp[nsuper] : 0;
    if (total <= 0) return;
    const int64_t CH = (int64_t)4 << 20;
    const int64_t nch = (total + CH - 1) / CH;
    int T = (int)std::min<int64_t>(nch, std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    if (const char* e = getenv("SF_VERIFY_THREADS")) T = std::max(1, std::min(64, atoi(e)));
    std::atomic<int64_t> next{0};
    std::mutex mu;
    auto work = [&] {
        std::vector<std::pair<int64_t, uint64_t>> loc;
        const uint64_t K = 0x9E3779B97F4A7C15ull;
        for (int64_t c = next.fetch_add(1); c < nch; c = next.fetch_add(1)) {
            const int64_t b = c * CH, e_end = std::min(total, b + CH);
            int64_t s = std::upper_bound(Lsxp, Lsxp + nsuper + 1, b) - Lsxp - 1;
            int64_t e = b;
            while (e < e_end) {
                while (e >= Lsxp[s + 1]) ++s;
                const int64_t stop = std::min(e_end, Lsxp[s + 1]);
                uint64_t acc = 0, m = (2ull * (uint64_t)e + 1ull) * K;
                typedef uint64_t __attribute__((may_alias)) bits64;            // the doubles' bit patterns
                const bits64* w = reinterpret_cast<const bits64*>(Lsx);
                for (; e < stop; ++e, m += 2ull * K) acc += w[e] * m;
                if (acc) loc.emplace_back(s, acc);
            }
        }
        std::lock_guard<std::mutex> g(mu);
        for (auto& pr : loc) out[(size_t)pr.first] += pr.second;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
}

void forget_plan(sf_chol_plan* plan) {
    std::lock_guard<std::mutex> g(g_res_mu);
    for (auto it = g_resident.begin(); it != g_resident.end();) {
        bool hit = it->second.plan == plan;
        for (sf_chol_plan* q : it->second.parts) hit = hit || q == plan;
        if (hit) it = g_resident.erase(it); else ++it;
    }
}
void destroy_plan(sf_chol_plan* plan) {
    if (!plan) return;
    forget_plan(plan);
    sf_chol_plan_destroy(plan);
}

// all handlers of a list working on ONE matrix (the reference runs numGPU + numCPU workers inside
// SparseFrame_factorize_supernodal, C:2267): communicators and the per-rank distributed plans, hung off handler 0
struct MultiState {
    std::mutex mu;
    std::vector<sf_comm*> comms;
    struct Entry { PlanKey key; std::vector<sf_chol_plan*> plans; uint64_t stamp; std::shared_ptr<SolverSlot> slot; };
    std::vector<Entry> cache;
    uint64_t clock = 0;
    uint64_t builds = 0;        // sets of per-rank plans built (cache misses)
    ~MultiState() {
        for (Entry& e : cache)
            for (sf_chol_plan* p : e.plans) destroy_plan(p);
        for (sf_comm* c : comms) sf_comm_destroy(c);
    }
};

struct HandlerState {
    std::mutex mu;
    struct Entry { PlanKey key; sf_chol_plan* plan; uint64_t stamp; };
    std::vector<Entry> cache;
    uint64_t clock = 0;
    uint64_t builds = 0;                // plans built by this handler (cache misses)
    MultiState* multi = nullptr;        // handler 0 only
    // Device pool: ONE buffer allocated in SparseFrame_allocate_gpu (where the reference allocates its eight slots, C:92-283) that
    // the handler lends to the plan of one pattern at a time for its factor -- so that the first SparseFrame_factorize of a pattern
    // does not start with a hipMalloc of tens of GB (12 - 790 ms, depending on what the box did before).  A factor that does not
    // fit the pool, or a second cached pattern, allocates as before; a multi-handler factorization gives the pools back first.
    int device = 0;
    double* pool = nullptr;
    size_t pool_bytes = 0;
    sf_chol_plan* pool_user = nullptr;
    void release_pool() {
        if (pool && !pool_user) { (void)hipSetDevice(device); (void)hipFree(pool); pool = nullptr; pool_bytes = 0; }
    }
    ~HandlerState() {
        for (Entry& e : cache) destroy_plan(e.plan);
        pool_user = nullptr;
        release_pool();
        delete multi;
    }
};
constexpr size_t MAX_CACHED_PLANS = 2;      // per handler: MATRIX_THREAD_NUM = 2 matrices in flight in the reference's driver

}  // namespace

struct gpu_info_struct {
    int gpuIndex_physical;
    size_t devMemSize;
    HandlerState* st;
};

// ONE factorization over all handlers of the list: rank r = handler r.  Every rank thread builds (or finds) its
// distributed plan, uploads the values and runs sf_chol_plan_factorize_distributed with the overlapped copy-back of its
// own pieces into Lsx_out (owned subtree panels; the top panels' pieces are dealt out over the ranks).
static int factorize_all_handlers(struct common_info_struct* common, struct gpu_info_struct* list, int lu,
                                  sf_long n, sf_long nsuper, const sf_long* Super, const sf_long* SuperMap,
                                  const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                                  const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui,
                                  const sf_float* Lx, const sf_float* Ux, sf_float* Lsx_out, sf_long* PivOut) {
    const int N = common->numGPU;
    MultiState& M = *list[0].st->multi;
    PlanKey key = make_key(lu, n, nsuper, Super, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui);
    key.h ^= 0x5bd1e995ull * (uint64_t)N;
    std::lock_guard<std::mutex> guard(M.mu);
    if (M.comms.empty()) {
        std::vector<int> devs(N);
        for (int r = 0; r < N; ++r) devs[r] = list[r].gpuIndex_physical;
        M.comms.assign(N, nullptr);
        const int rc = sf_comm_create_all(M.comms.data(), N, devs.data());
        if (rc) { M.comms.clear(); return rc; }
    }
    MultiState::Entry* entry = nullptr;
    for (MultiState::Entry& e : M.cache)
        if (e.key == key) { entry = &e; e.stamp = ++M.clock; break; }
    std::vector<int32_t> owner;
    if (!entry) {
        while (M.cache.size() >= MAX_CACHED_PLANS) {
            size_t lru = 0;
            for (size_t i = 1; i < M.cache.size(); ++i)
                if (M.cache[i].stamp < M.cache[lru].stamp) lru = i;
            for (sf_chol_plan* p : M.cache[lru].plans) destroy_plan(p);
            M.cache.erase(M.cache.begin() + lru);
        }
        owner.assign(nsuper, 0);
        // cost of a top flop relative to a subtree flop: its 1/N share plus the replicated chain and the all-reduce
        if (sf::subtree_partition(nsuper, Super, SuperMap, Lsip, Lsi, N, owner.data(), nullptr, nullptr, 1.0 / N + 0.25)) return SF_ERR_ARG;
        M.cache.push_back(MultiState::Entry{key, std::vector<sf_chol_plan*>(N, nullptr), ++M.clock, std::make_shared<SolverSlot>()});
        entry = &M.cache.back();
        ++M.builds;
    }
    std::vector<int> rcs(N, SF_OK);
    std::vector<std::thread> th;
    for (int r = 0; r < N; ++r)
        th.emplace_back([&, r] {
            int rc = SF_OK;
            sf_chol_plan*& plan = entry->plans[r];
            if (!plan)      // proportional mapping: own subtrees + the top supernodes above them
                rc = lu ? sf_lu_plan_create_mapped(&plan, list[r].gpuIndex_physical, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp,
                                                   Lp, Li, Up, Ui, owner.data(), r, N)
                        : sf_chol_plan_create_mapped(&plan, list[r].gpuIndex_physical, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp,
                                                     Lp, Li, owner.data(), r, N);
            if (!rc) rc = lu ? sf_lu_plan_set_values(plan, Lx, Ux) : sf_chol_plan_set_values(plan, Lx);
            // a rank without a usable plan cannot take part in the collectives; the others would wait for it, so plan
            // creation is checked by all ranks before any of them starts (below)
            rcs[r] = rc;
        });
    for (std::thread& t : th) t.join();
    th.clear();
    for (int r = 0; r < N; ++r)
        if (rcs[r]) {
            for (sf_chol_plan*& p : entry->plans) { destroy_plan(p); p = nullptr; }
            M.cache.erase(M.cache.begin() + (entry - M.cache.data()));
            return rcs[r];
        }
    for (int r = 0; r < N; ++r) forget_plan(entry->plans[r]);       // the host copies these plans' factors stood for are history
    for (int r = 0; r < N; ++r)
        if (int rc = apply_pivot_policy(entry->plans[r])) return rc;
    for (int r = 0; r < N; ++r)
        th.emplace_back([&, r] { rcs[r] = sf_chol_plan_factorize_distributed(entry->plans[r], M.comms[r], Lsx_out, 1); });
    for (std::thread& t : th) t.join();
    for (int r = 0; r < N; ++r)
        if (rcs[r]) return rcs[r];
    if (lu && PivOut)      // every rank reports the blocks of the panels it stores (the top panels' records agree on all ranks)
        for (int r = 0; r < N; ++r)
            if (int rc = sf_lu_plan_get_pivots(entry->plans[r], PivOut)) return rc;
    if (lu) {              // (a shared top panel's perturbations are counted by every rank of its group: an upper bound, 0 is exact)
        int64_t per = 0;
        for (int r = 0; r < N; ++r) per += entry->plans[r]->last_perturbed;
        note_perturbed(Lsx_out, per);
    }
    {
        // the factor is now spread over the ranks' plans; a solve of the struct path may gather it into a whole plan on the first
        // handler's device (sf_handlers_solve_resident)
        Resident R;
        R.parts = entry->plans;
        for (sf_chol_plan* q : entry->plans) R.part_epochs.push_back(q->epoch);
        R.slot = entry->slot;
        R.comms = M.comms;
        R.device = list[0].gpuIndex_physical;
        std::lock_guard<std::mutex> g(g_res_mu);
        g_resident[(const void*)Lsx_out] = std::move(R);
    }
    return SF_OK;
}

extern "C" {

int sf_handlers_allocate(struct common_info_struct* common, struct gpu_info_struct** list) {
    if (!common || !list) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    const int nphys = sf_device_count();
    int ndev = nphys;
    // SF_EMULATE_HANDLERS=N on a one-GPU box: N handlers that share device 0 (the multi-handler code path end to end,
    // with the all-reduce done by a kernel on that device instead of RCCL) -- tests and rehearsals only
    if (nphys == 1)
        if (const char* env = getenv("SF_EMULATE_HANDLERS")) ndev = std::max(1, std::min(16, atoi(env)));
    common->numGPU_physical = nphys;
    common->numGPU = ndev;       // one handler per device; no virtual-GPU splitting (reference C:36-41)
    common->numCPU = 0;          // the numeric phase has no CPU worker
    common->minDevMemSize = 0;
    common->minHostMemSize = 0;
    *list = (struct gpu_info_struct*)calloc(ndev > 0 ? ndev : 1, sizeof(struct gpu_info_struct));
    if (!*list) return 1;
    size_t min_mem = (size_t)-1;
    for (int d = 0; d < ndev; ++d) {
        hipDeviceProp_t prop;
        const int phys = nphys == 1 ? 0 : d;
        if (hipGetDeviceProperties(&prop, phys) != hipSuccess) continue;
        (*list)[d].gpuIndex_physical = phys;
        (*list)[d].devMemSize = prop.totalGlobalMem;
        (*list)[d].st = new (std::nothrow) HandlerState();
        min_mem = std::min(min_mem, (size_t)prop.totalGlobalMem);
        // pay the per-process, per-device one-time costs here (the reference creates its streams and cuBLAS / cuSOLVER handles in
        // this call, C:92-283) instead of inside the first SparseFrame_factorize: runtime and code-object initialisation, the first
        // device and pinned allocations
        if (d == phys && hipSetDevice(phys) == hipSuccess) {
            void* dp = nullptr;
            void* hp = nullptr;
            if (hipMalloc(&dp, 1 << 20) == hipSuccess) (void)hipFree(dp);
            if (hipHostMalloc(&hp, 1 << 20, hipHostMallocDefault) == hipSuccess) (void)hipHostFree(hp);
            sf::launch_noop(nullptr);
            (void)hipDeviceSynchronize();
            (void)hipGetLastError();
        }
    }
    if (ndev > 0 && (*list)[0].st) (*list)[0].st->multi = new (std::nothrow) MultiState();
    if (ndev > 0 && min_mem != (size_t)-1) {
        common->devSlotSize = sf_reference_slot_size(ndev, min_mem);
        common->minDevMemSize = common->devSlotSize * 8;
        // the pool: what the reference's eight slots would take (at most a quarter of the device); SF_DEVICE_POOL_MB overrides, 0 = none.
        // One handler per physical device only (not for emulated handlers sharing a device), and only where one handler = one matrix:
        // with several handlers the default is ONE matrix over all of them, whose per-rank plans need the memory themselves.
        size_t want = std::min(common->devSlotSize * 8, min_mem / 4);
        if (const char* env = getenv("SF_DEVICE_POOL_MB")) want = (size_t)strtoull(env, nullptr, 10) << 20;
        const char* mode = getenv("SF_MULTI");
        const bool per_matrix = ndev == 1 || (mode && strcmp(mode, "matrix") == 0);
        for (int d = 0; d < ndev && want > 0 && ndev == nphys && per_matrix; ++d) {
            HandlerState* st = (*list)[d].st;
            if (!st) continue;
            st->device = (*list)[d].gpuIndex_physical;
            void* q = nullptr;
            if (hipSetDevice(st->device) == hipSuccess && hipMalloc(&q, want) == hipSuccess) { st->pool = (double*)q; st->pool_bytes = want; }
            else (void)hipGetLastError();
        }
    } else {
        const char* env = getenv("SF_DEVSLOT");
        common->devSlotSize = env ? (size_t)strtoull(env, nullptr, 10) : ((size_t)1 << 30);
    }
    common->allocateTime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// number of device plans (sets of per-rank plans for a multi-handler factorization) the handlers of a list have built so far:
// a repeated sparsity pattern must not add to it (tests; SF_TRACE prints the same per call)
int64_t sf_handlers_plan_builds(struct gpu_info_struct* list, int n_handlers) {
    int64_t total = 0;
    if (!list) return 0;
    for (int d = 0; d < n_handlers; ++d) {
        if (!list[d].st) continue;
        std::lock_guard<std::mutex> guard(list[d].st->mu);
        total += (int64_t)list[d].st->builds;
        if (list[d].st->multi) total += (int64_t)list[d].st->multi->builds;
    }
    return total;
}

// the device pool of handler d: out[0] = bytes (0: none), out[1] = 1 while a plan's factor lives in it, out[2] = that plan's n
int sf_handlers_pool_info(struct gpu_info_struct* list, int d, sf_long* out) {
    if (!list || d < 0 || !out || !list[d].st) return SF_ERR_ARG;
    std::lock_guard<std::mutex> guard(list[d].st->mu);
    out[0] = (sf_long)list[d].st->pool_bytes;
    out[1] = list[d].st->pool_user ? 1 : 0;
    out[2] = list[d].st->pool_user ? list[d].st->pool_user->n : 0;
    return SF_OK;
}

int sf_handlers_free(struct common_info_struct* common, struct gpu_info_struct** list) {
    if (!list || !*list) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    const int nh = common ? std::max(common->numGPU, 0) : 0;
    for (int d = 0; d < nh; ++d) {
        delete (*list)[d].st;           // destroys the cached plans (device memory, pinned rings)
        (*list)[d].st = nullptr;
    }
    free(*list);
    *list = nullptr;
    if (common) {
        common->numGPU = 0;
        common->freeTime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

// One numeric factorization through the handler list.  Returns SF_OK or an SF_ERR_* code; Lsx_out receives the factor in
// the reference layout.
int sf_handlers_factorize(struct common_info_struct* common, struct gpu_info_struct* list, int lu, int serial,
                          sf_long n, sf_long nsuper, const sf_long* Super, const sf_long* SuperMap,
                          const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                          const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui,
                          const sf_float* Lx, const sf_float* Ux, sf_float* Lsx_out, sf_long* PivOut) {
    struct OneCallPolicy { ~OneCallPolicy() { t_piv_set = false; } } one_call_policy;      // sf_handlers_set_lu_pivoting_next_call holds for THIS call only
    if (!common || !Lsx_out || !Super || !Lsip || !Lsxp || !Lp || n < 0 || nsuper < 0) return SF_ERR_ARG;
    if (common->numGPU <= 0 || !list) {
        fprintf(stderr, "[sparseframe-hip] SparseFrame_factorize: no GPU handler (numGPU = %d); no CPU fallback\n", common->numGPU);
        return SF_ERR_NO_DEVICE;
    }
    if (n > 0 && (!SuperMap || !Lsi || !Li)) return SF_ERR_ARG;
    // Several handlers: all of them factorize this matrix together, as in the reference (C:2267) -- elimination-tree
    // subtrees per handler, RCCL for the parent-front merge.  SF_MULTI=matrix keeps one matrix per handler instead.
    if (common->numGPU > 1 && nsuper > 0 && list[0].st && list[0].st->multi) {
        const char* mode = getenv("SF_MULTI");
        if (!mode || strcmp(mode, "matrix") != 0)
            return factorize_all_handlers(common, list, lu, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui, Lx, Ux, Lsx_out, PivOut);
    }
    // one matrix = one handler; the caller's matrix threads (MATRIX_THREAD_NUM, C:3375) are spread over the devices
    struct gpu_info_struct& H = list[(serial >= 0 ? serial : 0) % common->numGPU];
    if (!H.st) return SF_ERR_NO_DEVICE;
    const bool trace = getenv("SF_TRACE") != nullptr;      // stderr: where the time of this call goes
    const auto tk0 = std::chrono::steady_clock::now();
    const PlanKey key = make_key(lu, n, nsuper, Super, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui);
    const auto tk1 = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> guard(H.st->mu);
    HandlerState& S = *H.st;
    sf_chol_plan* plan = nullptr;
    for (HandlerState::Entry& e : S.cache)
        if (e.key == key) { plan = e.plan; e.stamp = ++S.clock; break; }
    if (!plan) {
        // A factor that does not fit the device (or SF_DEVICE_BUDGET_MB, for tests and shared devices) is factorized OUT OF CORE:
        // top panels resident, subtree groups streamed through two buffers while the finished ones travel to Lsx_out -- the role the
        // reference gives its slot-sized stages (C:1721-1846, C:2421-2467).  Decided here, per pattern, before the plan is built.
        std::vector<int32_t> ooc_group;
        int ooc_ngroups = 1, ooc_mode = 0;
        auto plan_ooc = [&](double shrink) -> int {
            int64_t entries = 0;
            for (sf_long s = 0; s < nsuper; ++s) entries += (Super[s + 1] - Super[s]) * (Lsip[s + 1] - Lsip[s]);
            const int64_t per_entry = lu ? 16 : 8;
            // everything a plan keeps on the device besides the panels, generously: values, structure, relative maps, task tables, rings
            const int64_t overhead = 12 * (int64_t)Lp[n] + (lu && Up ? 12 * (int64_t)Up[n] : 0) + 24 * (int64_t)Lsip[nsuper] + ((int64_t)384 << 20);
            int64_t budget = 0;
            if (const char* env = getenv("SF_DEVICE_BUDGET_MB")) budget = (int64_t)strtoll(env, nullptr, 10) << 20;
            if (budget <= 0) {
                // (a small factor is not worth the question: the estimate of everything else a plan needs is generous, and a device that
                //  is nearly full should still take a matrix of a few MB -- the allocation itself will tell)
                if (entries * per_entry < ((int64_t)64 << 20)) { ooc_ngroups = 1; return SF_OK; }
                size_t fr = 0, tot = 0;
                if (hipSetDevice(H.gpuIndex_physical) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return SF_ERR_HIP;
                budget = (int64_t)fr - ((int64_t)1 << 30);
                // (the factor of an in-core plan goes into the handler's idle pool if it fits there: counted as free for exactly that)
                if (S.pool && !S.pool_user && (size_t)(entries * per_entry + 16) <= S.pool_bytes) budget += entries * per_entry + 16;
            }
            budget = (int64_t)((double)budget * shrink);
            ooc_ngroups = 1;
            if (entries * per_entry + overhead <= budget) return SF_OK;           // in core
            const int64_t budget_entries = (budget - overhead) / per_entry;
            if (budget_entries <= 0) return SF_ERR_ALLOC;
            ooc_group.assign((size_t)std::max<sf_long>(nsuper, 1), 0);
            int64_t ge = 0, te = 0, nd = 0;
            const int rc = sf::ooc_partition(nsuper, Super, SuperMap, Lsip, Lsi, budget_entries, ooc_group.data(), &ooc_ngroups, &ge, &te, &nd, &ooc_mode);
            if (trace || rc)
                fprintf(stderr, "[sparseframe-hip] factorize: %.2f GB of panels against a device budget of %.2f GB -> out of core: %d groups, "
                                "top %.2f GB %s + 2 buffers of %.2f GB%s\n", entries * per_entry / 1e9, budget / 1e9, ooc_ngroups,
                        te * per_entry / 1e9, ooc_mode == 0 ? "resident (top mode 0)" : (ooc_mode == 1 ? "for the active top panels (top mode 1)" : "for the active top panels (top mode 2)"), ge * per_entry / 1e9, rc == 2 ? " -- DOES NOT FIT" : "");
            return rc == 0 ? SF_OK : (rc == 2 ? SF_ERR_ALLOC : SF_ERR_ARG);
        };
        auto create = [&]() {
            const bool offered = S.pool && !S.pool_user;
            if (offered) sf_plan_offer_factor_buffer(S.pool, S.pool_bytes);
            struct Claim {      // whoever got the buffer is its user until it is dropped
                HandlerState& S; sf_chol_plan*& plan; bool offered;
                ~Claim() { sf_plan_offer_factor_buffer(nullptr, 0); if (offered && plan && sf_plan_factor_borrowed(plan)) S.pool_user = plan; }
            } claim{S, plan, offered};
            if (ooc_ngroups > 1)
                return lu ? sf_lu_plan_create_ooc(&plan, H.gpuIndex_physical, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui, ooc_group.data(), ooc_ngroups, ooc_mode)
                          : sf_chol_plan_create_ooc(&plan, H.gpuIndex_physical, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, ooc_group.data(), ooc_ngroups, ooc_mode);
            return lu ? sf_lu_plan_create(&plan, H.gpuIndex_physical, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui)
                      : sf_chol_plan_create(&plan, H.gpuIndex_physical, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li);
        };
        // make room first when the cache is full; when the device is out of memory, drop everything cached and retry once
        auto drop = [&](sf_chol_plan* q) { if (S.pool_user == q) S.pool_user = nullptr; destroy_plan(q); };
        while (S.cache.size() >= MAX_CACHED_PLANS) {
            size_t lru = 0;
            for (size_t i = 1; i < S.cache.size(); ++i)
                if (S.cache[i].stamp < S.cache[lru].stamp) lru = i;
            drop(S.cache[lru].plan);
            S.cache.erase(S.cache.begin() + lru);
        }
        int rc = plan_ooc(1.0);
        if ((rc != SF_OK || ooc_ngroups > 1) && !S.cache.empty() && !getenv("SF_DEVICE_BUDGET_MB")) {
            // the free memory the decision saw did not count what the cached plans hold: they go first, then the decision is taken again
            for (HandlerState::Entry& e : S.cache) drop(e.plan);
            S.cache.clear();
            (void)hipDeviceSynchronize();
            rc = plan_ooc(1.0);
        }
        if ((rc != SF_OK || ooc_ngroups > 1) && S.pool && !S.pool_user && !getenv("SF_DEVICE_BUDGET_MB")) {
            // ... and so does the idle pool when the factor is too large for it: the memory serves the plan better directly
            S.release_pool();
            rc = plan_ooc(1.0);
        }
        if (rc == SF_OK) rc = create();
        if ((rc == SF_ERR_ALLOC || rc == SF_ERR_HIP) && !S.cache.empty()) {
            for (HandlerState::Entry& e : S.cache) drop(e.plan);
            S.cache.clear();
            (void)hipGetLastError();
            rc = plan_ooc(1.0);
            if (rc == SF_OK) rc = create();
        }
        if ((rc == SF_ERR_ALLOC || rc == SF_ERR_HIP) && S.pool && !S.pool_user) {
            // the pool itself is in the way (a factor larger than the pool, on a device that holds little else): give it back for good
            S.release_pool();
            (void)hipGetLastError();
            rc = plan_ooc(1.0);
            if (rc == SF_OK) rc = create();
        }
        // the estimate of what else lives on the device was too low (or somebody else took memory meanwhile): cut deeper, twice
        for (double shrink = 0.8; rc == SF_ERR_ALLOC && shrink > 0.6; shrink -= 0.15) {
            (void)hipGetLastError();
            if (plan_ooc(shrink) != SF_OK) break;
            rc = create();
        }
        if (rc) return rc;
        S.cache.push_back(HandlerState::Entry{key, plan, ++S.clock});
        ++S.builds;
    }
    const auto tk2 = std::chrono::steady_clock::now();
    forget_plan(plan);              // whatever host copy this plan's factor stood for is about to be overwritten on the device
    int rc = apply_pivot_policy(plan);
    if (!rc) rc = sf_chol_plan_factorize_to_host(plan, Lx, Ux, Lsx_out);
    if (!rc && lu && PivOut) rc = sf_lu_plan_get_pivots(plan, PivOut);
    if (!rc && lu) note_perturbed(Lsx_out, plan->last_perturbed);
    if (!rc && plan->ooc_groups <= 1) {        // (an out-of-core factor exists on the host only: its solves take the host sweep)
        Resident R;
        R.plan = plan;
        R.epoch = plan->epoch;
        std::lock_guard<std::mutex> g(g_res_mu);
        g_resident[(const void*)Lsx_out] = std::move(R);
    } else if (!rc) {
        std::lock_guard<std::mutex> g(g_res_mu);
        g_resident.erase((const void*)Lsx_out);
    }
    if (trace) {
        const auto tk3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[sparseframe-hip] factorize: pattern hash %.1f ms, plan lookup/build %.1f ms, H2D + numeric + overlapped D2H %.1f ms\n",
                ms(tk0, tk1), ms(tk1, tk2), ms(tk2, tk3));
        fprintf(stderr, "[sparseframe-hip]   copy workers %s, last on CPUs", plan->dl_cpus_known == 1 ? "confined to the device's NUMA node" : "not confined");
        for (int w = 0; w < plan->dl_workers_last; ++w) fprintf(stderr, " %d", plan->dl_last_cpu[w]);
        fprintf(stderr, "\n");
    }
    return rc;
}

// SparseFrame_solve_supernodal's fast path: see sparseframe_flat.h.  g_res_mu is held for the whole solve: a factorization or a
// destruction of the same plan waits in forget_plan until it is over (lock order there: handler lock, then g_res_mu; here only
// g_res_mu), so nobody else touches the plan meanwhile.
int sf_handlers_solve_resident_sym(const sf_float* Lsx_host, const sf_float* b, sf_float* x, int lu, sf_long n, sf_long nsuper,
                                   const sf_long* Super, const sf_long* SuperMap, const sf_long* Lsip, const sf_long* Lsi,
                                   const sf_long* Lsxp, const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui) {
    if (!Lsx_host || !b || !x) return SF_ERR_ARG;
    const bool trace = getenv("SF_TRACE") != nullptr;
    auto why = [&](const char* m) { if (trace) fprintf(stderr, "[sparseframe-hip] resident solve not used: %s\n", m); return SF_ERR_ARG; };
    if (const char* e = getenv("SF_SOLVE"))
        if (!strcmp(e, "host")) return SF_ERR_ARG;
    const int mode = g_resident_mode.load();
    if (mode == 0) return SF_ERR_ARG;
    std::lock_guard<std::mutex> g(g_res_mu);
    auto it = g_resident.find((const void*)Lsx_host);
    if (it == g_resident.end()) return why("no resident factor is registered for this Lsx");
    Resident& R = it->second;
    sf_chol_plan* plan = R.plan;
    if (!R.parts.empty()) {
        // distributed factor: gather it (once per factorization) into a whole plan on the first handler's device -- panels travel
        // device to device, the solve then runs there.  Needs the symbolic arrays (to build that plan) and room for the whole
        // factor on one device; otherwise the caller solves on the host.
        for (size_t r = 0; r < R.parts.size(); ++r)
            if (R.parts[r]->epoch != R.part_epochs[r]) { g_resident.erase(it); return why("a rank's plan holds a later factorization"); }
        SolverSlot& S = *R.slot;
        // The factor stays where it is and the ranks solve together (sf_chol_plan_solve_distributed), one host thread per rank as in
        // the factorization.  SF_SOLVE=gather: gather the panels into a whole plan on the first handler's device instead (below).
        bool distributed = true;
        if (const char* e = getenv("SF_SOLVE")) distributed = strcmp(e, "gather") != 0;
        if (distributed && R.comms.size() == R.parts.size()) {
            // the host copy must still be what the devices hold: EVERY panel's fingerprint over the caller's array against the one
            // every rank that stores the panel computes over its own copy (k_factor_hash; Cholesky and LU)
            const size_t N = R.parts.size();
            const auto tv0 = std::chrono::steady_clock::now();
            if (mode == 1) {
                sf_chol_plan* P0 = R.parts[0];
                std::vector<uint64_t> hh;
                host_panel_hashes(Lsx_host, P0->h_Lsxp.data(), P0->nsuper, hh);
                if (trace) fprintf(stderr, "[sparseframe-hip] resident solve: host fingerprints %.1f ms\n",
                                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count());
                for (size_t r = 0; r < N; ++r) {
                    const uint64_t* dh = nullptr;
                    if (sf_plan_panel_hashes(R.parts[r], &dh) != SF_OK) return why("a rank's fingerprints could not be computed");
                    for (int64_t sn = 0; sn < P0->nsuper; ++sn)
                        if (R.parts[r]->h_XP[sn] >= 0 && dh[sn] != hh[(size_t)sn]) {
                            g_resident.erase(it);
                            ++g_fingerprint_fallbacks;
                            return why("the host copy differs from a rank's panel (fingerprint)");
                        }
                }
            }
            const auto tv1 = std::chrono::steady_clock::now();
            std::vector<int> rcs(N, SF_OK);
            std::vector<std::thread> th;
            for (size_t r = 0; r < N; ++r)
                th.emplace_back([&, r] { rcs[r] = sf_chol_plan_solve_distributed(R.parts[r], R.comms[r], b, x); });
            for (std::thread& t : th) t.join();
            if (trace) fprintf(stderr, "[sparseframe-hip] resident solve: verification %.1f ms, distributed solve %.1f ms\n",
                               std::chrono::duration<double, std::milli>(tv1 - tv0).count(),
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv1).count());
            for (size_t r = 0; r < N; ++r)
                if (rcs[r]) return why("the distributed solve failed");
            ++g_resident_solves;
            return SF_OK;
        }
        if (!S.plan) {
            if (!Super || !SuperMap || !Lsip || !Lsi || !Lsxp || !Lp || !Li || nsuper <= 0) return why("distributed factor and no symbolic arrays");
            size_t free_b = 0, total_b = 0;
            if (hipSetDevice(R.device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return why("hipMemGetInfo failed");
            const double need = (double)Lsxp[nsuper] * 8.0 * (lu ? 2.0 : 1.0) * 1.15 + 2e9;
            if ((double)free_b < need) return why("the first handler's device has no room for the whole factor");
            const int rc = lu ? sf_lu_plan_create(&S.plan, R.device, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui)
                              : sf_chol_plan_create(&S.plan, R.device, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li);
            if (rc) { S.plan = nullptr; (void)hipGetLastError(); return why("the whole plan could not be built"); }
            S.imported.clear();
        }
        if (S.imported != R.part_epochs) {
            if (sf_plan_import_from(S.plan, R.parts.data(), (int)R.parts.size()) != SF_OK) { S.imported.clear(); return why("gathering the panels failed"); }
            S.imported = R.part_epochs;
        }
        plan = S.plan;
    } else if (plan->epoch != R.epoch || plan->partial) {
        g_resident.erase(it);
        return why("the plan holds a later factorization");
    }
    // the host copy must still be what the device holds (it came from there bit by bit): the fingerprint of every panel of the
    // caller's array against the device's (one pass over the host array, threaded; the device side is cached per factorization)
    if (mode == 1) {
        const uint64_t* dh = nullptr;
        if (sf_plan_panel_hashes(plan, &dh) != SF_OK) return why("the device fingerprints could not be computed");
        std::vector<uint64_t> hh;
        host_panel_hashes(Lsx_host, plan->h_Lsxp.data(), plan->nsuper, hh);
        for (int64_t sn = 0; sn < plan->nsuper; ++sn)
            if (dh[sn] != hh[(size_t)sn]) {
                g_resident.erase(it);
                ++g_fingerprint_fallbacks;
                return why("the host copy differs from the device's (fingerprint)");
            }
    }
    const int rc = plan->lu ? sf_lu_plan_solve(plan, b, x) : sf_chol_plan_solve(plan, b, x);
    if (!rc) ++g_resident_solves;
    return rc;
}

// how often a verified resident solve found the caller's array different from the device's factor and left the solve to the host sweep:
// expected only after the caller changed Lsx; anything else (e.g. replicas of a shared panel that are no longer bit-identical) shows
// up here instead of as a silently slower solve (bench.py reports it, the tests assert on it)
int64_t sf_handlers_fingerprint_fallbacks(void) { return g_fingerprint_fallbacks.load(); }

int sf_handlers_solve_resident(const sf_float* Lsx_host, const sf_float* b, sf_float* x) {
    return sf_handlers_solve_resident_sym(Lsx_host, b, x, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// Test hook: how many values of the panels that several ranks store (the shared top supernodes of a multi-handler factorization)
// differ BITWISE between a rank's copy and the first holder's.  0 is the invariant the LU pivot decisions rest on (plan_create);
// -1: no multi-handler factor is registered for this host array.
int64_t sf_handlers_replica_mismatches(const sf_float* Lsx_host) {
    std::lock_guard<std::mutex> g(g_res_mu);
    auto it = g_resident.find((const void*)Lsx_host);
    if (it == g_resident.end() || it->second.parts.size() < 2) return -1;
    const std::vector<sf_chol_plan*>& parts = it->second.parts;
    const sf_chol_plan* P0 = parts[0];
    int64_t bad = 0;
    std::vector<double> a, b;
    for (int64_t s = 0; s < P0->nsuper; ++s) {
        const size_t len = (size_t)(P0->h_Lsip[s + 1] - P0->h_Lsip[s]) * (size_t)(P0->h_Super[s + 1] - P0->h_Super[s]);
        const sf_chol_plan* first = nullptr;
        for (sf_chol_plan* P : parts) {
            if (P->h_XP[s] < 0) continue;
            std::vector<double>& dst = first ? b : a;
            dst.resize(len * (P->lu ? 2 : 1));
            if (hipSetDevice(P->device) != hipSuccess ||
                hipMemcpy(dst.data(), P->d_Lsx + P->h_XP[s], len * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
            if (P->lu && !P->u_alias &&
                hipMemcpy(dst.data() + len, P->d_Lsx + P->xC + P->h_XP[s], len * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
            if (P->lu && P->u_alias) dst.resize(len);
            if (!first) { first = P; continue; }
            // (LU: the upper triangle of an L panel's diagonal block is not factor data -- it is scratch for the download, filled
            // with U11 on the ranks that copy that block column back, see k_lu_fill_u11)
            const size_t nsr = (size_t)(P0->h_Lsip[s + 1] - P0->h_Lsip[s]), nsc = (size_t)(P0->h_Super[s + 1] - P0->h_Super[s]);
            for (size_t i = 0; i < a.size(); ++i) {
                if (P->lu && i < len && (i % nsr) <= (i / nsr) && (i % nsr) < nsc) continue;
                bad += memcmp(&a[i], &b[i], sizeof(double)) != 0;
            }
        }
    }
    return bad;
}

int64_t sf_handlers_resident_solves(void) {
    std::lock_guard<std::mutex> g(g_res_mu);
    return g_resident_solves;
}

void sf_handlers_forget(const sf_float* Lsx_host) {
    if (!Lsx_host) return;
    {
        std::lock_guard<std::mutex> g(g_res_mu);
        g_resident.erase((const void*)Lsx_host);
    }
    std::lock_guard<std::mutex> g(g_piv_mu);
    g_perturbed.erase((const void*)Lsx_host);
}

// 1 when the library was built with the A/B switches of finished experiments (make EXP=1), 0 for a release build
int sf_build_experiments(void) {
#ifdef SF_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int sf_handlers_set_resident_solve(int mode) {
    if (mode < 0 || mode > 2) return SF_ERR_ARG;
    g_resident_mode.store(mode);
    return SF_OK;
}

int sf_handlers_set_lu_pivoting(double tol, double perturb) {
    if (!(tol >= 0.0) || tol > 1.0 || !(perturb >= 0.0)) return SF_ERR_ARG;
    std::lock_guard<std::mutex> g(g_piv_mu);
    g_piv_set = true;
    g_piv_tol = tol;
    g_piv_perturb = perturb;
    return SF_OK;
}

int sf_handlers_set_lu_pivoting_next_call(double tol, double perturb) {
    if (!(tol >= 0.0) || tol > 1.0 || !(perturb >= 0.0)) return SF_ERR_ARG;
    t_piv_set = true;
    t_piv_tol = tol;
    t_piv_perturb = perturb;
    return SF_OK;
}

int64_t sf_handlers_perturbed_pivots(const sf_float* Lsx_host) {
    std::lock_guard<std::mutex> g(g_piv_mu);
    auto it = g_perturbed.find((const void*)Lsx_host);
    return it == g_perturbed.end() ? -1 : it->second;
}

int SparseFrame_allocate_gpu(struct common_info_struct* common, struct gpu_info_struct** list) { return sf_handlers_allocate(common, list); }
int SparseFrame_free_gpu(struct common_info_struct* common, struct gpu_info_struct** list) { return sf_handlers_free(common, list); }

int SparseFrame_factorize_supernodal(struct common_info_struct* common, struct gpu_info_struct* list,
                                     struct matrix_info_struct* mi) {
    if (!common || !mi || !mi->Lsx) return SF_ERR_ARG;
    return sf_handlers_factorize(common, list, 0, mi->serial, mi->nrow, mi->nsuper, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi,
                                 mi->Lsxp, mi->Lp, mi->Li, nullptr, nullptr, mi->Lx, nullptr, mi->Lsx, nullptr);
}

int SparseFrame_factorize(struct common_info_struct* common, struct gpu_info_struct* list, struct matrix_info_struct* mi) {
    struct timespec a, b;
    clock_gettime(CLOCK_REALTIME, &a);
    const int rc = SparseFrame_factorize_supernodal(common, list, mi);
    clock_gettime(CLOCK_REALTIME, &b);
    if (mi) mi->factorizeTime = (b.tv_sec - a.tv_sec) + (b.tv_nsec - a.tv_nsec) / 1.0e9;
    return rc;
}

}  // extern "C"
