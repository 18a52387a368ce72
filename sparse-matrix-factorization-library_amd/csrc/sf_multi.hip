// Multi-GPU orchestration inside the library: the parent-front merge of a sharded factorization as RCCL
// collectives on the plan's own stream, driven from C.
//
//   sf_comm                 one rank's end of a group of `nranks` plans that factorize ONE matrix together
//     kind RCCL             ncclComm_t (RCCL = the ROCm build of NCCL, over xGMI inside a node).  librccl.so.1 is loaded at run
//                           time with dlopen, so a single-GPU user never needs it and a process that already loaded a copy
//                           (PyTorch bundles one) shares that copy.
//     kind LOCAL            ranks are threads of this process whose plans live on ONE device (emulated ranks: tests and
//                           rehearsals on a one-GPU box): event hand-shakes + a sum kernel on that device.  Same call sites
//                           and ordering as the RCCL kind; never used when the ranks have devices of their own.
//   sf_chol_plan_factorize_distributed   phase 0 (own subtrees), then per segment: pack -> all-reduce(sum) -> chain +
//                           this rank's share of the split GEMMs (sf_chol_plan.hip), all enqueued on the plan's stream with
//                           no host synchronisation; optionally with the overlapped copy-back of this rank's pieces.
//
// The reference has no inter-GPU path: its device handlers exchange panels through pinned host memory under one task
// queue (Cholesky/Source/SparseFrame.c:2267, 2421-2467).
#include <sparseframe_hip.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#include "sf_kernels.h"
#include "sf_plan_internal.h"

namespace {

// ---- RCCL entry points, resolved once ----
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            fprintf(stderr, "[sparseframe-hip] cannot load librccl.so.1 (%s): multi-GPU factorization is unavailable\n", dlerror());
            return;
        }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.handle, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))dlsym(api.handle, "ncclCommAbort");
        api.CommSplit = (decltype(api.CommSplit))dlsym(api.handle, "ncclCommSplit");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.CommSplit && api.AllReduce && api.GetErrorString;
        if (!api.ok) fprintf(stderr, "[sparseframe-hip] librccl.so.1 lacks an expected entry point\n");
    });
    return api;
}

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) {                                                                                \
            fprintf(stderr, "[sparseframe-hip] %s failed: %s (%s:%d)\n", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); \
            return SF_ERR_HIP;                                                                                  \
        }                                                                                                       \
    } while (0)

// ---- LOCAL kind: threads of one process, one device ----
constexpr int LOCAL_MAX = 16;
struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    double* bufs[LOCAL_MAX] = {};
    int64_t counts[LOCAL_MAX] = {};
    hipEvent_t ev_ready[LOCAL_MAX] = {}, ev_done[LOCAL_MAX] = {};
    hipEvent_t ev_sum = nullptr;
    int refs = 0;
    uint64_t failed_op = 0;         // number of the last collective a member failed in (0 = none): a failure does not outlive its collective
    std::vector<std::pair<uint32_t, LocalGroup*>> children;      // sub-groups by rank mask (guarded by mu)
    void barrier() {
        std::unique_lock<std::mutex> g(mu);
        const uint64_t gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(g, [&] { return generation != gen; });
    }
};

struct PtrTable { const double* p[LOCAL_MAX]; };

__global__ void __launch_bounds__(256)
k_sum_ranks(double* __restrict__ dst, PtrTable src, int nsrc, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        double v = dst[i];
        for (int q = 0; q < nsrc; ++q) v += src.p[q][i];
        dst[i] = v;
    }
}

}  // namespace

struct sf_comm {
    int kind = 0;       // 0 RCCL, 1 LOCAL
    int rank = 0, nranks = 1;
    int device = 0;
    ncclComm_t nccl = nullptr;
    LocalGroup* local = nullptr;
    uint64_t ops = 0;                   // LOCAL: collectives this member has taken part in (the members advance in lockstep)
    bool dead = false;                  // RCCL: aborted after a failure in the middle of a run; every later call fails at once
    // sub-communicators of the groups of a proportionally mapped factorization, by the mask of WORLD ranks they hold
    std::vector<std::pair<uint32_t, sf_comm*>> subs;
    std::vector<uint32_t> prepared;     // every mask a collective split has been made for (member or not)
};

namespace {

int local_allreduce(sf_comm* c, double* buf, int64_t count, hipStream_t st) {
    LocalGroup& G = *c->local;
    const int r = c->rank, n = G.n;
    const uint64_t op = ++c->ops;
    // (max / >=: a failure in the NEXT collective, set by a member that is already past this one's last barrier, may be seen here too;
    // that errs on the side of reporting)
    auto fail = [&] { std::lock_guard<std::mutex> g(G.mu); G.failed_op = std::max(G.failed_op, op); };
    auto failed = [&] { std::lock_guard<std::mutex> g(G.mu); return G.failed_op >= op; };
    G.bufs[r] = buf;
    G.counts[r] = count;
    // count < 0: this rank has failed and only keeps the hand-shake going so that the others do not wait for ever
    bool ok = count >= 0 && hipEventRecord(G.ev_ready[r], st) == hipSuccess;
    if (!ok) fail();
    G.barrier();
    ok = ok && !failed();
    if (r == 0) {
        for (int q = 1; q < n && ok; ++q) {
            ok = ok && G.counts[q] == count;        // every rank reduces the same segment
            ok = ok && hipStreamWaitEvent(st, G.ev_ready[q], 0) == hipSuccess;
        }
        if (ok && count > 0) {
            PtrTable t{};
            for (int q = 1; q < n; ++q) t.p[q - 1] = G.bufs[q];
            const int64_t blocks = std::min<int64_t>((count + 255) / 256, 4096);
            hipLaunchKernelGGL(k_sum_ranks, dim3((unsigned)blocks), dim3(256), 0, st, buf, t, n - 1, count);
            ok = hipGetLastError() == hipSuccess;
        }
        ok = ok && hipEventRecord(G.ev_sum, st) == hipSuccess;
        if (!ok) fail();
    }
    G.barrier();
    ok = ok && !failed();
    if (r > 0) {
        ok = ok && hipStreamWaitEvent(st, G.ev_sum, 0) == hipSuccess;
        if (ok && count > 0) ok = hipMemcpyAsync(buf, G.bufs[0], (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, st) == hipSuccess;
        ok = ok && hipEventRecord(G.ev_done[r], st) == hipSuccess;
    }
    if (!ok) fail();
    G.barrier();
    if (r == 0)
        for (int q = 1; q < n; ++q) ok = ok && hipStreamWaitEvent(st, G.ev_done[q], 0) == hipSuccess;
    return (ok && !failed()) ? SF_OK : SF_ERR_HIP;
}

}  // namespace

static LocalGroup* new_local_group(int n, int device) {
    LocalGroup* G = new (std::nothrow) LocalGroup();
    if (!G) return nullptr;
    G->n = n;
    G->refs = n;
    bool ok = hipSetDevice(device) == hipSuccess && hipEventCreateWithFlags(&G->ev_sum, hipEventDisableTiming) == hipSuccess;
    for (int r = 0; r < n; ++r) {
        ok = ok && hipEventCreateWithFlags(&G->ev_ready[r], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&G->ev_done[r], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { delete G; return nullptr; }
    return G;
}

// used by the handlers (sf_handlers.hip): communicators of all ranks of one process at once
int sf_comm_create_all(sf_comm** comms, int nranks, const int* devices) {
    if (!comms || nranks < 1 || nranks > LOCAL_MAX || !devices) return SF_ERR_ARG;
    bool same_device = nranks > 1;
    bool distinct = true;
    for (int a = 0; a < nranks; ++a)
        for (int b = a + 1; b < nranks; ++b) {
            if (devices[a] != devices[b]) same_device = false; else distinct = false;
        }
    for (int r = 0; r < nranks; ++r) comms[r] = nullptr;
    if (nranks > 1 && !same_device && !distinct) return SF_ERR_ARG;     // a mix of shared and own devices is not supported
    if (same_device) {
        LocalGroup* G = new_local_group(nranks, devices[0]);
        if (!G) return SF_ERR_HIP;
        for (int r = 0; r < nranks; ++r) {
            comms[r] = new sf_comm();
            comms[r]->kind = 1; comms[r]->rank = r; comms[r]->nranks = nranks; comms[r]->device = devices[0]; comms[r]->local = G;
        }
        return SF_OK;
    }
    if (!rccl().ok) return SF_ERR_NO_DEVICE;
    std::vector<ncclComm_t> nc(nranks, nullptr);
    NCCL_TRY(rccl().CommInitAll(nc.data(), nranks, devices));
    for (int r = 0; r < nranks; ++r) {
        comms[r] = new sf_comm();
        comms[r]->kind = 0; comms[r]->rank = r; comms[r]->nranks = nranks; comms[r]->device = devices[r]; comms[r]->nccl = nc[r];
    }
    return SF_OK;
}

// Sub-communicators for the groups `masks` (bit r = world rank r; the same sorted list on every rank).  Collective: every
// rank of `c` calls it with the same list -- ncclCommSplit per group for RCCL (a rank outside a group passes
// NCCL_SPLIT_NOCOLOR), a shared LocalGroup per mask for emulated ranks.  Masks met in an earlier call are skipped.
int sf_comm_prepare_groups(sf_comm* c, const uint32_t* masks, int nmasks) {
    if (!c || (nmasks > 0 && !masks)) return SF_ERR_ARG;
    const uint32_t all = c->nranks >= 32 ? 0xffffffffu : ((1u << c->nranks) - 1u);
    HIP_TRY(hipSetDevice(c->device));       // a handler thread starts with device 0 current: RCCL calls are made with the communicator's device
    for (int k = 0; k < nmasks; ++k) {
        const uint32_t m = masks[k] & all;
        if (m == all || m == 0) continue;                       // the world itself / nobody
        // a mask met before (another plan with the same groups) needs no second split; every rank keeps the same list, so
        // the collective calls still match up
        if (std::find(c->prepared.begin(), c->prepared.end(), m) != c->prepared.end()) continue;
        // (recorded as prepared only once the sub-communicator exists: after a failed split the next call tries again, on every rank)
        const bool mine = ((m >> c->rank) & 1u) != 0;
        const int gsize = __builtin_popcount(m), grank = __builtin_popcount(m & ((1u << c->rank) - 1u));
        if (c->kind == 0) {
            ncclComm_t sub = nullptr;
            NCCL_TRY(rccl().CommSplit(c->nccl, mine ? 0 : NCCL_SPLIT_NOCOLOR, c->rank, &sub, nullptr));
            c->prepared.push_back(m);
            if (!mine) continue;
            sf_comm* sc = new sf_comm();
            sc->kind = 0; sc->rank = grank; sc->nranks = gsize; sc->device = c->device; sc->nccl = sub;
            c->subs.push_back({m, sc});
        } else {
            if (!mine) { c->prepared.push_back(m); continue; }
            LocalGroup* child = nullptr;
            {
                std::lock_guard<std::mutex> g(c->local->mu);
                for (auto& kv : c->local->children)
                    if (kv.first == m) child = kv.second;
                if (!child) {
                    child = new_local_group(gsize, c->device);
                    if (child) c->local->children.push_back({m, child});
                }
            }
            if (!child) return SF_ERR_HIP;
            c->prepared.push_back(m);
            sf_comm* sc = new sf_comm();
            sc->kind = 1; sc->rank = grank; sc->nranks = gsize; sc->device = c->device; sc->local = child;
            c->subs.push_back({m, sc});
        }
    }
    return SF_OK;
}

// test hook (sf_test_inject_failure): rank `g_fail_rank` fails once at point `g_fail_where` (1: before the first collective of a
// distributed factorization, 2: in the middle of its segments, 3: before the first collective of a distributed solve, 4: in the
// middle of its sweeps)
static std::atomic<int> g_fail_rank{-1}, g_fail_where{0};
static bool injected(int rank, int where) {
    if (g_fail_where.load() != where || g_fail_rank.load() != rank) return false;
    g_fail_where.store(0);
    return true;
}

// A communicator that cannot go on (a failure between collectives of a running factorization or solve).  What this does and does
// NOT do: THIS rank's communicator and sub-communicators are aborted (ncclCommAbort, when the library has it) and marked dead, so
// every later call on them fails at once instead of enqueueing work nobody will match.  It does NOT release the peers: a peer that
// already sits in a collective this rank never joins keeps waiting there until ITS OWN side gives up (RCCL's watchdog / the
// launcher's timeout -- bench.py sets 600 s -- or an abort by whoever supervises the job); ncclCommAbort is a local operation.
// Everything that can fail on one rank alone is therefore done BEFORE the first data collective and agreed on (agree_status); a
// failure after that point is a device or link failure, for which "this rank reports, the job is torn down from outside" is the
// contract.  Emulated ranks (LocalGroup) are different: their hand-shakes are host-side and a failed rank keeps them going, so
// the peers return too.  NOTE: the RCCL paths below have run on ONE rank only (no multi-GPU box in four rounds); the emulated
// ranks exercise the same call sequence.
static void abort_comm(sf_comm* c) {
    if (!c || c->kind != 0 || c->dead) return;
    c->dead = true;
    for (auto& kv : c->subs) abort_comm(kv.second);
    if (c->nccl && rccl().CommAbort) { (void)rccl().CommAbort(c->nccl); c->nccl = nullptr; }
}

// Every rank of `comm` learns whether ANY of them has failed so far: one 8-byte sum on `st`, waited for on the host.  Called by all
// ranks whatever their own state, BEFORE the first data collective of a run, so that no rank enqueues collectives a failed peer
// will never match (those would spin on the GPU for ever).  A rank that has failed ALWAYS takes part and contributes 1: the status
// word is allocated with the plan (no allocation here), so the only way not to join is a device that no longer accepts a memcpy --
// then nothing this rank could enqueue would run either, and the peers are left to their timeouts (see abort_comm).
// Returns my_rc if this rank failed, SF_ERR_PEER if only others did.
static int agree_status(sf_chol_plan* p, sf_comm* comm, int my_rc, hipStream_t st) {
    if (comm->nranks == 1) return my_rc;
    bool ok = hipSetDevice(p->device) == hipSuccess;
    if (ok && !p->d_status) ok = hipMalloc((void**)&p->d_status, sizeof(double)) == hipSuccess;      // (plans made before round 4 only)
    double flag = my_rc ? 1.0 : 0.0, got = 1.0;
    ok = ok && hipMemcpyAsync(p->d_status, &flag, sizeof flag, hipMemcpyHostToDevice, st) == hipSuccess;
    int rc;
    if (ok) rc = sf_comm_allreduce_sum(comm, p->d_status, 1, st);
    else rc = comm->kind == 1 ? (sf_comm_allreduce_sum(comm, nullptr, -1, st), SF_ERR_HIP) : SF_ERR_HIP;
    if (!rc && (hipMemcpyAsync(&got, p->d_status, sizeof got, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)) rc = SF_ERR_HIP;
    if (my_rc) return my_rc;
    if (rc) return rc;
    return got != 0.0 ? SF_ERR_PEER : SF_OK;
}

static sf_comm* group_comm(sf_comm* c, uint32_t mask) {
    const uint32_t all = c->nranks >= 32 ? 0xffffffffu : ((1u << c->nranks) - 1u);
    if ((mask & all) == all) return c;
    for (auto& kv : c->subs)
        if (kv.first == (mask & all)) return kv.second;
    return nullptr;
}

extern "C" {

int sf_comm_unique_id(char* id128) {
    if (!id128) return SF_ERR_ARG;
    if (!rccl().ok) return SF_ERR_NO_DEVICE;
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    static_assert(sizeof(id.internal) == 128, "unique id is 128 bytes");
    memcpy(id128, id.internal, 128);
    return SF_OK;
}

int sf_comm_create_rccl(sf_comm** out, int device, int rank, int nranks, const char* id128) {
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return SF_ERR_ARG;
    *out = nullptr;
    if (!rccl().ok) return SF_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclComm_t nc = nullptr;
    NCCL_TRY(rccl().CommInitRank(&nc, nranks, id, rank));
    sf_comm* c = new (std::nothrow) sf_comm();
    if (!c) return SF_ERR_ALLOC;
    c->kind = 0; c->rank = rank; c->nranks = nranks; c->device = device; c->nccl = nc;
    *out = c;
    return SF_OK;
}

// test hook: ncclCommSplit of the whole communicator (every rank passes color 0), one all-reduce on the child, destroy --
// the sub-communicator calls of the proportional mapping on the real library even where only one rank exists
int sf_comm_selftest_split(sf_comm* c, void* device_buf, sf_long count, void* stream) {
    if (!c || c->kind != 0 || !rccl().ok) return SF_ERR_ARG;
    ncclComm_t sub = nullptr;
    NCCL_TRY(rccl().CommSplit(c->nccl, 0, c->rank, &sub, nullptr));
    if (!sub) return SF_ERR_HIP;
    const ncclResult_t r = rccl().AllReduce(device_buf, device_buf, (size_t)count, ncclDouble, ncclSum, sub, (hipStream_t)stream);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)rccl().CommDestroy(sub);
    return (r == ncclSuccess && e == hipSuccess) ? SF_OK : SF_ERR_HIP;
}

// Everything a distributed run of `p` needs from `comm`, done up front and checked: the sub-communicators of the plan's groups
// (ncclCommSplit) and ONE 8-byte sum on the world and on every group this rank belongs to, whose result must be the group's
// size.  Collective (every rank of `comm`, with its own plan of the same factorization).  A launcher calls it right after
// creating plan and communicator, so that a broken RCCL setup shows up as an error code there -- where falling back to another
// path is still possible -- and not in the middle of the first factorization.
int sf_chol_plan_prepare_comm(sf_chol_plan* p, sf_comm* comm) {
    if (!p || !comm || comm->nranks != p->nranks || comm->rank != p->rank) return SF_ERR_ARG;
    int rc = sf_comm_prepare_groups(comm, p->all_masks.data(), (int)p->all_masks.size());
    if (rc) return rc;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, sizeof(double)));
    std::vector<sf_comm*> cs{comm};
    for (uint32_t m : p->all_masks)                  // ascending masks: every rank meets the groups it shares with another in one order
        if (sf_comm* g = group_comm(comm, m))
            if (g != comm && ((m >> comm->rank) & 1u) && std::find(cs.begin(), cs.end(), g) == cs.end()) cs.push_back(g);
    for (sf_comm* c : cs) {
        const double one = 1.0;
        double got = 0.0;
        if (hipMemcpyAsync(d, &one, sizeof one, hipMemcpyHostToDevice, p->stream) != hipSuccess) { rc = SF_ERR_HIP; break; }
        if ((rc = sf_comm_allreduce_sum(c, d, 1, p->stream))) break;
        if (hipMemcpyAsync(&got, d, sizeof got, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
            hipStreamSynchronize(p->stream) != hipSuccess) { rc = SF_ERR_HIP; break; }
        if (got != (double)c->nranks) { rc = SF_ERR_HIP; break; }
    }
    (void)hipFree(d);
    return rc;
}

int sf_comm_rank(const sf_comm* c) { return c ? c->rank : -1; }
int sf_comm_size(const sf_comm* c) { return c ? c->nranks : 0; }

int sf_comm_destroy(sf_comm* c) {
    if (!c) return SF_OK;
    for (auto& kv : c->subs) sf_comm_destroy(kv.second);
    c->subs.clear();
    if (c->kind == 0 && c->nccl) (void)rccl().CommDestroy(c->nccl);
    if (c->kind == 1 && c->local) {
        LocalGroup* G = c->local;
        bool last;
        { std::lock_guard<std::mutex> g(G->mu); last = --G->refs == 0; }
        if (last) {
            (void)hipSetDevice(c->device);
            if (G->ev_sum) (void)hipEventDestroy(G->ev_sum);
            for (int r = 0; r < G->n; ++r) {
                if (G->ev_ready[r]) (void)hipEventDestroy(G->ev_ready[r]);
                if (G->ev_done[r]) (void)hipEventDestroy(G->ev_done[r]);
            }
            G->children.clear();        // the sub-groups are owned (reference-counted) by their members' communicators
            delete G;
        }
    }
    delete c;
    return SF_OK;
}

// sum `count` doubles at `device_buf` over the ranks, in place, ordered on `stream` (no host synchronisation)
int sf_comm_allreduce_sum(sf_comm* c, void* device_buf, sf_long count, void* stream) {
    if (!c || (count > 0 && !device_buf)) return SF_ERR_ARG;
    if (count < 0 && c->kind != 1) return SF_ERR_ARG;
    if (c->dead) return SF_ERR_PEER;
    if (c->kind == 1) return c->nranks == 1 ? SF_OK : local_allreduce(c, (double*)device_buf, count, (hipStream_t)stream);
    if (count == 0) return SF_OK;
    HIP_TRY(hipSetDevice(c->device));
    NCCL_TRY(rccl().AllReduce(device_buf, device_buf, (size_t)count, ncclDouble, ncclSum, c->nccl, (hipStream_t)stream));
    return SF_OK;
}

// One sharded factorization from this rank's point of view (plan from sf_chol_plan_create_distributed /
// sf_lu_plan_create_distributed with the same rank / nranks as `comm`): own subtrees, then for every segment the
// parent-front merge (one all-reduce of the packed block columns) and the segment's launches.  host_out != NULL: the
// pieces of the factor this rank is responsible for are copied into it while the factorization runs.
// The solve with a factor that stays distributed over the ranks' mapped plans (create_mapped), after
// sf_chol_plan_factorize_distributed: see the schedule in sf_chol_plan.hip ("distributed solve").  b_host: the whole right-hand
// side (permuted numbering) on every rank; x_host: every rank writes the entries it is responsible for (its subtrees' columns, and
// the columns of the shared supernodes whose group it leads) and leaves the others alone -- ranks that are threads of one process
// may share one x_host, separate processes merge theirs (the entries of a column range come from exactly one rank).
// The only communication is one small sum per shared supernode in the forward sweep (its columns of x, inside its group).
int sf_chol_plan_solve_distributed(sf_chol_plan* p, sf_comm* comm, const sf_float* b_host, sf_float* x_host) {
    if (!p || !comm || !b_host || !x_host || comm->nranks != p->nranks || comm->rank != p->rank) return SF_ERR_ARG;
    if (p->nranks == 1 && !p->partial) return sf_chol_plan_solve(p, b_host, x_host);
    int rc = sf_comm_prepare_groups(comm, p->all_masks.data(), (int)p->all_masks.size());     // collective: before any early return
    if (rc) return rc;
    // Everything that can fail on this rank alone happens BEFORE the first reduce, and the ranks agree on the outcome (a rank that
    // stores nothing still takes part in the agreement): nobody enqueues a sum a failed peer will never join.
    const int64_t n = p->n;
    const bool idle = p->solve_steps.empty() || n <= 0;
    hipStream_t st = p->stream;
    std::vector<double> xb((size_t)std::max<int64_t>(n, 1), 0.0);
    if (!idle && (!p->d_solve || !p->d_x)) rc = SF_ERR_ARG;
    if (!rc && hipSetDevice(p->device) != hipSuccess) rc = SF_ERR_HIP;
    const size_t nst = p->solve_steps.size();
    if (!rc && !idle) {
        // right-hand side: the columns this rank loads (the others start from zero: they only collect this rank's updates)
        for (const auto& r : p->solve_load)
            memcpy(xb.data() + r.first, b_host + r.first, (size_t)(r.second - r.first) * sizeof(double));
        if (hipMemcpyAsync(p->d_x, xb.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) rc = SF_ERR_HIP;
        if (!rc && p->d_solve_sync &&
            hipMemsetAsync(p->d_solve_sync, 0, (size_t)(1 + p->n_solve_sync + sf_chol_plan::SOLVE_TICKETS * nst) * sizeof(int), st) != hipSuccess)
            rc = SF_ERR_HIP;
    }
    if (!rc && injected(p->rank, 3)) rc = SF_ERR_HIP;
    if ((rc = agree_status(p, comm, rc, st))) return rc;
    if (idle) return SF_OK;               // a rank that stores nothing (more ranks than subtrees) reports nothing
    const double* fwd_base = p->d_Lsx;
    const double* bwd_base = p->lu ? p->d_Lsx + p->xC : p->d_Lsx;
    int* sync = p->d_solve_sync + 1;
    int* tickets = sync + p->n_solve_sync;
    for (size_t k = 0; k < nst; ++k) {
        const auto& s = p->solve_steps[k];
        if (!rc && k > 0 && injected(p->rank, 4)) rc = SF_ERR_HIP;
        for (int q = 0; q < s.red_count; ++q) {
            const auto& R = p->solve_reduces[(size_t)s.red_first + q];
            sf_comm* gc = group_comm(comm, R.mask);
            if (!gc) { if (!rc) rc = SF_ERR_ARG; continue; }
            // after a failure: emulated ranks keep the hand-shake of every remaining sum going (their peers wait on the host);
            // RCCL peers are released by aborting the communicator below
            if (rc) { if (gc->kind == 1) (void)sf_comm_allreduce_sum(gc, nullptr, -1, (void*)st); continue; }
            rc = sf_comm_allreduce_sum(gc, (void*)(p->d_x + R.off), R.cnt, (void*)st);
        }
        if (!rc) sf_solve_step_fwd(p, k, fwd_base, sync, tickets, st);
    }
    if (rc) abort_comm(comm);
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    sf::launch_solve_transpose_diag(p->d_solve, p->d_solveT_list, p->n_solveT, bwd_base, p->d_solveT, st);      // see sf_chol_plan_solve
    for (size_t k = nst; k-- > 0;) sf_solve_step_bwd(p, k, bwd_base, sync, tickets, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(xb.data(), p->d_x, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    int sinfo = 0;
    if (p->d_solve_sync) HIP_TRY(hipMemcpyAsync(&sinfo, p->d_solve_sync, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sinfo) return SF_ERR_HIP;
    for (const auto& r : p->solve_own)
        memcpy(x_host + r.first, xb.data() + r.first, (size_t)(r.second - r.first) * sizeof(double));
    return SF_OK;
}

int sf_chol_plan_factorize_distributed(sf_chol_plan* p, sf_comm* comm, sf_float* host_out, int sync) {
    if (!p || !comm || comm->nranks != p->nranks || comm->rank != p->rank) return SF_ERR_ARG;
    if (p->nranks == 1) {
        if (host_out) return SF_ERR_ARG;        // single rank: sf_chol_plan_factorize_to_host
        return sf_chol_plan_factorize(p, sync);
    }
    int rc = SF_OK;
    // the groups of a proportionally mapped plan: their sub-communicators are made once, by all ranks together
    if ((rc = sf_comm_prepare_groups(comm, p->all_masks.data(), (int)p->all_masks.size()))) return rc;
    bool dl = false;
    if (host_out) { rc = sf_dl_begin(p, host_out); dl = rc == SF_OK; }
    if (!rc) rc = sf_chol_plan_factorize_phase(p, 0, 0);
    if (!rc && injected(p->rank, 1)) rc = SF_ERR_HIP;
    // The ranks agree on their state before the first data collective (one 8-byte sum on the second stream, which has nothing in
    // front of it: the own subtrees enqueued above keep running meanwhile).  A rank whose copy workers or phase 0 could not be
    // set up stops HERE, and so do its peers -- instead of enqueueing all-reduces that would wait for it on the GPU for ever.
    // (a rank without shared segments has no second stream: its own one then)
    if ((rc = agree_status(p, comm, rc, p->stream2 ? p->stream2 : p->stream))) {
        if (dl) (void)sf_dl_end(p);
        (void)sf_chol_plan_sync(p);
        return rc;
    }
    const sf_long nseg = sf_chol_plan_num_segments(p);
    // Software pipeline over the segments: a segment's sum is issued on the plan's SECOND stream -- for a look-ahead segment
    // (Segment::early) one segment ahead of its use, so the collective of block J+1 travels while the chain of block J runs on the
    // main stream -- and the main stream picks the sums up in order.  Every rank of a group issues the group's collectives in the
    // same order (the order of the segment list), on one stream.
    std::vector<char> begun(nseg, 0);
    auto begin = [&](sf_long k) -> int {
        begun[k] = 1;
        sf_comm* gc = group_comm(comm, p->segments[k].mask);
        if (!gc) return SF_ERR_ARG;
        void* buf = nullptr;
        sf_long cnt = 0;
        int r = sf_seg_begin(p, k, &buf, &cnt);
        // emulated ranks hand-shake on the host: a failed rank keeps taking part so that the others return too
        if (r) { if (gc->kind == 1) (void)sf_comm_allreduce_sum(gc, nullptr, -1, sf_plan_stream2(p)); return r; }
        r = sf_comm_allreduce_sum(gc, buf, cnt, sf_plan_stream2(p));
        return r ? r : sf_seg_reduced(p, k);
    };
    for (sf_long k = 0; k < nseg; ++k) {
        if (!rc && k == nseg / 2 && injected(p->rank, 2)) rc = SF_ERR_HIP;
        if (rc) {
            sf_comm* gc = group_comm(comm, p->segments[k].mask);
            if (!begun[k]) {
                if (gc && gc->kind == 1) (void)sf_comm_allreduce_sum(gc, nullptr, -1, sf_plan_stream2(p));
                begun[k] = 1;
            }
            if (sf_seg_is_owner_segment(p, k) && gc && gc->kind == 1) (void)sf_comm_allreduce_sum(gc, nullptr, -1, (void*)p->stream);   // its broadcast
            continue;
        }
        if (!begun[k]) rc = begin(k);
        if (!rc && k + 1 < nseg && !begun[k + 1] && sf_seg_early(p, k + 1)) rc = begin(k + 1);
        if (!rc) rc = sf_seg_finish(p, k);
        if (sf_seg_is_owner_segment(p, k)) {
            // owner-computes prototype (SF_TOP_OWNER=1): only the block's owner has run its chain; its finished block column now
            // travels to the group as a sum whose other terms are zero, on the MAIN stream (the far GEMMs behind it read the block)
            sf_comm* gc = group_comm(comm, p->segments[k].mask);
            void* buf = nullptr;
            sf_long cnt = 0;
            if (!rc) rc = gc ? sf_seg_bcast_begin(p, k, &buf, &cnt) : SF_ERR_ARG;
            if (rc) { if (gc && gc->kind == 1) (void)sf_comm_allreduce_sum(gc, nullptr, -1, (void*)p->stream); continue; }
            rc = sf_comm_allreduce_sum(gc, buf, cnt, (void*)p->stream);
            if (rc && getenv("SF_TRACE")) fprintf(stderr, "[sparseframe-hip] rank %d: broadcast of segment %lld failed (%d)\n", p->rank, (long long)k, rc);
            if (!rc) rc = sf_seg_bcast_finish(p, k);
            if (rc && getenv("SF_TRACE")) fprintf(stderr, "[sparseframe-hip] rank %d: segment %lld after its broadcast failed (%d)\n", p->rank, (long long)k, rc);
        }
    }
    int rc_dl = SF_OK;
    if (host_out) rc_dl = sf_dl_end(p);         // (also releases the copy workers when rc != 0)
    if (rc) {
        // in the middle of the segments: emulated peers were kept going by the hand-shakes above; this rank's RCCL communicator is
        // aborted and marked dead (later calls fail at once) -- peers already inside a collective wait for their own timeout, see abort_comm
        abort_comm(comm);
        (void)sf_chol_plan_sync(p);
        return rc;
    }
    if (sync || host_out) {
        const int rs = sf_chol_plan_sync(p);
        if ((rs || rc_dl) && getenv("SF_TRACE")) fprintf(stderr, "[sparseframe-hip] rank %d: sync %d, copy-back %d\n", p->rank, rs, rc_dl);
        return rs ? rs : rc_dl;
    }
    return SF_OK;
}

// test hook: the next distributed factorization (where 1, 2) / solve (3, 4) fails on `rank` at that point, once
void sf_test_inject_failure(int rank, int where) {
    g_fail_rank.store(rank);
    g_fail_where.store(where);
}

}  // extern "C"
