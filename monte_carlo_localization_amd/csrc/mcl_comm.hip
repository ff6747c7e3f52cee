// mcl_comm.hip -- mcl_comm_*: one process per GPU, the collectives of a sharded update over RCCL on the engine's own stream.
#include "mcl_engine_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>

using namespace mcl_host;

extern "C" {

// ---------------------------------------------------------------------------------------------
// One process per GPU, the exchange in native code: the engine holds an RCCL communicator and one call runs a whole sharded
// update -- the three collectives of an update (all-gather of the compact parent lists, all-reduce MAX of one double,
// all-reduce SUM of 5 + 3 G + 1 doubles) are enqueued on the ENGINE'S OWN STREAM between its kernels: no second stream, no
// event hop, no interpreter between the stages; the host waits once, for the summed vector.  RCCL is taken from the process
// at run time (dlopen: the library a torch process already carries, else the ROCm one): the engine does not link it, and a
// host without RCCL keeps every other entry point.  The rendezvous (128-byte id from rank 0 to every rank) is the host's.
// ---------------------------------------------------------------------------------------------
// The few RCCL types and enumerators the entry points below need, declared here: the library is taken with dlopen at run time
// and must build on a ROCm tree without the RCCL development headers (values as in rccl/rccl.h: they are RCCL's / NCCL's ABI).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclChar = 0, ncclUint64 = 5, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
}

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

static RcclApi &rccl_api()
{
    static RcclApi api;
    if (api.lib || !api.why.empty()) return api;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *nm : names) {
        api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);        // the copy the process already has (a torch process: torch's)
        if (api.lib) break;
    }
    for (const char *nm : names) {
        if (api.lib) break;
        api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    }
    if (!api.lib) { api.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return api; }
    auto sym = [&](const char *n) { void *p = dlsym(api.lib, n); if (!p && api.why.empty()) api.why = std::string("RCCL symbol missing: ") + n; return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.why.empty()) api.lib = nullptr;
    return api;
}

struct mcl_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0;
    unsigned char *d_chunk_local = nullptr, *d_chunk_all = nullptr;     // the lists: this shard's chunk, every shard's
    size_t chunk_capacity = 0;                                           // entries per chunk the buffers hold
    double *d_red = nullptr;                                             // [0] MAX exchange | [1 .. k] SUM exchange | [1 + k] error word (ranks that failed)
    double *h_red = nullptr;                                             // pinned copy of it
    uint64_t bytes_received = 0, bytes_payload = 0;                      // of the last update's list exchange
    int host_waits = 0;
    // dense exchange (an update without lists: the first after the particles were set): every shard's fixed-point weights,
    // their global CDF and every shard's packed records; allocated when first needed
    uint64_t *d_qall = nullptr, *d_cdfall = nullptr;
    double4 *d_recall = nullptr;
    size_t dense_capacity = 0;                                           // particles (all shards) the three arrays hold
    bool last_dense = false, last_kept = false;
    bool vec_valid = false;                                              // vec holds the sums of an update of the current particle set
    uint64_t dense_weights_bytes = 0, dense_records_bytes = 0;
    // what the shards' lists look like (the previous update's summed vector, or mcl_comm_set_lists after a dense update)
    bool lists_known = false;
    int64_t counts[mcl::kMaxShards] = {};
    uint64_t totals[mcl::kMaxShards] = {};
    double vec[5 + 3 * mcl::kMaxShards + 2] = {};                        // the last summed vector
    // failure protocol (mcl_comm_update): how long a host wait may last before the communicator is aborted; the update count;
    // MCL_COMM_FAIL = "<rank>:<update>:<stage>" (test switch: that rank reports a failure / stalls before that stage of that update)
    double timeout_ms = 30000.0;
    bool dead = false;                                                   // aborted (a wait ran out, or a collective call failed): create again
    unsigned long long updates = 0;
    int fail_rank = -1; long long fail_update = -1; std::string fail_stage;
    // the lists of the NEXT update, gathered right after this update's sums (beside whatever the host does between updates)
    bool gathered = false;
    unsigned long long gathered_epoch = 0;
    unsigned long long gathered_epoch_all = 0;                           // the update (count) after which EVERY rank pre-gathered the lists
    int64_t gathered_entries = 0;
    int64_t gathered_counts[mcl::kMaxShards] = {};
};

}  // extern "C"

// (declared in mcl_engine_internal.h: the engine calls them when the particle set changes / the engine is destroyed)
void comm_forget(mcl_comm *c)
{
    if (!c) return;
    c->lists_known = false;
    c->gathered = false;
    c->gathered_entries = 0;
    c->vec_valid = false;
}

void comm_free(mcl_comm *c)
{
    if (!c) return;
    if (c->comm && rccl_api().CommDestroy) (void)rccl_api().CommDestroy(c->comm);        // (an aborted communicator is gone already: comm == nullptr)
    if (c->d_chunk_local) (void)hipFree(c->d_chunk_local);
    if (c->d_chunk_all) (void)hipFree(c->d_chunk_all);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->h_red) (void)hipHostFree(c->h_red);
    if (c->d_qall) (void)hipFree(c->d_qall);
    if (c->d_cdfall) (void)hipFree(c->d_cdfall);
    if (c->d_recall) (void)hipFree(c->d_recall);
    delete c;
}

extern "C" {

// ---- failure protocol of a sharded update --------------------------------------------------------------------------------------
// A rank that fails must not hang its peers (the reference's error model is log-and-skip-a-tick, cpp:236-240, 680, 756; a node
// blocked inside a collective for ever is not that).  Two layers:
//   * SOFT failure -- anything a rank finds wrong locally while its communicator still works (an engine stage that fails, list
//     state that does not match the exchange, MCL_COMM_FAIL): the rank stops its local work but STILL ISSUES EVERY COLLECTIVE of
//     the update, with the sizes every rank derives from the shared numbers, and raises the ERROR WORD that rides behind the
//     summed vector (one more double of the SUM all-reduce: the number of ranks that failed).  Every rank then returns from the
//     same update: the failing ones with their own status and message, the others with MCL_ERR_PEER.  The communicator stays
//     usable; the particle set must be set or initialised again on every rank (as after MCL_ERR_HIP from mcl_update), and the
//     next update is a dense one.
//   * HARD failure -- a collective call that fails, or a host wait that lasts longer than MCL_COMM_TIMEOUT_MS (default 30 000;
//     a peer that died or never arrived): the communicator is ABORTED (ncclCommAbort: the collective kernels in flight end), the
//     call returns MCL_ERR_TIMEOUT / MCL_ERR_HIP, and mcl_comm_create must run again (on every rank: their waits run out too).
// Nothing here re-executes a process; recovery that needs a new process is the host's (a fresh child, never an exec of a
// process that has touched the GPU).
static void comm_abort(mcl_engine_t *h, const char *why)
{
    mcl_comm *c = h->comm;
    if (!c) return;
    RcclApi &api = rccl_api();
    if (c->comm) {
        if (api.CommAbort) (void)api.CommAbort(c->comm);
        // (without ncclCommAbort in the library the communicator is leaked rather than destroyed: ncclCommDestroy waits for peers)
        c->comm = nullptr;
    }
    c->dead = true;
    comm_forget(c);
    (void)hipStreamSynchronize(h->stream);          // the aborted kernels and whatever else was enqueued drain
    (void)why;
}

#define NCCLCHK(h, call)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) {                                                                                  \
            const std::string m_ = std::string(#call) + ": " + rccl_api().GetErrorString(r_) + " (communicator aborted: mcl_comm_create again)"; \
            comm_abort(h, "collective call failed");                                                              \
            return fail(h, MCL_ERR_HIP, m_);                                                                      \
        }                                                                                                         \
    } while (0)

// the host wait of a sharded update, bounded: MCL_OK when the engine's stream has drained, else the communicator is aborted
static int comm_wait(mcl_engine_t *h)
{
    mcl_comm *c = h->comm;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) { c->host_waits += 1; return MCL_OK; }
        if (e != hipErrorNotReady) {
            const std::string m = std::string("hipStreamQuery: ") + hipGetErrorString(e);
            comm_abort(h, "stream error");
            return fail(h, MCL_ERR_HIP, m);
        }
        if ((spins & 63u) == 63u) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > c->timeout_ms) break;
        }
    }
    comm_abort(h, "timeout");
    return fail(h, MCL_ERR_TIMEOUT, "a collective of the sharded update did not finish within MCL_COMM_TIMEOUT_MS (a peer failed or never arrived): "
                                    "communicator aborted, mcl_comm_create again");
}

int mcl_comm_available(const char **why)
{
    RcclApi &api = rccl_api();
    if (why) *why = api.lib ? "" : api.why.c_str();
    return api.lib ? MCL_OK : MCL_ERR_UNSUPPORTED;
}

int mcl_comm_unique_id(unsigned char id[128])
{
    RcclApi &api = rccl_api();
    if (!id) return MCL_ERR_INVALID_ARG;
    if (!api.lib) return MCL_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return MCL_ERR_HIP;
    std::memcpy(id, &u, 128);
    return MCL_OK;
}

int mcl_comm_create(mcl_engine_t *h, const unsigned char id[128], int32_t n_ranks, int32_t rank)
{
    if (!h || !id || n_ranks <= 0 || n_ranks > mcl::kMaxShards || rank < 0 || rank >= n_ranks) return MCL_ERR_INVALID_ARG;
    RcclApi &api = rccl_api();
    if (!api.lib) return fail(h, MCL_ERR_UNSUPPORTED, api.why);
    if (h->cfg.weight_mode != MCL_WEIGHT_LOG)
        return fail(h, MCL_ERR_UNSUPPORTED, "a sharded set needs weight_mode LOG");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->comm) { comm_free(h->comm); h->comm = nullptr; }
    mcl_comm *c = new mcl_comm();
    c->n_ranks = n_ranks; c->rank = rank;
    if (const char *e = getenv("MCL_COMM_TIMEOUT_MS")) { const double v = atof(e); if (v > 0.0) c->timeout_ms = v; }
    if (const char *e = getenv("MCL_COMM_FAIL")) {                    // "<rank>:<update>:<stage>", stage = resample | rays | weights | stall
        int fr = -1; long long fu = -1; char st[32] = {0};
        if (std::sscanf(e, "%d:%lld:%31s", &fr, &fu, st) == 3) { c->fail_rank = fr; c->fail_update = fu; c->fail_stage = st; }
    }
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    const ncclResult_t r = api.CommInitRank(&c->comm, n_ranks, u, rank);          // collective: every rank is in this call
    if (r != ncclSuccess) { c->comm = nullptr; comm_free(c); return fail(h, MCL_ERR_HIP, std::string("ncclCommInitRank: ") + api.GetErrorString(r)); }
    const size_t words = 1 + 5 + 3 * (size_t)n_ranks + 2 + 1;                     // MAX | summed vector | error word
    if (hipMalloc(&c->d_red, words * 8) != hipSuccess || hipHostMalloc(&c->h_red, words * 8) != hipSuccess) {
        comm_free(c);
        return fail(h, MCL_ERR_HIP, "mcl_comm_create: allocation failed");
    }
    h->comm = c;
    return MCL_OK;
}

// The three collectives of an update on known data, before any particle depends on them: all-reduce MAX of the rank, all-reduce
// SUM of ones, all-gather of one 64-byte chunk per rank.  COLLECTIVE.  A host that finds a rank failing here keeps its other
// exchange (dist.py: torch's collectives) instead of learning it in the first update.
int mcl_comm_selftest(mcl_engine_t *h)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    mcl_comm *c = h->comm;
    if (!c) return fail(h, MCL_ERR_NOT_READY, "mcl_comm_create first");
    if (c->dead) return fail(h, MCL_ERR_NOT_READY, "the communicator was aborted: mcl_comm_create again");
    RcclApi &api = rccl_api();
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int G = c->n_ranks;
    unsigned char *d_buf = nullptr;
    HIPCHK(h, hipMalloc(&d_buf, 64 * (size_t)(G + 1)));
    unsigned char mine[64];
    for (int i = 0; i < 64; ++i) mine[i] = (unsigned char)(c->rank * 7 + i);
    double two[2] = {(double)c->rank, 1.0};
    auto run = [&]() -> int {
        HIPCHK(h, hipMemcpyAsync(d_buf, mine, 64, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(c->d_red, two, 16, hipMemcpyHostToDevice, h->stream));
        NCCLCHK(h, api.AllReduce(c->d_red, c->d_red, 1, ncclDouble, ncclMax, c->comm, h->stream));
        NCCLCHK(h, api.AllReduce(c->d_red + 1, c->d_red + 1, 1, ncclDouble, ncclSum, c->comm, h->stream));
        NCCLCHK(h, api.AllGather(d_buf, d_buf + 64, 64, ncclChar, c->comm, h->stream));
        std::vector<unsigned char> all(64 * (size_t)G);
        HIPCHK(h, hipMemcpyAsync(c->h_red, c->d_red, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(all.data(), d_buf + 64, all.size(), hipMemcpyDeviceToHost, h->stream));
        const int rw = comm_wait(h);
        if (rw) return rw;
        if (c->h_red[0] != (double)(G - 1) || c->h_red[1] != (double)G) return fail(h, MCL_ERR_HIP, "mcl_comm_selftest: an all-reduce returned a wrong value");
        for (int r = 0; r < G; ++r)
            for (int i = 0; i < 64; ++i)
                if (all[(size_t)r * 64 + i] != (unsigned char)(r * 7 + i)) return fail(h, MCL_ERR_HIP, "mcl_comm_selftest: the all-gather returned wrong bytes");
        return MCL_OK;
    };
    const int rc = run();
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d_buf);
    return rc;
}

int mcl_comm_destroy(mcl_engine_t *h)
{
    if (!h) return MCL_ERR_INVALID_ARG;
    if (h->comm) {
        (void)hipSetDevice(h->cfg.device);
        (void)hipStreamSynchronize(h->stream);
        comm_free(h->comm);
        h->comm = nullptr;
    }
    return MCL_OK;
}

// What a rank found wrong locally during one sharded update (the first thing): its local work stops, its collectives go on.
struct CommLocal {
    int code = MCL_OK;
    std::string msg;
    bool bad() const { return code != MCL_OK; }
    void note(mcl_engine_t *h, int rc) { if (rc != MCL_OK && code == MCL_OK) { code = rc; msg = h->err; } }
};

// MCL_COMM_FAIL: does this rank fail (or stall) before `stage` of this update?
static bool comm_injected(const mcl_comm *c, const char *stage)
{
    return c->fail_rank == c->rank && c->fail_update == (long long)c->updates && c->fail_stage == stage;
}

// rays -> local max -> all-reduce MAX -> weights, scan, list -> this shard's part of the sums -> all-reduce SUM (with the error
// word) -> pinned host.  Returns a HARD failure, or MCL_OK with *failed_ranks = the summed error word (0: the sums are good).
static int comm_rays_to_sums(mcl_engine_t *h, const float *obs, int32_t n_beams, bool sync_rays, CommLocal &loc, double *failed_ranks)
{
    mcl_comm *c = h->comm;
    RcclApi &api = rccl_api();
    const size_t k = 5 + 3 * (size_t)c->n_ranks + 2;
    if (!loc.bad() && comm_injected(c, "rays")) loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the ray stage (MCL_COMM_FAIL)"));
    if (!loc.bad()) loc.note(h, stage_rays_launch(h, obs, n_beams, false, sync_rays ? nullptr : c->d_red));
    if (!loc.bad() && sync_rays) {        // after an overflow: wait, let the synchronous stage fall back to the self-contained kernel
        loc.note(h, stage_rays_finish(h, obs, n_beams));
        c->host_waits += 1;
        if (!loc.bad()) launch_copy_double(h->stream, h->d_scalars, c->d_red);
    }
    if (loc.bad()) launch_set_double(h->stream, c->d_red, -INFINITY);      // (any finite-or-not value will do: the update is void)
    NCCLCHK(h, api.AllReduce(c->d_red, c->d_red, 1, ncclDouble, ncclMax, c->comm, h->stream));
    if (!loc.bad() && comm_injected(c, "weights")) loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the weights stage (MCL_COMM_FAIL)"));
    if (!loc.bad()) loc.note(h, stage_weights_launch(h, 0.0, c->d_red));
    if (!loc.bad()) {
        launch_stage_pack(h->stream, h->d_result, c->d_red + 1, c->n_ranks, c->rank, h->compact_pending ? 1 : 0, (unsigned long long)h->compact_cap);
        if (hipGetLastError() != hipSuccess) loc.note(h, fail(h, MCL_ERR_HIP, "k_stage_pack launch failed"));
    }
    if (loc.bad() && hipMemsetAsync(c->d_red + 1, 0, k * 8, h->stream) != hipSuccess) { comm_abort(h, "memset"); return fail(h, MCL_ERR_HIP, "hipMemsetAsync failed (communicator aborted)"); }
    if (comm_injected(c, "stall"))        // test switch: this rank's stream is busy for ~3 x the bound before the last collective (a peer that does not arrive)
        launch_spin_ms(h->stream, std::min(3.0 * c->timeout_ms, 2000.0));
    launch_set_double(h->stream, c->d_red + 1 + k, loc.bad() ? 1.0 : 0.0);
    NCCLCHK(h, api.AllReduce(c->d_red + 1, c->d_red + 1, k + 1, ncclDouble, ncclSum, c->comm, h->stream));
    if (hipMemcpyAsync(c->h_red, c->d_red, (1 + k + 1) * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { comm_abort(h, "memcpy"); return fail(h, MCL_ERR_HIP, "hipMemcpyAsync failed (communicator aborted)"); }
    const int rw = comm_wait(h);                                      // THE host wait of the update
    if (rw) return rw;
    *failed_ranks = c->h_red[1 + k];
    if (*failed_ranks != 0.0 || loc.bad()) return MCL_OK;
    unpack_result(h);
    stage_rays_note(h);
    h->timings[4] = elapsed(h->ev[EV_RAYS], h->ev[EV_SENSOR]);
    loc.note(h, layout_adopt(h, h->N));                               // (after the last collective: a failure here is this rank's alone)
    return MCL_OK;
}

// export of this shard's list + all-gather of the chunks, on the engine's stream
static int comm_gather_lists(mcl_engine_t *h, const int64_t *counts, CommLocal &loc)
{
    mcl_comm *c = h->comm;
    RcclApi &api = rccl_api();
    const int G = c->n_ranks;
    int64_t longest = 0, listed = 0;
    for (int r = 0; r < G; ++r) { longest = std::max(longest, counts[r]); listed += counts[r]; }
    const int64_t entries = std::max<int64_t>(64, (longest + 63) & ~(int64_t)63);
    c->gathered = false;
    if ((size_t)entries > c->chunk_capacity) {
        // (without buffers of the agreed size this rank cannot take part in the all-gather: a hard failure)
        bool ok = hipStreamSynchronize(h->stream) == hipSuccess;
        if (c->d_chunk_local) (void)hipFree(c->d_chunk_local);
        if (c->d_chunk_all) (void)hipFree(c->d_chunk_all);
        c->d_chunk_local = c->d_chunk_all = nullptr; c->chunk_capacity = 0;
        const size_t cap = (size_t)entries + (size_t)entries / 4;            // lists breathe from update to update
        ok = ok && hipMalloc(&c->d_chunk_local, cap * 44) == hipSuccess && hipMalloc(&c->d_chunk_all, cap * 44 * (size_t)G) == hipSuccess;
        if (!ok) { comm_abort(h, "alloc"); return fail(h, MCL_ERR_HIP, "mcl_comm_update: no memory for the list exchange (communicator aborted)"); }
        c->chunk_capacity = cap;
    }
    if (!loc.bad()) loc.note(h, export_compact_launch(h, c->d_chunk_local, entries, h->cfg.device, h->stream));
    NCCLCHK(h, api.AllGather(c->d_chunk_local, c->d_chunk_all, (size_t)entries * 44, ncclChar, c->comm, h->stream));
    c->bytes_received = (uint64_t)entries * 44u * (uint64_t)(G - 1);
    c->bytes_payload = (uint64_t)(listed - counts[c->rank]) * 44u;
    c->gathered = !loc.bad(); c->gathered_epoch = h->list_epoch; c->gathered_entries = entries;
    for (int r = 0; r < G; ++r) c->gathered_counts[r] = counts[r];
    return MCL_OK;
}

// the shards' list lengths and weight totals out of a summed vector
static void comm_note_lists(mcl_comm *c, const double *vec)
{
    const int G = c->n_ranks;
    for (int r = 0; r < G; ++r) {
        c->counts[r] = (int64_t)vec[5 + 3 * r] - 1;
        c->totals[r] = ((uint64_t)vec[6 + 3 * r] + ((uint64_t)vec[7 + 3 * r] << 32));      // exact: halves < 2^32
    }
    c->lists_known = true;
}

int mcl_comm_set_lists(mcl_engine_t *h, const int64_t *counts, const uint64_t *totals)
{
    if (!h || !h->comm || !counts || !totals) return MCL_ERR_INVALID_ARG;
    mcl_comm *c = h->comm;
    for (int r = 0; r < c->n_ranks; ++r) { c->counts[r] = counts[r]; c->totals[r] = totals[r]; }
    c->lists_known = true;
    return MCL_OK;
}

// An update without lists (the first after the particles were set or initialised, or a shard whose list outgrew its arrays):
// every shard's fixed-point weights and packed records are gathered whole (8 + 32 B per particle of the other shards), every
// rank scans the same global CDF and draws its own children from it -- the same thresholds as every other path.
// (A set without any weight is noted as this rank's failure -- on every rank, they read the same total -- and the update goes
//  on through its collectives like any other void update.)
static int comm_resample_dense(mcl_engine_t *h, const double action[3], CommLocal &loc)
{
    mcl_comm *c = h->comm;
    RcclApi &api = rccl_api();
    const int G = c->n_ranks;
    const int64_t n = h->N, nt = n * G;
    if ((size_t)nt > c->dense_capacity) {
        bool ok = hipStreamSynchronize(h->stream) == hipSuccess;
        if (c->d_qall) (void)hipFree(c->d_qall);
        if (c->d_cdfall) (void)hipFree(c->d_cdfall);
        if (c->d_recall) (void)hipFree(c->d_recall);
        c->d_qall = c->d_cdfall = nullptr; c->d_recall = nullptr; c->dense_capacity = 0;
        ok = ok && hipMalloc(&c->d_qall, (size_t)nt * 8) == hipSuccess && hipMalloc(&c->d_cdfall, (size_t)nt * 8) == hipSuccess &&
             hipMalloc(&c->d_recall, (size_t)nt * sizeof(double4)) == hipSuccess;
        if (!ok) { comm_abort(h, "alloc"); return fail(h, MCL_ERR_HIP, "mcl_comm_update: no memory for the dense exchange (communicator aborted)"); }
        c->dense_capacity = (size_t)nt;
    }
    if (!loc.bad() && (size_t)nt / mcl::kScanTile + 2 > h->blocktot_capacity) {          // spine scratch of the scan, sized for one shard so far
        graph_reset(h);
        bool ok = hipStreamSynchronize(h->stream) == hipSuccess;
        dfree(h->d_blocktot);
        h->blocktot_capacity = 0;
        ok = ok && hipMalloc(&h->d_blocktot, ((size_t)nt / mcl::kScanTile + 2) * 8) == hipSuccess;
        if (ok) h->blocktot_capacity = (size_t)nt / mcl::kScanTile + 2;
        else loc.note(h, fail(h, MCL_ERR_HIP, "mcl_comm_update: no memory for the scan of the whole set"));
    }
    const int cur = h->cur;
    if (!loc.bad() && !h->pack_valid[cur]) {
        launch_pack_records(h->stream, h->d_x[cur], h->d_y[cur], h->d_th[cur], n, h->d_pack[cur]);
        h->pack_valid[cur] = true;
    }
    NCCLCHK(h, api.AllGather(h->d_q, c->d_qall, (size_t)n, ncclUint64, c->comm, h->stream));
    NCCLCHK(h, api.AllGather(h->d_pack[cur], c->d_recall, (size_t)n * sizeof(double4), ncclChar, c->comm, h->stream));
    if (!loc.bad()) loc.note(h, scan_weights(h, c->d_qall, c->d_cdfall, nt, 0, nullptr));
    // the draw needs the global fixed-point total on the host (a launch argument): one more wait, in an update that has no lists
    uint64_t q_total = 0;
    if (!loc.bad() && hipMemcpyAsync(&c->h_red[0], c->d_cdfall + (nt - 1), 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess)
        loc.note(h, fail(h, MCL_ERR_HIP, "hipMemcpyAsync of the weight total failed"));
    const int rw = comm_wait(h);
    if (rw) return rw;
    if (loc.bad()) return MCL_OK;
    std::memcpy(&q_total, &c->h_red[0], 8);
    if (q_total == 0) { loc.note(h, fail(h, MCL_ERR_NOT_READY, "the particle set carries no weight")); return MCL_OK; }
    ParentSource src;
    src.records = c->d_recall;
    c->last_dense = true;
    c->dense_weights_bytes = (uint64_t)n * 8u * (uint64_t)(G - 1);
    c->dense_records_bytes = (uint64_t)n * 32u * (uint64_t)(G - 1);
    if (comm_injected(c, "resample")) { loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the resampling stage (MCL_COMM_FAIL)")); return MCL_OK; }
    loc.note(h, stage_resample_launch(h, src, c->d_cdfall, nt, q_total, (int64_t)c->rank * n, nt, action));
    return MCL_OK;
}

int mcl_comm_update(mcl_engine_t *h, const double action[3], const float *obs, int32_t n_beams, double pose_out[3])
{
    if (!h || !action || !obs || !pose_out) return MCL_ERR_INVALID_ARG;
    mcl_comm *c = h->comm;
    // (the checks up to here must come out the same on every rank -- a host that calls with different arguments on different
    //  ranks has a bug no protocol repairs; what CAN differ between ranks goes through CommLocal below)
    if (!c) return fail(h, MCL_ERR_NOT_READY, "mcl_comm_create first");
    if (c->dead) return fail(h, MCL_ERR_NOT_READY, "the communicator was aborted: mcl_comm_create again");
    if (n_beams != h->B) return fail(h, MCL_ERR_INVALID_ARG, "bad observation");
    const int G = c->n_ranks;
    const int64_t n_per_shard = h->N;
    if (n_per_shard * G >= MCL_MAX_TOTAL_PARTICLES) return fail(h, MCL_ERR_INVALID_ARG, "particle total must stay below 2^27");
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(h, hipSetDevice(h->cfg.device));
    c->host_waits = 0;
    c->last_dense = false;
    c->updates += 1;
    CommLocal loc;
    // "particles not set on this rank" is a local condition: it goes through the protocol like any other
    if (!ready(h, true)) loc.note(h, fail(h, MCL_ERR_NOT_READY, "map, beam angles and particles must be set first"));
    const int64_t *counts = c->counts;
    const uint64_t *totals = c->totals;
    // Lists or not is decided from what EVERY rank knows alike (the previous update's summed vector): all ranks take the same branch.
    bool lists = c->lists_known && !h->env_comm_no_lists;
    uint64_t weight = 0;
    for (int r = 0; r < G && lists; ++r) {
        lists = counts[r] >= 0;
        weight += counts[r] > 0 ? totals[r] : 0ull;
    }
    lists = lists && weight != 0;
    int rc;
    // Adaptive resampling (E9): the set is kept when the effective sample size of the WHOLE set (the previous update's summed
    // vector: every rank has the same numbers) is at least r / 1000 of it -- no exchange at all then
    bool keep = false;
    if (h->cfg.resample_neff_permille > 0 && c->vec_valid && h->carry_valid) {
        const double sw = c->vec[0], sww = c->vec[5 + 3 * G + 1];
        keep = sww > 0.0 && sw * sw >= ((double)h->cfg.resample_neff_permille / 1000.0) * (double)(n_per_shard * G) * sww;
    }
    c->last_kept = keep;
    if (keep) {
        c->gathered = false;
        c->lists_known = false;
        if (!loc.bad()) loc.note(h, mcl_stage_keep(h, (int64_t)c->rank * n_per_shard, n_per_shard * G, action));
    } else if (lists) {
        if (!loc.bad() && counts[c->rank] != h->compact_n)
            loc.note(h, fail(h, MCL_ERR_NOT_READY, "this engine's list is not the one the last exchange described (state changed on one rank only?)"));
        // (1) the lists: already here when the previous update gathered them (same lists, same lengths), else now.  Whether they
        // were pre-gathered is the same on every rank: every rank gathers after a good update and forgets after a void one.
        bool have = c->gathered_entries > 0 && c->gathered_epoch_all == c->updates - 1;
        for (int r = 0; r < G && have; ++r) have = c->gathered_counts[r] == counts[r];
        if (have && !(c->gathered && c->gathered_epoch == h->list_epoch) && !loc.bad())
            loc.note(h, fail(h, MCL_ERR_NOT_READY, "the gathered lists are stale on this rank (particle state changed on one rank only?)"));
        if (!have) { rc = comm_gather_lists(h, counts, loc); if (rc) return rc; }
        const int64_t entries = c->gathered_entries;
        c->gathered = false;
        c->lists_known = false;                // (known again once this update's vector is here)
        if (!loc.bad() && comm_injected(c, "resample")) loc.note(h, fail(h, MCL_ERR_HIP, "injected failure before the resampling stage (MCL_COMM_FAIL)"));
        if (!loc.bad())
            loc.note(h, stage_resample_compact_launch(h, c->d_chunk_all, G, entries, counts, totals, n_per_shard, c->rank, (int64_t)c->rank * n_per_shard,
                                                      n_per_shard * G, action, nullptr));
    } else {
        c->gathered = false;
        c->lists_known = false;
        rc = comm_resample_dense(h, action, loc);
        if (rc) return rc;
    }
    // (2) + (3)
    double failed = 0.0;
    rc = comm_rays_to_sums(h, obs, n_beams, false, loc, &failed);
    if (rc) return rc;
    const size_t k = 5 + 3 * (size_t)G + 2;
    if (failed == 0.0 && !loc.bad() && c->h_red[1 + k - 2] != 0.0) {
        // some shard's fix-up lists overflowed (debug_force_exact at size, a pathological map): every rank once more from the ray stage on
        rc = comm_rays_to_sums(h, obs, n_beams, true, loc, &failed);
        if (rc) return rc;
    }
    if (failed != 0.0 || loc.bad()) {
        // a void update, on every rank alike: the lists and sums are forgotten, the particle set must be set or initialised again
        comm_forget(c);
        c->gathered_entries = 0;
        h->have_particles = false;
        if (loc.bad()) return fail(h, loc.code, loc.msg + " [sharded update void on every rank]");
        char m[160];
        std::snprintf(m, sizeof m, "%d rank(s) of the sharded set reported a failure in this update: it is void on every rank (set or initialise the particles again)", (int)failed);
        return fail(h, MCL_ERR_PEER, m);
    }
    for (size_t i = 0; i < k; ++i) c->vec[i] = c->h_red[1 + i];
    for (int i = 0; i < 5; ++i) h->global_sums[i] = c->vec[i];
    c->vec_valid = true;
    stage_commit_carry(h);
    comm_note_lists(c, c->vec);
    bool next_keeps = false;
    if (h->cfg.resample_neff_permille > 0) {
        const double sw = c->vec[0], sww = c->vec[5 + 3 * G + 1];
        next_keeps = sww > 0.0 && sw * sw >= ((double)h->cfg.resample_neff_permille / 1000.0) * (double)(n_per_shard * G) * sww;
    }
    // the lists this update wrote are final: gather them for the next update now, beside the host's work between updates
    // (every rank reads the same vector, so every rank takes the same decision)
    c->gathered_entries = 0;
    {
        bool all = true;
        uint64_t wsum = 0;
        for (int r = 0; r < G; ++r) { all = all && c->counts[r] >= 0; wsum |= c->counts[r] > 0 ? c->totals[r] : 0ull; }
        if (all && wsum != 0 && !h->env_comm_no_pregather && !next_keeps) {
            CommLocal pre;                         // (a local failure of the export shows in the next update: `gathered` stays false here)
            rc = comm_gather_lists(h, c->counts, pre);
            if (rc) return rc;
            c->gathered_epoch_all = c->updates;
        }
    }
    const double sw = c->vec[0], kk = sw > 0.0 ? 1.0 / sw : 1.0;          // expected_pose (cpp:702-716) over the whole set
    pose_out[0] = c->vec[1] * kk; pose_out[1] = c->vec[2] * kk; pose_out[2] = std::atan2(c->vec[3] * kk, c->vec[4] * kk);
    h->timings[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MCL_OK;
}

int mcl_comm_get_vector(const mcl_engine_t *h, double *vec_out, int32_t n)
{
    if (!h || !h->comm || !vec_out || n != 5 + 3 * h->comm->n_ranks + 2) return MCL_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) vec_out[i] = h->comm->vec[i];
    return MCL_OK;
}

int mcl_comm_last_exchange(const mcl_engine_t *h, int32_t *dense, uint64_t *weights_bytes, uint64_t *records_bytes)
{
    if (!h || !h->comm) return MCL_ERR_INVALID_ARG;
    if (dense) *dense = h->comm->last_kept ? 2 : h->comm->last_dense ? 1 : 0;       // 0 lists, 1 dense, 2 none (the set was kept)
    if (weights_bytes) *weights_bytes = h->comm->last_dense ? h->comm->dense_weights_bytes : 0;
    if (records_bytes) *records_bytes = h->comm->last_dense ? h->comm->dense_records_bytes : 0;
    return MCL_OK;
}

int mcl_comm_stats(const mcl_engine_t *h, uint64_t *list_bytes_received, uint64_t *list_payload_bytes, int32_t *host_waits)
{
    if (!h || !h->comm) return MCL_ERR_INVALID_ARG;
    if (list_bytes_received) *list_bytes_received = h->comm->bytes_received;
    if (list_payload_bytes) *list_payload_bytes = h->comm->bytes_payload;
    if (host_waits) *host_waits = h->comm->host_waits;
    return MCL_OK;
}

}  // extern "C"
