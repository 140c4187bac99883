// Library-side collectives (r04): the exchanges of a target-sharded run -- the per-round all-gather of directed neighbour entries
// (the role of the master's running graph, interleaved.jl:124-140,166-183) and the all-gather of the significant level-0 pairs --
// issued by the LIBRARY on a communicator of its own, on the context's stream, instead of by callbacks into the host language.
// r01-r03 ran every collective from Python (dist.py on torch.distributed); with the communicator inside the library an exchange is
// one ncclAllGather enqueued between two kernels of the engine's stream, which is what a collective inside a device round needs.
//
// RCCL is reached through dlopen (librccl.so.1: the copy a host process has already loaded -- torch ships one -- or ROCm's), so the
// library keeps loading on hosts without RCCL and single-GPU users never touch it.  The rendezvous is the caller's: rank 0 asks
// fw_comm_unique_id for the 128-byte id, ships it to the other ranks by whatever channel the host language has (torch.distributed
// broadcast in bench.py, Distributed.jl's remotecall in the Julia shim) and every rank calls fw_comm_init.
// One process per GPU; ncclCommInitRank refuses two ranks on one device, so the world-2 tests on ONE GPU keep the callback form
// (gloo) and this path is exercised there with a world of one rank.  NOT measured on 8 GPUs (no such node was available).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "fw_internal.h"

namespace {

// the handful of RCCL entry points used, with the ABI of rccl.h (NCCL 2.x: stable since 2.0)
typedef struct {
    char internal[128];
} fwNcclUniqueId;
typedef void *fwNcclComm;
enum { fwNcclChar = 0, fwNcclInt64 = 4 };  // ncclDataType_t: ncclInt8 / ncclChar = 0, ncclInt64 = 4
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(fwNcclUniqueId *) = nullptr;
    int (*CommInitRank)(fwNcclComm *, int, fwNcclUniqueId, int) = nullptr;
    int (*CommDestroy)(fwNcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, fwNcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string where;
};

RcclApi *rccl_api(std::string *err)
{
    static RcclApi api;
    static bool tried = false;
    static std::string load_err;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {  // a copy the process already holds first (one RCCL per process)
            api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (api.handle) {
                api.where = std::string(n) + " (already loaded)";
                break;
            }
        }
        for (size_t i = 0; !api.handle && i < sizeof(names) / sizeof(names[0]); ++i) {
            api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) api.where = names[i];
        }
        if (!api.handle) {
            load_err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?");
        } else {
            api.GetUniqueId = (int (*)(fwNcclUniqueId *))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(fwNcclComm *, int, fwNcclUniqueId, int))dlsym(api.handle, "ncclCommInitRank");
            api.CommDestroy = (int (*)(fwNcclComm))dlsym(api.handle, "ncclCommDestroy");
            api.AllGather = (int (*)(const void *, void *, size_t, int, fwNcclComm, hipStream_t))dlsym(api.handle, "ncclAllGather");
            api.GetErrorString = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
                load_err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
                api.handle = nullptr;
            }
        }
    }
    if (!api.handle) {
        if (err) *err = load_err;
        return nullptr;
    }
    return &api;
}

const char *rccl_err(RcclApi *R, int rc) { return R->GetErrorString ? R->GetErrorString(rc) : "?"; }

}  // namespace

// per-context communicator state (fw_ctx::comm)
struct FwComm {
    fwNcclComm comm = nullptr;
    int rank = 0, world = 1;
    FwDevBuf d_hdr, d_send, d_recv;  // header (2 x int64 per rank), payload buffers (grow-only)
    std::vector<int64_t> h_hdr;
    int64_t cap = 0;  // capacity of the payload buffers in bytes per rank (grow-only)
    int64_t cur = 0;  // this exchange's stride in records per rank: pow2(max over the ranks of this round's counts)
    int32_t rec = 0;
    // counters (bench.py reports them)
    int64_t calls = 0, collectives = 0, entries = 0, bytes = 0;
    double seconds = 0.0;
};

namespace {

double rc_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct CommX {
    fw_ctx *c;
    double t0;
};

// the callbacks return small non-zero codes to the exchange's caller; every one of them leaves a message in the context as well
template <typename... A>
int comm_fail(fw_ctx *c, int code, const char *fmt, A... a)
{
    (void)fw_fail(c, FW_ERR_DEVICE, fmt, a...);
    return code;
}

// fw_dev_exchange::prepare on the library's own communicator: header all-gather (record count + one auxiliary integer per rank),
// room for max(count) records per rank
int comm_prepare(void *user, int64_t n_local, int64_t aux_local, int32_t rec_bytes, void **d_send, void **d_recv, int64_t *counts, int64_t *aux,
                 int64_t *cap_records)
{
    CommX *X = (CommX *)user;
    fw_ctx *c = X->c;
    FwComm *K = c->comm;
    RcclApi *R = rccl_api(nullptr);
    if (!K || !K->comm || !R) return comm_fail(c, 1, "comm_prepare: no communicator (fw_comm_init)");
    X->t0 = rc_now();
    const int W = K->world;
    if (fw_dev_reserve(c, K->d_hdr, sizeof(int64_t) * 2 * (size_t)(W + 1))) return comm_fail(c, 2, "comm_prepare: no device memory for the header");
    int64_t *dh = (int64_t *)K->d_hdr.ptr;  // [0..2): this rank's header, [2..2 + 2 W): everybody's
    const int64_t mine[2] = {n_local, aux_local};
    if (hipMemcpyAsync(dh, mine, sizeof(mine), hipMemcpyHostToDevice, c->stream) != hipSuccess) return comm_fail(c, 3, "comm_prepare: header upload failed");
    int rc = R->AllGather(dh, dh + 2, 2, fwNcclInt64, K->comm, c->stream);
    if (rc) {
        fw_fail(c, FW_ERR_DEVICE, "ncclAllGather (header): %s", rccl_err(R, rc));
        return 4;
    }
    K->h_hdr.resize(2 * (size_t)W);
    if (hipMemcpyAsync(K->h_hdr.data(), dh + 2, sizeof(int64_t) * 2 * (size_t)W, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return comm_fail(c, 5, "comm_prepare: header download failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess) return comm_fail(c, 6, "comm_prepare: stream error after the header all-gather");
    int64_t mx = 1, tot = 0;
    for (int r = 0; r < W; ++r) {
        counts[r] = K->h_hdr[2 * (size_t)r];
        aux[r] = K->h_hdr[2 * (size_t)r + 1];
        if (counts[r] < 0) return comm_fail(c, 8, "comm_prepare: rank %d announced %lld records", r, (long long)counts[r]);
        mx = std::max(mx, counts[r]);
        tot += counts[r];
    }
    // the stride of THIS exchange is the round's own maximum (rounded up to a power of two), not the buffers' capacity: the
    // buffers only grow, and a level-0 exchange of a million pairs must not make every later round ship a million records per rank
    int64_t cur = 1;
    while (cur < mx) cur <<= 1;
    const int64_t need = cur * (int64_t)rec_bytes;
    if (K->cap < need) {
        if (fw_dev_reserve(c, K->d_send, (size_t)need)) return comm_fail(c, 7, "comm_prepare: no device memory for %lld bytes per rank", (long long)need);
        if (fw_dev_reserve(c, K->d_recv, (size_t)W * (size_t)need)) return comm_fail(c, 7, "comm_prepare: no device memory for %d x %lld bytes", W, (long long)need);
        K->cap = need;
    }
    K->cur = cur;
    K->rec = rec_bytes;
    *d_send = K->d_send.ptr;
    *d_recv = K->d_recv.ptr;
    *cap_records = cur;
    K->entries += tot;
    return 0;
}

int comm_exchange(void *user)
{
    CommX *X = (CommX *)user;
    fw_ctx *c = X->c;
    FwComm *K = c->comm;
    RcclApi *R = rccl_api(nullptr);
    if (!K || !K->comm || !R) return comm_fail(c, 1, "comm_exchange: no communicator (fw_comm_init)");
    // the library's pack kernels / copies into the send buffer ran on the context's stream or were synchronous: stream order suffices
    const size_t bytes = (size_t)K->cur * (size_t)K->rec;
    int rc = R->AllGather(K->d_send.ptr, K->d_recv.ptr, bytes, fwNcclChar, K->comm, c->stream);
    if (rc) {
        fw_fail(c, FW_ERR_DEVICE, "ncclAllGather (payload): %s", rccl_err(R, rc));
        return 2;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return comm_fail(c, 3, "comm_exchange: stream error after the payload all-gather");  // "returns when the data is in place" (fw_dev_exchange)
    K->calls += 1;
    K->collectives += 2;
    K->bytes += (int64_t)bytes * K->world;
    K->seconds += rc_now() - X->t0;
    return 0;
}

}  // namespace

void fwi_comm_free(fw_ctx *c)
{
    if (!c || !c->comm) return;
    FwComm *K = c->comm;
    RcclApi *R = rccl_api(nullptr);
    if (K->comm && R) (void)R->CommDestroy(K->comm);
    if (K->d_hdr.ptr) (void)hipFree(K->d_hdr.ptr);
    if (K->d_send.ptr) (void)hipFree(K->d_send.ptr);
    if (K->d_recv.ptr) (void)hipFree(K->d_recv.ptr);
    delete K;
    c->comm = nullptr;
}

extern "C" {

int fw_comm_unique_id(uint8_t *id128)
{
    if (!id128) return fw_fail(nullptr, FW_ERR_ARG, "fw_comm_unique_id: NULL output");
    std::string err;
    RcclApi *R = rccl_api(&err);
    if (!R) return fw_fail(nullptr, FW_ERR_DEVICE, "fw_comm_unique_id: %s", err.c_str());
    fwNcclUniqueId id;
    const int rc = R->GetUniqueId(&id);
    if (rc) return fw_fail(nullptr, FW_ERR_DEVICE, "ncclGetUniqueId: %s", rccl_err(R, rc));
    memcpy(id128, id.internal, sizeof(id.internal));
    return FW_OK;
}

int fw_comm_init(fw_ctx *c, const uint8_t *id128, int32_t rank, int32_t world_size)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    if (!id128 || world_size < 1 || rank < 0 || rank >= world_size) return fw_fail(c, FW_ERR_ARG, "fw_comm_init: bad id / rank %d of %d", rank, world_size);
    std::string err;
    RcclApi *R = rccl_api(&err);
    if (!R) return fw_fail(c, FW_ERR_DEVICE, "fw_comm_init: %s", err.c_str());
    fwi_comm_free(c);
    (void)hipSetDevice(c->P.device);
    FwComm *K = new FwComm();
    fwNcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    const int rc = R->CommInitRank(&K->comm, world_size, id, rank);
    if (rc) {
        delete K;
        return fw_fail(c, FW_ERR_DEVICE, "ncclCommInitRank(rank %d of %d): %s", rank, world_size, rccl_err(R, rc));
    }
    K->rank = rank;
    K->world = world_size;
    c->comm = K;
    return FW_OK;
}

int fw_comm_destroy(fw_ctx *c)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    fwi_comm_free(c);
    return FW_OK;
}

int fw_comm_stats(const fw_ctx *c, int64_t *calls, int64_t *collectives, int64_t *entries, int64_t *bytes, double *seconds)
{
    if (!c || !c->comm) return fw_fail(c, FW_ERR_STATE, "fw_comm_stats: no communicator (fw_comm_init)");
    if (calls) *calls = c->comm->calls;
    if (collectives) *collectives = c->comm->collectives;
    if (entries) *entries = c->comm->entries;
    if (bytes) *bytes = c->comm->bytes;
    if (seconds) *seconds = c->comm->seconds;
    return FW_OK;
}

int fw_level0_comm(fw_ctx *c, int64_t *nnz_out)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    if (!c->comm) return fw_fail(c, FW_ERR_STATE, "fw_level0_comm: no communicator (fw_comm_init)");
    CommX X{c, 0.0};
    fw_dev_exchange x{&X, comm_prepare, comm_exchange};
    return fw_level0_sharded_dev(c, c->comm->rank, c->comm->world, &x, nnz_out);
}

int fw_learn_network_comm(fw_ctx *c, const fw_learn_opts *opts_in, int64_t *n_edges_out)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    if (!c->comm) return fw_fail(c, FW_ERR_STATE, "fw_learn_network_comm: no communicator (fw_comm_init)");
    fw_learn_opts opt{};
    opt.feed_forward = 1;
    opt.round_size = 1;
    if (opts_in) opt = *opts_in;
    opt.rank = c->comm->rank;  // the communicator's, whatever the caller wrote
    opt.world_size = c->comm->world;
    CommX X{c, 0.0};
    fw_dev_exchange x{&X, comm_prepare, comm_exchange};
    return fw_learn_network_dev(c, &opt, &x, n_edges_out);
}

// FW_FZ row-block sharding (fw_use_cor_buffer / fw_compute_cor_mat_rows): the in-place all-gather of the row blocks
int fw_cor_mat_allgather_comm(fw_ctx *c, int64_t rows_per_rank)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    if (!c->comm) return fw_fail(c, FW_ERR_STATE, "fw_cor_mat_allgather_comm: no communicator (fw_comm_init)");
    if (!c->d_cor || !c->cor_external) return fw_fail(c, FW_ERR_STATE, "fw_cor_mat_allgather_comm: needs fw_use_cor_buffer + fw_compute_cor_mat_rows first");
    RcclApi *R = rccl_api(nullptr);
    FwComm *K = c->comm;
    const int64_t p64 = c->P.p;
    if (rows_per_rank <= 0 || p64 <= 0 || rows_per_rank > c->cor_capacity / p64 / K->world)
        return fw_fail(c, FW_ERR_ARG, "fw_cor_mat_allgather_comm: rows_per_rank %lld does not fit %d blocks into the buffer of %lld floats (p = %lld)",
                       (long long)rows_per_rank, K->world, (long long)c->cor_capacity, (long long)p64);
    // the in-place all-gather needs this rank's block where fw_compute_cor_mat_rows wrote it: at rank * rows_per_rank
    if (c->cor_rows_rank != K->rank || c->cor_rows_world != K->world || c->cor_rows_per_rank != rows_per_rank)
        return fw_fail(c, FW_ERR_STATE, "fw_cor_mat_allgather_comm: fw_compute_cor_mat_rows ran as rank %d of %d with %lld rows per rank, the communicator is rank %d of %d and %lld rows were passed",
                       c->cor_rows_rank, c->cor_rows_world, (long long)c->cor_rows_per_rank, K->rank, K->world, (long long)rows_per_rank);
    const size_t block = (size_t)rows_per_rank * (size_t)c->P.p;  // floats per rank
    if ((int64_t)(block * (size_t)K->world) > c->cor_capacity) return fw_fail(c, FW_ERR_ARG, "fw_cor_mat_allgather_comm: the buffer holds %lld floats, %d blocks of %zu need more", (long long)c->cor_capacity, K->world, block);
    const double t0 = rc_now();
    const int rc = R->AllGather(c->d_cor + (size_t)K->rank * block, c->d_cor, block * sizeof(float), fwNcclChar, K->comm, c->stream);
    if (rc) return fw_fail(c, FW_ERR_DEVICE, "ncclAllGather (matrix rows): %s", rccl_err(R, rc));
    FW_HIP(c, hipStreamSynchronize(c->stream));
    K->calls += 1;
    K->collectives += 1;
    K->bytes += (int64_t)(block * sizeof(float)) * K->world;
    K->seconds += rc_now() - t0;
    return FW_OK;
}

}  // extern "C"
