// Context lifecycle, data hand-over, level-0 host logic (Benjamini-Hochberg + neighbour lists) and the
// extern "C" entry points of include/flashweave_amd.h.  Device work lives in fw_fz.hip / fw_mi.hip,
// the HITON-PC host driver in fw_hiton.cpp.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <numeric>

#include <mutex>

#include "fw_internal.h"

static thread_local std::string g_create_err;

int fw_fail(const fw_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        // concurrent device-HITON chains (fw_hiton.cpp, one host thread each) may fail at the same time: the message is
        // a std::string, so writers are serialised; the first message of a call wins (fw_learn_network clears it)
        static std::mutex err_mu;
        std::lock_guard<std::mutex> lk(err_mu);
        ctx->err = buf;
    } else {
        g_create_err = buf;
    }
    return code;
}

int fw_dev_reserve(fw_ctx *ctx, FwDevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return FW_OK;
    if (b.ptr) FW_HIP(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 2 + 256;
    hipError_t e = hipMalloc(&b.ptr, want);
    if (e != hipSuccess) return fw_fail(ctx, FW_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return FW_OK;
}

int fw_pin_reserve(fw_ctx *ctx, FwPinned &b, size_t bytes)
{
    if (bytes <= b.cap) return FW_OK;
    if (b.ptr) FW_HIP(ctx, hipHostFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 2 + 256;
    hipError_t e = hipHostMalloc(&b.ptr, want, hipHostMallocDefault);
    if (e != hipSuccess) return fw_fail(ctx, FW_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return FW_OK;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" {

int fw_abi_version(void) { return FW_ABI_VERSION; }

void fw_params_default(fw_params *P, int32_t kind, int32_t n, int32_t p)
{
    if (!P) return;
    memset(P, 0, sizeof(*P));
    P->kind = kind;
    P->n = n;
    P->p = p;
    P->device = 0;
    P->max_k = 3;
    P->hps = 5;
    P->fdr = 1;
    P->n_obs_min = -1;
    P->max_tests = 10000000;
    P->alpha = 0.01;
    P->recursive_pcor = 1;
}

const char *fw_last_error(const fw_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int fw_ctx_create(const fw_params *P, fw_ctx **out)
{
    if (out) *out = nullptr;
    if (!P || !out) return fw_fail(nullptr, FW_ERR_ARG, "fw_ctx_create: NULL argument");
    if (P->kind != FW_MI && P->kind != FW_MI_NZ && P->kind != FW_FZ && P->kind != FW_FZ_NZ)
        return fw_fail(nullptr, FW_ERR_ARG, "fw_ctx_create: unknown test kind %d", P->kind);
    if (P->n <= 0 || P->p <= 1) return fw_fail(nullptr, FW_ERR_ARG, "fw_ctx_create: need n > 0 and p > 1 (n=%d, p=%d)", P->n, P->p);
    if (P->max_k < 0 || P->max_k > FW_MAX_K)
        return fw_fail(nullptr, FW_ERR_LIMIT, "fw_ctx_create: max_k=%d outside [0, %d]", P->max_k, FW_MAX_K);
    if (P->max_k > FW_MAX_K_FAST && (P->kind == FW_FZ || P->kind == FW_FZ_NZ) && !P->recursive_pcor)
        return fw_fail(nullptr, FW_ERR_LIMIT, "fw_ctx_create: max_k=%d with recursive_pcor = 0 (conditioning on job-local Gram matrices serves max_k <= %d)",
                       P->max_k, FW_MAX_K_FAST);
    if (!(P->alpha > 0.0 && P->alpha < 1.0)) return fw_fail(nullptr, FW_ERR_ARG, "fw_ctx_create: alpha must be in (0,1)");
    if (P->hps < 0) return fw_fail(nullptr, FW_ERR_ARG, "fw_ctx_create: hps must be >= 0");
    if (P->dense_rules && (P->kind == FW_FZ || P->kind == FW_FZ_NZ))
        return fw_fail(nullptr, FW_ERR_ARG, "fw_ctx_create: dense_rules applies to the discrete tests only");
    if (P->no_cor_mat && (P->kind != FW_FZ || P->recursive_pcor))
        return fw_fail(nullptr, FW_ERR_ARG, "fw_ctx_create: no_cor_mat (dense_cor = false) needs FW_FZ with recursive_pcor = 0 "
                                            "(without a matrix the conditional tests come from the data, tests.jl:253)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fw_fail(nullptr, FW_ERR_DEVICE, "fw_ctx_create: no HIP device available (%s); this engine has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (P->device < 0 || P->device >= ndev)
        return fw_fail(nullptr, FW_ERR_DEVICE, "fw_ctx_create: device %d out of range (have %d)", P->device, ndev);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, P->device)) != hipSuccess)
        return fw_fail(nullptr, FW_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fw_fail(nullptr, FW_ERR_DEVICE, "fw_ctx_create: device %d is %s; kernels are built for gfx950 only", P->device,
                       prop.gcnArchName);
    if ((e = hipSetDevice(P->device)) != hipSuccess) return fw_fail(nullptr, FW_ERR_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    fw_ctx *c = new (std::nothrow) fw_ctx();
    if (!c) return fw_fail(nullptr, FW_ERR_NOMEM, "out of host memory");
    c->P = *P;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        delete c;
        return fw_fail(nullptr, FW_ERR_DEVICE, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    for (FwPoolBuf &b : c->pb)
        if ((e = hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreate(&b.ev0)) != hipSuccess || (e = hipEventCreate(&b.ev1)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&b.evd, hipEventDisableTiming)) != hipSuccess) {
            fw_ctx_destroy(c);
            return fw_fail(nullptr, FW_ERR_DEVICE, "stream/event creation failed: %s", hipGetErrorString(e));
        }
    for (FwPoolBuf &b : c->pb) b.launch_stream = b.stream;
    // continuous: the automatic n_obs_min is known immediately (learning.jl:59-61); discrete needs levels
    c->n_obs_min_eff = P->n_obs_min >= 0 ? P->n_obs_min : ((P->kind == FW_FZ || P->kind == FW_FZ_NZ) ? 20 : -1);
    *out = c;
    return FW_OK;
}

static void free_dev(FwDevBuf &b)
{
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.cap = 0;
}
static void free_pin(FwPinned &b)
{
    if (b.ptr) (void)hipHostFree(b.ptr);
    b.ptr = nullptr;
    b.cap = 0;
}

int fw_ctx_destroy(fw_ctx *c)
{
    if (!c) return FW_OK;
    (void)hipSetDevice(c->P.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    fwi_comm_free(c);
    fwi_host_workers_free(c);
#ifdef FW_FZ_FASTDBG
    fwi_fz_fastdbg_print();
#endif
    if (c->cor_external) c->d_cor = nullptr;  // caller-owned (fw_use_cor_buffer)
    void *ptrs[] = {c->d_data, c->d_xc, c->d_sd, c->d_cor, c->d_thr, c->d_fzs_stat, c->d_nzbits, c->d_hibits, c->d_levels, c->d_maxvals, c->d_firstnz, c->d_xlnx, c->d_gthr, c->d_vals};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    free_dev(c->d_jobs);
    free_dev(c->d_acc);
    free_dev(c->d_out);
    free_dev(c->d_tmp0);
    free_dev(c->d_tmp1);
    free_dev(c->d_tmp2);
    free_dev(c->d_segs);
    free_dev(c->d_segout);
    free_dev(c->d_nzrecs);
    free_dev(c->d_arena);
    free_dev(c->d_bh);
    free_dev(c->d_l0m_i);
    free_dev(c->d_l0m_d);
    for (FwDevBuf &b : c->d_mig_tab) free_dev(b);
    for (int q = 0; q < FW_DH_MAX_CHAINS; ++q) {
        free_dev(c->d_dh[q]);
        free_pin(c->h_dh[q]);
        if (c->dh_stream[q]) (void)hipStreamDestroy(c->dh_stream[q]);
        if (c->dh_hp_stream[q]) (void)hipStreamDestroy(c->dh_hp_stream[q]);
        for (int e = 0; e < 2; ++e)
            if (c->dh_hp_ev[q][e]) (void)hipEventDestroy(c->dh_hp_ev[q][e]);
    }
    free_pin(c->h_jobs);
    free_pin(c->h_acc);
    free_pin(c->h_out);
    for (FwPoolBuf &b : c->pb) {
        if (b.stream) (void)hipStreamSynchronize(b.stream);
        free_pin(b.h_in);
        free_pin(b.h_out);
        free_dev(b.d_in);
        free_dev(b.d_out);
        free_dev(b.d_acc);
        if (b.ev0) (void)hipEventDestroy(b.ev0);
        if (b.ev1) (void)hipEventDestroy(b.ev1);
        if (b.evd) (void)hipEventDestroy(b.evd);
        if (b.stream) (void)hipStreamDestroy(b.stream);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FW_OK;
}

#define CHECK_CTX(c)                                          \
    do {                                                      \
        if (!(c)) return fw_fail(nullptr, FW_ERR_ARG, "NULL context"); \
        (void)hipSetDevice((c)->P.device);                    \
    } while (0)

int fw_set_data_dense_f32(fw_ctx *c, const float *data)
{
    CHECK_CTX(c);
    if (c->P.kind != FW_FZ && c->P.kind != FW_FZ_NZ) return fw_fail(c, FW_ERR_ARG, "fw_set_data_dense_f32: context is not FW_FZ / FW_FZ_NZ");
    if (!data) return fw_fail(c, FW_ERR_ARG, "fw_set_data_dense_f32: NULL data");
    if (c->P.kind == FW_FZ_NZ) {
        int rc = fwi_fznz_upload(c, data);
        if (rc) return rc;
        c->have_data = true;
        c->have_level0 = false;
        c->have_network = false;
        return FW_OK;
    }
    const size_t bytes = sizeof(float) * (size_t)c->P.n * c->P.p;
    if (!c->d_data) FW_HIP(c, hipMalloc(&c->d_data, bytes));
    FW_HIP(c, hipMemcpy(c->d_data, data, bytes, hipMemcpyHostToDevice));
    c->have_data = true;
    c->have_fzs_stat = false;
    c->have_cor = false;
    c->have_level0 = false;
    c->have_network = false;
    return FW_OK;
}

int fw_set_cor_mat(fw_ctx *c, const float *cor)
{
    CHECK_CTX(c);
    if (c->P.kind != FW_FZ) return fw_fail(c, FW_ERR_ARG, "fw_set_cor_mat: context is not FW_FZ");
    if (c->P.no_cor_mat) return fw_fail(c, FW_ERR_STATE, "fw_set_cor_mat: the context was created with no_cor_mat (dense_cor = false)");
    if (!cor) return fw_fail(c, FW_ERR_ARG, "fw_set_cor_mat: NULL matrix");
    const size_t cells = (size_t)c->P.p * c->P.p, bytes = sizeof(float) * cells;
    // the device arithmetic relies on |entries| <= 1 (what cor() produces, cov2cor clamps); NaN is allowed and propagates
    for (size_t t = 0; t < cells; ++t)
        if (std::fabs(cor[t]) > 1.0f)
            return fw_fail(c, FW_ERR_ARG, "fw_set_cor_mat: entry %zu = %g is outside [-1, 1]: not a correlation matrix", t,
                           (double)cor[t]);
    if (!c->d_cor) FW_HIP(c, hipMalloc(&c->d_cor, bytes));
    FW_HIP(c, hipMemcpy(c->d_cor, cor, bytes, hipMemcpyHostToDevice));
    c->have_cor = true;
    c->have_level0 = false;
    c->have_network = false;
    return FW_OK;
}

int fw_compute_cor_mat(fw_ctx *c)
{
    CHECK_CTX(c);
    if (c->P.kind != FW_FZ) return fw_fail(c, FW_ERR_ARG, "fw_compute_cor_mat: context is not FW_FZ");
    if (c->P.no_cor_mat) return fw_fail(c, FW_ERR_STATE, "fw_compute_cor_mat: the context was created with no_cor_mat (dense_cor = false)");
    return fwi_fz_compute_cor(c);
}

int fw_get_cor_mat(const fw_ctx *c, float *out)
{
    CHECK_CTX(c);
    if (!c->have_cor) return fw_fail(c, FW_ERR_STATE, "fw_get_cor_mat: no correlation matrix resident");
    if (!out) return fw_fail(c, FW_ERR_ARG, "fw_get_cor_mat: NULL output");
    FW_HIP(c, hipMemcpy(out, c->d_cor, sizeof(float) * (size_t)c->P.p * c->P.p, hipMemcpyDeviceToHost));
    return FW_OK;
}

// learning.jl:51-64 automatic n_obs_min for discrete tests (needs maximum(levels))
static void resolve_n_obs_min_discrete(fw_ctx *c)
{
    if (c->P.n_obs_min >= 0) {
        c->n_obs_min_eff = c->P.n_obs_min;
        return;
    }
    int64_t max_level = 0;
    for (int32_t l : c->levels) max_level = std::max<int64_t>(max_level, l);
    int64_t n_strata = 1;
    for (int j = 0; j < c->P.max_k; ++j) {
        n_strata *= max_level;
        if (n_strata > 8) break;
    }
    n_strata = std::min<int64_t>(n_strata, 8);
    c->n_obs_min_eff = (int64_t)c->P.hps * 2 * 2 * n_strata;
}

int fw_set_data_csc_i32(fw_ctx *c, const int64_t *colptr, const int32_t *rowval, const int32_t *nzval)
{
    CHECK_CTX(c);
    if (c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) return fw_fail(c, FW_ERR_ARG, "fw_set_data_csc_i32: context is not discrete");
    if (!colptr || (colptr[c->P.p] > 0 && (!rowval || !nzval))) return fw_fail(c, FW_ERR_ARG, "fw_set_data_csc_i32: NULL array");
    int rc = fwi_mi_upload(c, colptr, rowval, nzval);
    if (rc) return rc;
    resolve_n_obs_min_discrete(c);
    c->have_data = true;
    c->have_level0 = false;
    c->have_network = false;
    return FW_OK;
}

int fw_set_data_dense_i32(fw_ctx *c, const int32_t *data)
{
    CHECK_CTX(c);
    if (c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) return fw_fail(c, FW_ERR_ARG, "fw_set_data_dense_i32: context is not discrete");
    if (!data) return fw_fail(c, FW_ERR_ARG, "fw_set_data_dense_i32: NULL data");
    const int n = c->P.n, p = c->P.p;
    std::vector<int64_t> colptr(p + 1, 0);
    std::vector<int32_t> rowval, nzval;
    for (int v = 0; v < p; ++v) {
        for (int i = 0; i < n; ++i) {
            int32_t x = data[(size_t)v * n + i];
            if (x != 0) {
                rowval.push_back(i);
                nzval.push_back(x);
            }
        }
        colptr[v + 1] = (int64_t)rowval.size();
    }
    return fw_set_data_csc_i32(c, colptr.data(), rowval.data(), nzval.data());
}

int fw_get_levels(const fw_ctx *c, int32_t *levels, int32_t *max_vals)
{
    CHECK_CTX(c);
    if (c->levels.empty()) return fw_fail(c, FW_ERR_STATE, "fw_get_levels: no discrete data uploaded");
    if (levels) memcpy(levels, c->levels.data(), sizeof(int32_t) * c->levels.size());
    if (max_vals) memcpy(max_vals, c->max_vals.data(), sizeof(int32_t) * c->max_vals.size());
    return FW_OK;
}

int64_t fw_effective_n_obs_min(const fw_ctx *c) { return c ? c->n_obs_min_eff : -1; }

int fw_set_row_views(fw_ctx *c, int32_t on)
{
    CHECK_CTX(c);
    c->mi_view = on ? 1 : 0;
    return FW_OK;
}

// ---- level 0 --------------------------------------------------------------------------------------
// BH (statfuns.jl:326-350) on the p < alpha subset + neighbour lists (tests.jl:372-388).
static int fw_level0_impl(fw_ctx *c, int64_t *nnz_out, int rank, int world, fw_allgather_fn allgather, void *user,
                          const fw_dev_exchange *xdev = nullptr);

int fw_level0(fw_ctx *c, int64_t *nnz_out) { return fw_level0_impl(c, nnz_out, 0, 1, nullptr, nullptr); }

int fw_level0_sharded_dev(fw_ctx *c, int32_t rank, int32_t world_size, const fw_dev_exchange *x, int64_t *nnz_out)
{
    CHECK_CTX(c);
    if (world_size < 1 || rank < 0 || rank >= world_size) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded_dev: rank %d outside world of %d", rank, world_size);
    if (world_size > 1 && (!x || !x->prepare || !x->exchange)) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded_dev: world_size > 1 needs both exchange callbacks");
    return fw_level0_impl(c, nnz_out, rank, world_size, nullptr, nullptr, x);
}

int fw_use_cor_buffer(fw_ctx *c, void *d_cor, int64_t capacity_floats)
{
    CHECK_CTX(c);
    if (c->P.kind != FW_FZ) return fw_fail(c, FW_ERR_ARG, "fw_use_cor_buffer: context is not FW_FZ");
    if (c->P.no_cor_mat) return fw_fail(c, FW_ERR_STATE, "fw_use_cor_buffer: the context was created with no_cor_mat (dense_cor = false)");
    if (!d_cor || capacity_floats < (int64_t)c->P.p * c->P.p) return fw_fail(c, FW_ERR_ARG, "fw_use_cor_buffer: needs at least p * p floats of device memory");
    if (c->d_cor && !c->cor_external) (void)hipFree(c->d_cor);
    c->d_cor = (float *)d_cor;
    c->cor_external = true;
    c->cor_capacity = capacity_floats;
    c->have_cor = false;
    c->have_level0 = false;
    c->have_network = false;
    return FW_OK;
}

int fw_compute_cor_mat_rows(fw_ctx *c, int32_t rank, int32_t world_size, int64_t *row0, int64_t *rows_per_rank)
{
    CHECK_CTX(c);
    if (c->P.kind != FW_FZ) return fw_fail(c, FW_ERR_ARG, "fw_compute_cor_mat_rows: context is not FW_FZ");
    if (c->P.no_cor_mat) return fw_fail(c, FW_ERR_STATE, "fw_compute_cor_mat_rows: the context was created with no_cor_mat (dense_cor = false)");
    if (world_size < 1 || rank < 0 || rank >= world_size || !row0 || !rows_per_rank) return fw_fail(c, FW_ERR_ARG, "fw_compute_cor_mat_rows: invalid argument");
    c->have_level0 = false;
    c->have_network = false;
    return fwi_fz_compute_cor_rows(c, rank, world_size, row0, rows_per_rank);
}

int fw_cor_mat_ready(fw_ctx *c)
{
    CHECK_CTX(c);
    if (c->P.kind != FW_FZ || !c->d_cor) return fw_fail(c, FW_ERR_STATE, "fw_cor_mat_ready: no correlation matrix buffer");
    c->have_cor = true;
    return FW_OK;
}

int fw_level0_sharded(fw_ctx *c, int32_t rank, int32_t world_size, fw_allgather_fn allgather, void *user, int64_t *nnz_out)
{
    CHECK_CTX(c);
    if (world_size < 1 || rank < 0 || rank >= world_size) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded: rank %d outside world of %d", rank, world_size);
    if (world_size > 1 && !allgather) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded: world_size > 1 needs an allgather callback");
    return fw_level0_impl(c, nnz_out, rank, world_size, allgather, user);
}

static int fw_level0_impl(fw_ctx *c, int64_t *nnz_out, int rank, int world, fw_allgather_fn allgather, void *user,
                          const fw_dev_exchange *xdev)
{
    CHECK_CTX(c);
    const double t0 = now_s();
    const int p = c->P.p;
    if (c->P.kind == FW_FZ && !c->P.no_cor_mat) {
        if (!c->have_cor) {
            int rc = fwi_fz_compute_cor(c);
            if (rc) return rc;
        }
    } else if (!c->have_data) {  // (no_cor_mat: the level-0 kernels multiply and screen the centred columns themselves)
        return fw_fail(c, FW_ERR_STATE, "fw_level0: no data uploaded");
    }
    if (c->n_obs_min_eff > c->P.n)  // learning.jl:66-73
        return fw_fail(c, FW_ERR_NOBS,
                       "Dataset has an insufficient number of observations, need at least %lld ('n_obs_min') for reliable tests",
                       (long long)c->n_obs_min_eff);
    std::vector<int32_t> pi, pj;
    std::vector<double> stat, pval;
    int64_t m = 0;
    // BH + neighbour lists run on the device (fw_bh.hip); FW_HOST_BH=1 keeps the host restatement below, which
    // tests/test_gpu_*.py use to cross-check the two
    const char *hb_env = fw_knob("FW_HOST_BH");
    const bool host_bh = hb_env && atoi(hb_env) == 1;
    FwL0Dev dev;
    FwL0Dev *devp = host_bh ? nullptr : &dev;
    // Target-sharded runs: for the discrete kinds, whose pair screen is the expensive part of level 0
    // (cfg4: 73 of 140 ms), every rank screens its share of the pair tiles and the significant pairs are all-gathered; the
    // Fisher-z kinds screen a resident matrix in well under a millisecond per 10^8 pairs and stay replicated.  BH and the
    // neighbour lists are then built redundantly on every rank from the same merged list (their result does not depend
    // on the order of the list).
    const bool sharded = world > 1 && (c->P.kind == FW_MI || c->P.kind == FW_MI_NZ);
    c->l0_rank = sharded ? rank : 0;
    c->l0_world = sharded ? world : 1;
    int rc = (c->P.kind == FW_FZ)      ? fwi_fz_level0(c, pi, pj, stat, pval, &m, devp)
             : (c->P.kind == FW_FZ_NZ) ? fwi_fznz_level0(c, pi, pj, stat, pval, &m, devp)
                                       : fwi_mi_level0(c, pi, pj, stat, pval, &m, (sharded && !xdev) ? nullptr : devp);
    c->l0_rank = 0;
    c->l0_world = 1;
    if (rc) return rc;
    if (sharded && xdev) {
        // payload stays on the device: pack -> caller's collective -> unpack (fw_xchg.hip).  m = sum_r m_r - (W - 1) * npairs
        // (every rank reports npairs minus ITS unreliable pairs)
        if (host_bh) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded_dev: FW_HOST_BH=1 is a single-rank debugging mode");
        FwL0Dev merged;
        int64_t msum = 0;
        if ((rc = fwi_l0_exchange_dev(c, xdev, world, dev, m, &merged, &msum))) return rc;
        dev = merged;
        m = msum - (int64_t)(world - 1) * ((int64_t)p * (p - 1) / 2);
    } else if (sharded) {
        // one extra record carries this rank's count of reliable tests: m = sum_r m_r - (W - 1) * npairs (every rank reports
        // npairs minus ITS unreliable pairs)
        pi.push_back(-1);
        pj.push_back(rank);
        stat.push_back((double)m);
        pval.push_back(0.0);
        int64_t ntot = 0;
        const int32_t *ai = nullptr, *aj = nullptr;
        const double *as = nullptr, *ap = nullptr;
        rc = allgather(user, (int64_t)pi.size(), pi.data(), pj.data(), stat.data(), pval.data(), &ntot, &ai, &aj, &as, &ap);
        if (rc) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded: allgather callback failed (%d)", rc);
        std::vector<int32_t> qi, qj;
        std::vector<double> qs, qp;
        qi.reserve((size_t)ntot);
        qj.reserve((size_t)ntot);
        qs.reserve((size_t)ntot);
        qp.reserve((size_t)ntot);
        const int64_t npairs = (int64_t)p * (p - 1) / 2;
        int64_t msum = 0;
        int nrec = 0;
        for (int64_t t = 0; t < ntot; ++t) {
            if (ai[t] < 0) {
                msum += (int64_t)as[t];
                ++nrec;
                continue;
            }
            qi.push_back(ai[t]);
            qj.push_back(aj[t]);
            qs.push_back(as[t]);
            qp.push_back(ap[t]);
        }
        if (nrec != world) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded: %d of %d ranks reported", nrec, world);
        m = msum - (int64_t)(world - 1) * npairs;
        pi.swap(qi);
        pj.swap(qj);
        stat.swap(qs);
        pval.swap(qp);
        if (!host_bh) {  // back to the device for the BH / neighbour-list epilogue
            const size_t k = pi.size();
            if ((rc = fw_dev_reserve(c, c->d_tmp1, (k + 1) * 2 * sizeof(int32_t)))) return rc;
            if ((rc = fw_dev_reserve(c, c->d_tmp2, (k + 1) * 2 * sizeof(double)))) return rc;
            int32_t *oi = (int32_t *)c->d_tmp1.ptr, *oj = oi + k;
            double *os = (double *)c->d_tmp2.ptr, *op = os + k;
            if (k) {
                FW_HIP(c, hipMemcpyAsync(oi, pi.data(), k * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                FW_HIP(c, hipMemcpyAsync(oj, pj.data(), k * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                FW_HIP(c, hipMemcpyAsync(os, stat.data(), k * sizeof(double), hipMemcpyHostToDevice, c->stream));
                FW_HIP(c, hipMemcpyAsync(op, pval.data(), k * sizeof(double), hipMemcpyHostToDevice, c->stream));
                FW_HIP(c, hipStreamSynchronize(c->stream));
            }
            dev = FwL0Dev{};
            dev.i = oi;
            dev.j = oj;
            dev.stat64 = os;
            dev.pval = op;
            dev.k = k;
        }
    }
    if (!host_bh) {
        const double t_bh0 = now_s();
        if ((rc = fwi_bh_csr_device(c, dev, m))) return rc;
        c->cnt.t_level0_host_s += now_s() - t_bh0;  // here: the device epilogue incl. its device-to-host copies
        c->have_level0 = true;
        c->have_network = false;
        c->cnt.level0_tests += (int64_t)p * (p - 1) / 2;
        c->cnt.t_level0_s += now_s() - t0;
        if (nnz_out) *nnz_out = c->nb_off[p];
        return FW_OK;
    }
    c->d_nb_off = nullptr;
    c->d_nb_idx = nullptr;
    c->d_nb_stat = c->d_nb_p = nullptr;
    c->d_cand = nullptr;
    c->nb_host_valid = true;
    const size_t k = pi.size();
    const double t_host0 = now_s();
    if (c->P.fdr && k > 0) {
        // statfuns.jl:326-350.  Ascending sort by p: LSD radix sort on the IEEE bit pattern (p >= 0 -> order preserving).
        // Ties need no stable order: the backward cumulative minimum gives every member of a tie group the same value.
        std::vector<uint64_t> key(k), key2(k);
        std::vector<uint32_t> val(k), val2(k);
        for (size_t t = 0; t < k; ++t) {
            uint64_t u;
            memcpy(&u, &pval[t], 8);
            key[t] = u;
            val[t] = (uint32_t)t;
        }
        std::vector<uint32_t> hist(65536);
        for (int pass = 0; pass < 4; ++pass) {
            const int sh = 16 * pass;
            std::fill(hist.begin(), hist.end(), 0u);
            for (size_t t = 0; t < k; ++t) hist[(key[t] >> sh) & 0xFFFF]++;
            uint32_t run = 0;
            for (size_t d = 0; d < 65536; ++d) {
                const uint32_t h = hist[d];
                hist[d] = run;
                run += h;
            }
            for (size_t t = 0; t < k; ++t) {
                const uint32_t dst = hist[(key[t] >> sh) & 0xFFFF]++;
                key2[dst] = key[t];
                val2[dst] = val[t];
            }
            key.swap(key2);
            val.swap(val2);
        }
        const double md = (double)m;
        double next_adj = 0.0;
        for (size_t i = k; i-- > 0;) {
            double pv;
            memcpy(&pv, &key[i], 8);
            double adj = pv * md / (double)(i + 1);
            if (i == k - 1)
                adj = std::min(adj, 1.0);
            else
                adj = std::min(next_adj, adj);
            next_adj = adj;
            pval[val[i]] = adj;
        }
    }
    // neighbour lists (tests.jl:372-388): adj p < alpha, partners ascending.  Two stable counting sorts give the
    // (X ascending, Y ascending) pair order of the reference's condensed arrays; appending in that order yields
    // ascending partner lists for both endpoints.
    std::vector<uint32_t> keep;
    keep.reserve(k);
    for (size_t t = 0; t < k; ++t)
        if (pval[t] < c->P.alpha) keep.push_back((uint32_t)t);
    const size_t kk = keep.size();
    std::vector<uint32_t> byj(kk), byij(kk);
    {
        std::vector<uint32_t> cnt((size_t)p + 1, 0);
        for (uint32_t t : keep) cnt[pj[t] + 1]++;
        for (int v = 0; v < p; ++v) cnt[v + 1] += cnt[v];
        for (uint32_t t : keep) byj[cnt[pj[t]]++] = t;
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (uint32_t t : byj) cnt[pi[t] + 1]++;
        for (int v = 0; v < p; ++v) cnt[v + 1] += cnt[v];
        for (uint32_t t : byj) byij[cnt[pi[t]]++] = t;
    }
    c->nb_off.assign((size_t)p + 1, 0);
    for (uint32_t t : byij) {
        c->nb_off[pi[t] + 1]++;
        c->nb_off[pj[t] + 1]++;
    }
    for (int v = 0; v < p; ++v) c->nb_off[v + 1] += c->nb_off[v];
    const int64_t tot = c->nb_off[p];
    c->nb_idx.assign((size_t)tot, 0);
    c->nb_stat.assign((size_t)tot, 0.0);
    c->nb_p.assign((size_t)tot, 0.0);
    std::vector<int64_t> fill(c->nb_off.begin(), c->nb_off.end() - 1);
    for (uint32_t t : byij) {
        const int X = pi[t], Y = pj[t];
        const int64_t a = fill[X]++, b = fill[Y]++;
        c->nb_idx[a] = Y;
        c->nb_stat[a] = stat[t];
        c->nb_p[a] = pval[t];
        c->nb_idx[b] = X;
        c->nb_stat[b] = stat[t];
        c->nb_p[b] = pval[t];
    }
    c->cnt.t_level0_host_s += now_s() - t_host0;
    c->have_level0 = true;
    c->have_network = false;
    c->cnt.level0_tests += (int64_t)p * (p - 1) / 2;
    c->cnt.t_level0_s += now_s() - t0;
    if (nnz_out) *nnz_out = tot;
    return FW_OK;
}

int fw_level0_get(const fw_ctx *c, int64_t *off, int32_t *idx, double *stat, double *adj_p)
{
    CHECK_CTX(c);
    if (!c->have_level0) return fw_fail(c, FW_ERR_STATE, "fw_level0_get: fw_level0 has not run");
    if (int rc = fwi_nb_host_ensure(const_cast<fw_ctx *>(c))) return rc;  // lists may still live on the device only
    if (off) memcpy(off, c->nb_off.data(), sizeof(int64_t) * c->nb_off.size());
    const size_t tot = c->nb_idx.size();
    if (idx && tot) memcpy(idx, c->nb_idx.data(), sizeof(int32_t) * tot);
    if (stat && tot) memcpy(stat, c->nb_stat.data(), sizeof(double) * tot);
    if (adj_p && tot) memcpy(adj_p, c->nb_p.data(), sizeof(double) * tot);
    return FW_OK;
}

// ---- per-pair batches --------------------------------------------------------------------------------
static int check_var(const fw_ctx *c, int32_t v) { return v >= 0 && v < c->P.p; }

int fw_test_batch(fw_ctx *c, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff, const int32_t *zflat,
                  fw_test_result *out)
{
    CHECK_CTX(c);
    if (m < 0 || (m > 0 && (!X || !Y || !zoff || !out))) return fw_fail(c, FW_ERR_ARG, "fw_test_batch: NULL argument");
    if (m == 0) return FW_OK;
    for (int64_t t = 0; t < m; ++t) {
        const int64_t k = zoff[t + 1] - zoff[t];
        if (k < 0 || k > FW_MAX_K) return fw_fail(c, FW_ERR_LIMIT, "fw_test_batch: test %lld has %lld conditioning variables (max %d)", (long long)t, (long long)k, FW_MAX_K);
        if (!check_var(c, X[t]) || !check_var(c, Y[t])) return fw_fail(c, FW_ERR_ARG, "fw_test_batch: variable index out of range in test %lld", (long long)t);
        for (int64_t q = zoff[t]; q < zoff[t + 1]; ++q)
            if (!check_var(c, zflat[q])) return fw_fail(c, FW_ERR_ARG, "fw_test_batch: conditioning variable out of range in test %lld", (long long)t);
    }
    if (c->P.kind == FW_FZ && !c->P.recursive_pcor) {
        // no cor_mat for conditional tests (tests.jl:253 -> pcor): they stream their sample columns; univariate tests keep
        // the matrix path (level 0 computes the matrix anyway)
        if (!c->have_data) return fw_fail(c, FW_ERR_STATE, "fw_test_batch: no data uploaded");
        if ((int64_t)c->P.n < c->n_obs_min_eff) {
            for (int64_t t = 0; t < m; ++t) out[t] = fw_test_result{0.0, 1.0, 0, 0};
            return FW_OK;
        }
        std::vector<int64_t> iu, ic;
        // (no_cor_mat: univariate tests come from the data too -- Float64 sums of the two columns, not the Float32 MFMA value level 0 screens)
        for (int64_t t = 0; t < m; ++t) ((zoff[t + 1] == zoff[t] && !c->P.no_cor_mat) ? iu : ic).push_back(t);
        for (int part = 0; part < 2; ++part) {
            const std::vector<int64_t> &ix = part ? ic : iu;
            if (ix.empty()) continue;
            if (part == 0 && !c->have_cor)
                if (int rc = fwi_fz_compute_cor(c)) return rc;
            std::vector<int32_t> x2(ix.size()), y2(ix.size()), zf;
            std::vector<int64_t> zo(ix.size() + 1, 0);
            for (size_t q = 0; q < ix.size(); ++q) {
                x2[q] = X[ix[q]];
                y2[q] = Y[ix[q]];
                zf.insert(zf.end(), zflat + zoff[ix[q]], zflat + zoff[ix[q] + 1]);
                zo[q + 1] = (int64_t)zf.size();
            }
            if (zf.empty()) zf.push_back(0);
            std::vector<fw_test_result> o2(ix.size());
            const int rc = part ? fwi_fzs_test_batch(c, (int64_t)ix.size(), x2.data(), y2.data(), zo.data(), zf.data(), o2.data())
                                : fwi_fz_test_batch(c, (int64_t)ix.size(), x2.data(), y2.data(), zo.data(), zf.data(), o2.data());
            if (rc) return rc;
            for (size_t q = 0; q < ix.size(); ++q) out[ix[q]] = o2[q];
        }
        return FW_OK;
    }
    if (c->P.kind == FW_FZ) {
        if (!c->have_cor) return fw_fail(c, FW_ERR_STATE, "fw_test_batch: no correlation matrix (fw_compute_cor_mat / fw_set_cor_mat)");
        return fwi_fz_test_batch(c, m, X, Y, zoff, zflat, out);
    }
    if (!c->have_data) return fw_fail(c, FW_ERR_STATE, "fw_test_batch: no data uploaded");
    if (c->P.kind == FW_FZ_NZ) {
        if ((int64_t)c->P.n < c->n_obs_min_eff) {  // tests.jl:11 / :254 on the full row count
            for (int64_t t = 0; t < m; ++t) out[t] = fw_test_result{0.0, 1.0, 0, 0};
            return FW_OK;
        }
        return fwi_fznz_test_batch(c, m, X, Y, zoff, zflat, out);
    }
    return fwi_mi_test_batch(c, m, X, Y, zoff, zflat, out);
}

}  // extern "C"

// ---- progressive, segment-parallel evaluation of test_subsets jobs -------------------------------------
// A job's subsets are ranked in the reference's enumeration order.  Ranks are evaluated in windows that grow
// geometrically (256, 1k, 4k, ... ranks); each window is cut into segments, one workgroup per segment, and the
// per-segment outputs are merged IN RANK ORDER on the host: the first stopping rank ends the job exactly where the
// sequential reference would have stopped (tests.jl:326-336); otherwise the (p, rank) maximum with "later wins
// ties" is carried forward (tests.jl:338-341).  Speculation is bounded by the window growth factor.
static uint64_t binom_sat(int64_t m, int t)
{
    if (t < 0 || m < t) return 0;
    const uint64_t SAT = 1ull << 62;
    long double r = 1.0L;
    for (int i = 1; i <= t; ++i) r = r * (long double)(m - t + i) / (long double)i;
    if (r > (t > 5 ? 2.0e18L : 3.6e18L)) return SAT;  // the intermediate v * (m - t + i) is up to t * C(m, t): stay below 2^64 / 5 (t = 6, 7: the device's fz_binom_sat7 bound)
    uint64_t v = 1;
    for (int i = 1; i <= t; ++i) v = v * (uint64_t)(m - t + i) / (uint64_t)i;  // exact: product of i consecutive ints / i!
    return v;
}

// lexicographic unranking (same order as the device code): rank -> subset size and positions
static void unrank_host(uint64_t r, int a, int max_k, int *s_out, int *pos)
{
    int s = max_k;
    while (s > 1) {
        const uint64_t cnt = binom_sat(a, s);
        if (r < cnt) break;
        r -= cnt;
        --s;
    }
    *s_out = s;
    int prev = -1;
    for (int d = 0; d < s; ++d) {
        const int t = s - d;
        int c = prev + 1;
        for (;;) {
            const uint64_t with_c = binom_sat(a - 1 - c, t - 1);
            if (r < with_c) break;
            r -= with_c;
            ++c;
        }
        pos[d] = c;
        prev = c;
    }
}

// geometric growth of the evaluation window of a job (speculation bound vs number of latency-bound rounds);
// FW_WINDOW_GROWTH is a tuning knob for profiling runs
static uint64_t fw_window_growth(size_t n_live, uint64_t launched_ranks)
{
    static const long forced = [] {
        const char *e = fw_knob("FW_WINDOW_GROWTH");
        long v = e ? atol(e) : 0;
        return (v >= 2 && v <= 64) ? v : 0l;
    }();
    if (forced) return (uint64_t)forced;
    // many jobs in flight: launches are full, keep speculation tight (x4); few jobs: the round latency dominates
    // and a wider window (x16) saves rounds (measured on cfg3: 3236 -> 2024 launches for +9 % evaluated tests);
    // a launch that does not even fill the GPU (a few thousand workgroups of 256 ranks) costs the same whether its
    // windows are 16 or 256 times larger: grow faster there, the extra speculative tests are free
    static const uint64_t small = [] {
        const char *e = fw_knob("FW_SMALL_LAUNCH");
        return e ? (uint64_t)atoll(e) : (uint64_t)(1u << 22);  // cfg3 sweep: 0 -> 630 ms, 1M 613, 4M 569, 8M 588, 64M 582
    }();
    if (launched_ranks < small) return 256;
    return n_live > 2048 ? 4 : 16;
}

static void no_power_result(const fw_ctx *c, const int32_t *acc, int a, FwJobOut &o)
{
    // tests.jl:254-262: every test lacks power -> the first one is returned (0, 1, 0, false)
    o = FwJobOut{};
    o.stat = 0.0;
    o.pval = 1.0;
    o.num_tests = 1;
    o.evaluated = 0;
    o.status = FW_SUBSETS_STOPPED;
    const int s = std::min<int>(c->P.max_k, a);
    o.n_zs = s;
    for (int q = 0; q < s; ++q) o.zs[q] = acc[q];
}

int fwi_pool_add(fw_ctx *c, FwPool &pool, int32_t X, int32_t Y, const int32_t *acc, int a, int64_t tag)
{
    FwPoolJob j;
    j.X = X;
    j.Y = Y;
    j.tag = tag;
    j.acc.assign(acc, acc + a);
    uint64_t N = 0;
    for (int s = c->P.max_k; s >= 1; --s) {
        N += binom_sat(a, s);
        if (N > (1ull << 62)) N = 1ull << 62;
    }
    const uint64_t mt = c->P.max_tests > 0 ? (uint64_t)c->P.max_tests : 0;
    if (mt && mt < N) N = mt;
    j.N = N;
    j.next = 0;
    // first window: fz 256 ranks (one test per lane of one workgroup), 16384 once |accepted| >= 64 -- a job that large
    // either stops within the first few tests or runs for tens of thousands, so small first windows are wasted round
    // trips (cfg3: 1731 -> 1154 launches per pass, +4 % evaluated tests, 0.527 -> 0.50 s); discrete: 16
    static const uint64_t w0_big = [] { const char *e = fw_knob("FW_W0_BIG"); return e ? (uint64_t)atoll(e) : (uint64_t)16384; }();
    j.width = (c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) ? (a >= 64 ? w0_big : 256ull) : 16ull;
    j.best_p = -1.0;
    j.best_stat = 0.0;
    j.best_rank = 0;
    j.best_df = 0;
    j.done = false;
    j.out = FwJobOut{};
    pool.live.push_back(std::move(j));
    return FW_OK;
}

static void finish_job(const fw_ctx *c, FwPoolJob &j, bool want_zs)
{
    j.done = true;
    j.out.n_zs = 0;
    if (!want_zs || j.no_zs) return;  // the HITON driver never looks at the conditioning set of the returned result
    int s = 0, pos[FW_MAX_K] = {0};
    unrank_host(j.best_rank, (int)j.acc.size(), c->P.max_k, &s, pos);  // conditioning set of the returned result
    j.out.n_zs = s;
    for (int q = 0; q < FW_MAX_K; ++q) j.out.zs[q] = q < s ? j.acc[pos[q]] : 0;
    j.done = true;
}

// One window of every live job = one kernel launch on the pool's own stream.  fwi_pool_launch only enqueues;
// fwi_pool_collect waits for it and merges.  Two pools can therefore overlap: the host merges / re-posts one
// while the GPU evaluates the other.
int fwi_pool_launch(fw_ctx *c, FwPool &pool)
{
    pool.ns = 0;
    pool.inflight = false;
    if (pool.owner_epoch) {  // drop jobs whose owner's accepted set has changed since they were posted
        size_t w = 0;
        for (size_t ji = 0; ji < pool.live.size(); ++ji) {
            FwPoolJob &j = pool.live[ji];
            if (j.epoch != (*pool.owner_epoch)[(size_t)j.tag]) {
                pool.dropped_evaluated += j.out.evaluated;
                pool.dropped_alg_bytes += fwi_alg_bytes(c, (int)j.acc.size(), j.out.evaluated);
                continue;
            }
            if (w != ji) pool.live[w] = std::move(j);
            ++w;
        }
        pool.live.resize(w);
    }
    if (pool.live.empty()) return FW_OK;
    const bool stream = c->P.kind == FW_FZ && !c->P.recursive_pcor;  // one wavefront per test on the sample columns (fw_fzs.hip)
    const bool fz = (c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) && !stream;  // one lane per test on a correlation matrix
    const bool nzs = c->P.kind == FW_FZ_NZ;
    if (c->P.kind == FW_FZ && c->P.n < c->n_obs_min_eff) {  // no device work: fwi_pool_collect fills the results
        pool.inflight = true;
        return FW_OK;
    }
    if (stream) {  // job matrices kept across the rounds of this pool (ctx->d_arena): a new pool, or an arena another pool used, starts empty
        if (pool.gram_epoch == 0 || pool.gram_epoch != c->gram_epoch || pool.gram_top * sizeof(double) > ((size_t)4 << 30)) {
            pool.gram_epoch = ++c->gram_epoch;
            pool.gram_top = 0;
        }
    }
    const double tb0 = now_s();
    FwPoolBuf &pb = c->pb[pool.buf];
    // window of every live job, then a segment length that yields a few thousand workgroups
    uint64_t total = 0, acc_total = 0;
    size_t n_launch = 0;
    for (FwPoolJob &j : pool.live) {
        j.launched = !j.hold;
        if (!j.launched) continue;
        ++n_launch;
        total += std::min(j.width, j.N - j.next);
        acc_total += j.acc.size();
    }
    pool.launched_ranks = total;
    if (n_launch == 0) return FW_OK;  // everything is on hold: nothing to do this round
    // fz: one lane per test (256-rank granularity); discrete: one wavefront per test (4-rank granularity)
    const uint64_t q = fz ? 256 : 4, smin = fz ? 256 : 8, smax = fz ? 8192 : 256;
    static const uint64_t seg_target = [] {
        const char *e = fw_knob("FW_SEG_TARGET");  // workgroups per launch the segment length aims for
        return e && atoll(e) > 0 ? (uint64_t)atoll(e) : (uint64_t)0;
    }();
    const uint64_t seg_tgt = seg_target ? seg_target : (fz ? 3072 : 4096);  // fz: runs of up to 32 ranks per lane
    uint64_t seglen = (total / seg_tgt + q - 1) / q * q;
    seglen = std::max<uint64_t>(smin, std::min<uint64_t>(seglen, smax));
    size_t ns = 0;
    for (const FwPoolJob &j : pool.live)
        if (j.launched) ns += (size_t)((std::min(j.width, j.N - j.next) + seglen - 1) / seglen);
    // Accepted lists live in a device arena for as long as their job does: a job uploads its list once, with its
    // first window, not with every window (at cfg3 the lists of ~1500 live jobs are ~1 MB per round).  The arena is a
    // bump allocator; when it is full every live job is re-staged from offset 0.
    int rc;
    size_t new_ints = 0;
    for (const FwPoolJob &j : pool.live)
        if (j.launched && j.acc_dev_off < 0) new_ints += j.acc.size();
    if (pool.arena_top + new_ints > pb.d_acc.cap / sizeof(int32_t) || pool.arena_top == 0) {
        size_t live_ints = 0;
        for (FwPoolJob &j : pool.live) {
            j.acc_dev_off = -1;
            live_ints += j.acc.size();
        }
        if ((rc = fw_dev_reserve(c, pb.d_acc, std::max<size_t>(4 * live_ints, (size_t)1 << 20) * sizeof(int32_t)))) return rc;
        pool.arena_top = 0;
        new_ints = acc_total;
    }
    const size_t in_bytes = ns * sizeof(FwSeg) + std::max<size_t>(new_ints, 1) * sizeof(int32_t);
    if ((rc = fw_pin_reserve(c, pb.h_in, in_bytes))) return rc;
    if ((rc = fw_pin_reserve(c, pb.h_out, ns * sizeof(FwSegOut)))) return rc;
    if ((rc = fw_dev_reserve(c, pb.d_in, ns * sizeof(FwSeg)))) return rc;
    // results: one 64-byte record per workgroup, written straight into pinned host memory (posted PCIe writes) -- saves
    // the device-to-host copy of every round; FW_ZC_OUT=0 stages them through device memory instead (profiling knob)
    static const bool zc_out = !(fw_knob("FW_ZC_OUT") && atoi(fw_knob("FW_ZC_OUT")) == 0);
    if (!zc_out && (rc = fw_dev_reserve(c, pb.d_out, ns * sizeof(FwSegOut)))) return rc;
    FwSegOut *dout = zc_out ? (FwSegOut *)pb.h_out.ptr : (FwSegOut *)pb.d_out.ptr;
    FwSeg *segs = (FwSeg *)pb.h_in.ptr;
    int32_t *hacc = (int32_t *)((char *)pb.h_in.ptr + ns * sizeof(FwSeg));
    pool.seg_job.resize(ns);
    pool.nzrecs.clear();
    size_t arena_floats = 0;
    size_t si = 0, ns_tab = 0;
    size_t staged = 0;
    // fz / fz_nz with max_k <= 3: segments of jobs with at most FW_TAB_A accepted variables come first (table kernel)
    static const bool no_tab = fw_knob("FW_NO_TAB") != nullptr;  // profiling knob: force the in-lane caching kernel
    const bool no_hk = fw_knob("FW_NO_HK") != nullptr;  // profiling / test knob: generic size-4/5 kernel for every job
    const bool hk = c->P.max_k > 3;  // max_k 4-5: level-2 table kernel up to FW_HK_A accepted variables (plain fz only)
    const bool split = fz && !no_tab && (!hk || (c->P.kind == FW_FZ && !stream && !no_hk));
    const size_t tab_a = hk ? (size_t)FW_HK_A : (size_t)FW_TAB_A;
    // record of every launched job in pool.nzrecs: with `split` the records are pushed in two passes (short lists first), i.e. NOT in
    // pool.live order -- the re-layout below must follow this map, not a running counter
    std::vector<size_t> job_rec(pool.live.size(), (size_t)-1);
    for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
        for (size_t ji = 0; ji < pool.live.size(); ++ji) {
            FwPoolJob &j = pool.live[ji];
            if (!j.launched) continue;
            if (split && (j.acc.size() <= tab_a) != (pass == 0)) continue;
            if (j.acc_dev_off < 0) {
                memcpy(hacc + staged, j.acc.data(), j.acc.size() * sizeof(int32_t));
                j.acc_dev_off = (int64_t)(pool.arena_top + staged);
                staged += j.acc.size();
            }
            const int64_t aoff = j.acc_dev_off;
            const uint64_t lo = j.next, hi = lo + std::min(j.width, j.N - lo);
            for (uint64_t sgs = lo; sgs < hi; sgs += seglen) {
                FwSeg sg{};
                sg.X = j.X;
                sg.Y = j.Y;
                sg.acc_off = aoff;
                sg.acc_len = (int32_t)j.acc.size();
                sg.start = sgs;
                sg.end = std::min(hi, sgs + seglen);
                sg.pad = (int32_t)pool.nzrecs.size();  // fz_nz: index of the job's record in this launch
                segs[si] = sg;
                pool.seg_job[si] = (int64_t)ji;
                ++si;
            }
            if (nzs || stream) {  // per-job record: fz_nz sub-matrix / recursive_pcor = 0 job-local correlation matrix
                job_rec[ji] = pool.nzrecs.size();
                FwNzJob r{};
                r.X = j.X;
                r.Y = j.Y;
                r.acc_off = aoff;
                r.acc_len = (int32_t)j.acc.size();
                r.m = r.acc_len + 2;
                if (stream) {
                    // the matrix of a job stays in the arena while the job lives: a job that takes several windows (pool rounds) has it
                    // computed once (whole cfg3: fzs_gram_kernel was 30 % of the kernel time with one computation per round)
                    if (j.gram_off < 0 || j.gram_epoch != pool.gram_epoch) {
                        j.gram_off = (int64_t)pool.gram_top;
                        j.gram_epoch = pool.gram_epoch;
                        pool.gram_top += (size_t)r.m * r.m;
                        r.nR = 1;  // to be computed in this launch
                        c->cnt.gram_jobs += 1;  // (the job-matrix form's own algorithmic unit, fw_counters)
                        c->cnt.gram_alg_bytes += (double)r.m * (double)c->P.n * 4.0;
                        c->cnt.gram_alg_flops += 2.0 * (double)c->P.n * 0.5 * (double)r.m * (double)(r.m - 1);
                    }
                    r.cor_off = (long long)j.gram_off;
                    arena_floats = pool.gram_top;
                } else {
                    r.cor_off = (long long)arena_floats;
                    arena_floats += (size_t)r.m * r.m;
                }
                pool.nzrecs.push_back(r);
            }
        }
        if (split && pass == 0) ns_tab = si;
    }
    if (stream && pool.gram_top * sizeof(double) > c->d_arena.cap) {
        // the arena has to grow and loses its contents: every matrix of this launch is computed again at fresh offsets (the jobs
        // that sit this round out are caught by the epoch)
        pool.gram_epoch = ++c->gram_epoch;
        pool.gram_top = 0;
        for (size_t ji = 0; ji < pool.live.size(); ++ji) {
            FwPoolJob &j = pool.live[ji];
            if (!j.launched) continue;
            FwNzJob &r = pool.nzrecs[job_rec[ji]];
            j.gram_off = (int64_t)pool.gram_top;
            j.gram_epoch = pool.gram_epoch;
            r.cor_off = (long long)j.gram_off;
            if (r.nR == 0) {  // a cached matrix that is computed again: counted like the ones the first loop scheduled
                c->cnt.gram_jobs += 1;
                c->cnt.gram_alg_bytes += (double)r.m * (double)c->P.n * 4.0;
                c->cnt.gram_alg_flops += 2.0 * (double)c->P.n * 0.5 * (double)r.m * (double)(r.m - 1);
            }
            r.nR = 1;
            pool.gram_top += (size_t)r.m * r.m;
        }
        arena_floats = pool.gram_top;
    }
    const double tb1 = now_s();
    c->cnt.t_host_build_s += tb1 - tb0;
    FW_HIP(c, hipMemcpyAsync(pb.d_in.ptr, pb.h_in.ptr, ns * sizeof(FwSeg), hipMemcpyHostToDevice, pb.launch_stream));
    if (staged)
        FW_HIP(c, hipMemcpyAsync((int32_t *)pb.d_acc.ptr + pool.arena_top, hacc, staged * sizeof(int32_t), hipMemcpyHostToDevice,
                                 pb.launch_stream));
    pool.arena_top += staged;
    const FwSeg *dsegs = (const FwSeg *)pb.d_in.ptr;
    const int32_t *dacc = (const int32_t *)pb.d_acc.ptr;
    if (nzs) {
        const bool f64 = !c->P.recursive_pcor;  // fz_nz without a matrix: Float64 view correlations, conditioned as StatsBase.partialcor does
        rc = fwi_fznz_submatrices(c, (int64_t)pool.nzrecs.size(), pool.nzrecs.data(), arena_floats, dacc, pb.launch_stream, f64);
        if (rc) return rc;
        if (f64) {
            int m_max = 2;
            for (const FwNzJob &r : pool.nzrecs) m_max = std::max(m_max, (int)r.m);
            rc = fwi_fzs_segments_nz(c, (int64_t)ns, dsegs, dacc, dout, pb, m_max);
        } else {
            rc = fwi_fznz_segments(c, (int64_t)ns, (int64_t)ns_tab, dsegs, dacc, dout, pb);
        }
    } else {
        rc = stream ? fwi_fzs_segments(c, (int64_t)ns, dsegs, dacc, dout, pb, pool.nzrecs.data(), (int64_t)pool.nzrecs.size(), arena_floats)
             : fz   ? fwi_fz_segments(c, (int64_t)ns, (int64_t)ns_tab, dsegs, dacc, dout, pb)
                : fwi_mi_segments(c, (int64_t)ns, dsegs, dacc, dout, pb);
    }
    if (rc) return rc;
    if (stream && arena_floats * sizeof(double) > ((size_t)8 << 30))
        pool.gram_epoch = 0;  // fwi_fzs_segments streamed this launch instead (matrices beyond its arena limit): nothing was cached
    if (!zc_out)
        FW_HIP(c, hipMemcpyAsync(pb.h_out.ptr, pb.d_out.ptr, ns * sizeof(FwSegOut), hipMemcpyDeviceToHost, pb.launch_stream));
    FW_HIP(c, hipEventRecord(pb.evd, pb.launch_stream));
    pool.ns = ns;
    pool.inflight = true;
    pool.t_launch = now_s();
    c->cnt.t_host_launch_s += pool.t_launch - tb1;
    return FW_OK;
}

int fwi_pool_collect(fw_ctx *c, FwPool &pool, std::vector<FwPoolJob> &finished)
{
    if (!pool.inflight) return FW_OK;
    pool.inflight = false;
    if (c->P.kind == FW_FZ && c->P.n < c->n_obs_min_eff) {
        for (FwPoolJob &j : pool.live) {
            no_power_result(c, j.acc.data(), (int)j.acc.size(), j.out);
            j.done = true;
            finished.push_back(std::move(j));
        }
        pool.live.clear();
        return FW_OK;
    }
    FwPoolBuf &pb = c->pb[pool.buf];
    const double tb1 = now_s();
    FW_HIP(c, hipEventSynchronize(pb.evd));
    const double tb2 = now_s();
    c->cnt.t_host_wait_s += tb2 - tb1;
    float ms = 0.0f;
    FW_HIP(c, hipEventElapsedTime(&ms, pb.ev0, pb.ev1));
    c->cnt.t_dev_subsets_s += 1e-3 * (double)ms;
    {  // FW_TRACE_ROUNDS=<file>: one line per pool round (profiling aid; see profiles/README.md)
        static FILE *tf = [] {
            const char *e = fw_knob("FW_TRACE_ROUNDS");
            return e ? fopen(e, "w") : (FILE *)nullptr;
        }();
        static double t_prev = 0.0;
        if (tf) {
            unsigned long long ev_sum = 0;
            for (size_t q = 0; q < pool.ns; ++q) ev_sum += ((const FwSegOut *)pb.h_out.ptr)[q].evaluated;
            fprintf(tf, "%zu %zu %.1f %.1f %.1f %llu\n", pool.live.size(), pool.ns, 1e3 * (double)ms, 1e6 * (tb2 - tb1),
                    t_prev > 0.0 ? 1e6 * (tb2 - t_prev) : 0.0, ev_sum);
            fflush(tf);
        }
        t_prev = tb2;
    }
    c->cnt.kernel_launches += 1;
    c->cnt.subsets_launches += 1;
    const size_t ns = pool.ns;
    const FwSegOut *so = (const FwSegOut *)pb.h_out.ptr;
    // in-order merge (segments of a job are contiguous and in rank order)
    for (size_t s = 0; s < ns; ++s) {
        FwPoolJob &j = pool.live[(size_t)pool.seg_job[s]];
        j.out.evaluated += (int64_t)so[s].evaluated;
        if (j.done) continue;  // a later (speculative) segment of a job that already stopped
        if (so[s].stop_rank != FW_RANK_NONE) {
            j.out.stat = so[s].stop_stat;
            j.out.pval = so[s].stop_pval;
            j.out.df = so[s].stop_df;
            j.out.suff_power = so[s].stop_power;
            j.out.status = FW_SUBSETS_STOPPED;
            j.out.num_tests = (int64_t)(so[s].stop_rank + 1);
            j.best_rank = so[s].stop_rank;
            j.done = true;
            if (so[s].stop_df == -2) {  // fz_nz: too few rows with T != 0 and candidate != 0 -> no test at all
                j.out.df = 0;
                j.out.num_tests = 0;
                j.no_zs = true;
            }
        } else if (so[s].best_pval >= j.best_p) {
            j.best_p = so[s].best_pval;
            j.best_stat = so[s].best_stat;
            j.best_rank = so[s].best_rank;
            j.best_df = so[s].best_df;
        }
    }
    size_t w = 0;
    const uint64_t growth = fw_window_growth(pool.live.size(), pool.launched_ranks);
    for (size_t ji = 0; ji < pool.live.size(); ++ji) {
        FwPoolJob &j = pool.live[ji];
        if (!j.done && j.launched) {
            j.next += std::min(j.width, j.N - j.next);
            j.width *= growth;
            if (j.next >= j.N) {
                j.out.stat = j.best_stat;
                j.out.pval = j.best_p < 0.0 ? 0.0 : j.best_p;
                j.out.df = j.best_df;
                j.out.suff_power = 1;
                j.out.status = FW_SUBSETS_ALL_SIG;
                j.out.num_tests = (int64_t)j.N;
                j.done = true;
            }
        }
        if (j.done) {
            finish_job(c, j, pool.want_zs);
            {  // FW_TRACE_JOBS=<file>: |accepted|, evaluated tests, reference-order tests, status per finished job
                static FILE *jf = [] {
                    const char *e = fw_knob("FW_TRACE_JOBS");
                    return e ? fopen(e, "w") : (FILE *)nullptr;
                }();
                if (jf) fprintf(jf, "%zu %lld %lld %d\n", j.acc.size(), (long long)j.out.evaluated, (long long)j.out.num_tests, j.out.status);
            }
            finished.push_back(std::move(j));
        } else {
            if (w != ji) pool.live[w] = std::move(j);
            ++w;
        }
    }
    pool.live.resize(w);
    c->cnt.t_host_merge_s += now_s() - tb2;
    return FW_OK;
}

int fwi_pool_round(fw_ctx *c, FwPool &pool, std::vector<FwPoolJob> &finished)
{
    int rc = fwi_pool_launch(c, pool);
    if (rc) return rc;
    return fwi_pool_collect(c, pool, finished);
}

int fwi_subsets_dispatch(fw_ctx *c, int64_t m, const FwJob *jobs, const int32_t *acc, int64_t acc_total, FwJobOut *out)
{
    (void)acc_total;
    FwPool pool;
    pool.want_zs = true;
    for (int64_t i = 0; i < m; ++i) fwi_pool_add(c, pool, jobs[i].X, jobs[i].Y, acc + jobs[i].acc_off, jobs[i].acc_len, i);
    std::vector<FwPoolJob> fin;
    while (!pool.live.empty()) {
        int rc = fwi_pool_round(c, pool, fin);
        if (rc) return rc;
    }
    for (FwPoolJob &j : fin) out[j.tag] = j.out;
    return FW_OK;
}

static double binom_d(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    double r = 1.0;
    for (int i = 1; i <= k; ++i) r = r * (double)(n - k + i) / (double)i;
    return std::floor(r + 0.5);
}

double fwi_alg_bytes(const fw_ctx *c, int a, int64_t evaluated)
{
    double bytes = 0.0, left = (double)evaluated;
    for (int s = c->P.max_k; s >= 1 && left > 0; --s) {
        const double cnt = std::min(left, binom_d(a, s));
        double per;
        if (c->P.kind == FW_FZ && !c->P.recursive_pcor)  // streaming variant: k + 2 sample columns per test (B_fzS)
            per = (double)(s + 2) * (double)c->P.n * 4.0 + 32.0;
        else if (c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ)  // gather variant (fz_nz: on the job-local matrix)
            per = 4.0 * (double)((s + 2) * (s + 1) / 2) + 32.0;
        else
            per = (double)(s + 2) * (double)c->P.n * (c->P.kind == FW_MI ? 1.0 : 2.0) / 8.0 + 32.0;
        bytes += cnt * per;
        left -= cnt;
    }
    return bytes;
}

extern "C" {

int fw_test_subsets_batch(fw_ctx *c, int64_t m, const int32_t *T, const int32_t *cand, const int64_t *accoff,
                          const int32_t *accflat, fw_subsets_result *out)
{
    CHECK_CTX(c);
    if (m < 0 || (m > 0 && (!T || !cand || !accoff || !out))) return fw_fail(c, FW_ERR_ARG, "fw_test_subsets_batch: NULL argument");
    if (m == 0) return FW_OK;
    // FW_FZ with the recursive form reads the resident Pearson matrix; with recursive_pcor = 0 the tests stream the sample columns
    // and need the data only (the same split as fw_test_batch)
    const bool needs_cor = c->P.kind == FW_FZ && c->P.recursive_pcor;
    if (needs_cor ? !c->have_cor : !c->have_data)
        return fw_fail(c, FW_ERR_STATE, "fw_test_subsets_batch: no %s resident", needs_cor ? "correlation matrix" : "data");
    if (c->P.max_k < 1) return fw_fail(c, FW_ERR_ARG, "fw_test_subsets_batch: max_k must be >= 1");
    std::vector<FwJob> jobs;
    std::vector<int64_t> map;
    jobs.reserve((size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        const int64_t len = accoff[i + 1] - accoff[i];
        if (len < 0 || len > INT32_MAX) return fw_fail(c, FW_ERR_ARG, "fw_test_subsets_batch: bad accoff at job %lld", (long long)i);
        if (!check_var(c, T[i]) || !check_var(c, cand[i])) return fw_fail(c, FW_ERR_ARG, "fw_test_subsets_batch: variable out of range in job %lld", (long long)i);
        for (int64_t q = accoff[i]; q < accoff[i + 1]; ++q)
            if (!check_var(c, accflat[q])) return fw_fail(c, FW_ERR_ARG, "fw_test_subsets_batch: accepted variable out of range in job %lld", (long long)i);
        if (len == 0) {  // tests.jl:285 sentinel
            fw_subsets_result r{};
            r.stat = NAN;
            r.pval = NAN;
            r.df = -1;
            r.suff_power = 1;
            r.status = FW_SUBSETS_EMPTY;
            r.n_zs = 0;
            r.num_tests = -1;
            r.frac = NAN;
            out[i] = r;
            continue;
        }
        FwJob j{};
        j.X = T[i];
        j.Y = cand[i];
        j.acc_off = accoff[i];
        j.acc_len = (int32_t)len;
        jobs.push_back(j);
        map.push_back(i);
    }
    std::vector<FwJobOut> jo(jobs.size());
    int rc = fwi_subsets_dispatch(c, (int64_t)jobs.size(), jobs.data(), accflat, accoff[m], jo.data());
    if (rc) return rc;
    for (size_t q = 0; q < jobs.size(); ++q) {
        const FwJobOut &o = jo[q];
        fw_subsets_result r{};
        r.stat = o.stat;
        r.pval = o.pval;
        r.df = o.df;
        r.suff_power = o.suff_power;
        r.status = o.status;
        r.n_zs = o.n_zs;
        for (int z = 0; z < FW_MAX_K; ++z) r.zs[z] = z < o.n_zs ? o.zs[z] : 0;
        r.num_tests = o.num_tests;
        double total = 0.0;  // tests.jl:313,327-332
        for (int s = c->P.max_k; s >= 1; --s) total += binom_d(jobs[q].acc_len, s);
        r.frac = total > 0 ? (double)o.num_tests / total : NAN;
        out[map[q]] = r;
        c->cnt.cond_tests_ref += o.num_tests;
        c->cnt.cond_tests_evaluated += o.evaluated;
        c->cnt.alg_bytes_subsets += fwi_alg_bytes(c, jobs[q].acc_len, o.evaluated);
        c->cnt.subsets_calls += 1;
    }
    return FW_OK;
}

int fw_get_counters(const fw_ctx *c, fw_counters *out)
{
    CHECK_CTX(c);
    if (!out) return fw_fail(c, FW_ERR_ARG, "fw_get_counters: NULL output");
    *out = c->cnt;
    return FW_OK;
}

int fw_selftest(fw_ctx *c, int which, uint64_t cases, uint64_t seed, uint64_t *mismatches)
{
    if (!c || !mismatches) return fw_fail(c, FW_ERR_ARG, "fw_selftest: NULL argument");
    *mismatches = 0;
    if (which == FW_SELFTEST_DIV) {
        unsigned long long bad = 0;
        const int rc = fwi_selftest_div(c, (unsigned long long)cases, (unsigned long long)seed, &bad);
        *mismatches = (uint64_t)bad;
        return rc;
    }
    return fw_fail(c, FW_ERR_ARG, "fw_selftest: unknown test %d", which);
}

int fw_reset_counters(fw_ctx *c)
{
    CHECK_CTX(c);
    c->cnt = fw_counters{};
    return FW_OK;
}

}  // extern "C"
