// Internal declarations shared by the translation units of libflashweave_amd.so.
// Nothing here is part of the ABI (include/flashweave_amd.h is).
#pragma once
// The library is written for gfx950 (MI355X, CDNA4) ONLY: mi_level0_mfma_kernel issues v_mfma_scale_f32_32x32x64_f8f6f4 and the kernels
// size their static LDS for 160 KB per CU.  There is no dual path; another --offload-arch fails here instead of deep inside a kernel.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libflashweave_amd targets gfx950 only (make ARCH=gfx950)"
#endif
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flashweave_amd.h"

struct FwDevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct FwPinned {
    void *ptr = nullptr;
    size_t cap = 0;
};

// Job descriptor for the test_subsets kernels (device layout)
struct FwJob {
    int32_t X;        // T
    int32_t Y;        // candidate
    int64_t acc_off;  // offset into the flat accepted array
    int32_t acc_len;  // |accepted|
    int32_t pad;
};

// Device-side result of one job (mirrors fw_subsets_result without frac)
// Conditioning sets of 6 and 7 variables (ABI 6: FW_MAX_K = 7; tests.jl:311-343 has no cap) take general-form kernels through the host job
// pool; every table kernel, the device rounds and the persistent discrete kernel size their arrays for FW_MAX_K_FAST.
#define FW_MAX_K_FAST 5

struct FwJobOut {
    double stat;
    double pval;
    int64_t num_tests;  // reference-equivalent count
    int64_t evaluated;  // tests evaluated by the kernel (speculation included)
    int32_t df;
    int32_t suff_power;
    int32_t status;
    int32_t n_zs;
    int32_t zs[FW_MAX_K];
    int32_t pad[3];
};

// One segment of a job: ranks [start, end) of the job's subset enumeration (device layout)
struct FwSeg {
    int32_t X;
    int32_t Y;
    int64_t acc_off;
    int32_t acc_len;
    int32_t pad;
    uint64_t start;
    uint64_t end;
    // r06, fz device rounds: the target's LOCAL correlation matrix -- (tm_m x tm_m) Float32 over [T, its level-0 neighbours in ascending id
    // order], a copy of those entries of the p x p matrix (fw_devhiton.hip: dh_tmat_build_kernel) -- and the sorted ids; 0: none (host pool,
    // light targets): the kernel gathers from the p x p matrix
    uint64_t tm, tm_ids;
    int32_t tm_m, pad1;
};

#define FW_RANK_NONE (~0ull)
#define FW_TAB_A 512  // fz / fz_nz: largest |accepted| served by the LDS-table kernel (csrc/fw_fz.hip, FZ_TAB_CAP)
#ifndef FW_HK_A
#define FW_HK_A 88    // fz, max_k 4-5: largest |accepted| served by the level-2 table kernel (one (z1, z2) table <= FZ_HK_CAP)
#endif

// Per-job record of the HE-S (fz_nz) path: the correlation sub-matrix of a job is computed over the rows where both
// T and the candidate are non-zero (statfuns.jl:138-155 cor_subset!), lives in a device arena and is indexed locally
// (0 = T, 1 = candidate, 2 + i = accepted[i]).
struct FwNzJob {
    int32_t X, Y;
    long long acc_off;  // into the flat accepted array of the launch
    int32_t acc_len;
    int32_t m;          // acc_len + 2
    long long cor_off;  // offset (floats) of the m x m matrix in the arena
    int32_t nR;         // rows with X != 0 and Y != 0            (device-computed)
    int32_t pad;
    double zscale;      // sqrt(nR - 3) / 2, 0 if nR <= 3          (device-computed)
    double rxy;         // unrounded Float64 pair correlation of (X, Y) over R, statfuns.jl:114 (device-computed)
    double thr[4];      // |r| significance thresholds for this nR (device-computed)
};

// Device-side result of one segment.  The conditioning sets are recovered on the host from the ranks.
struct FwSegOut {
    uint64_t stop_rank;  // first rank in the segment that ends the job (non-significant or max_tests), else FW_RANK_NONE
    double stop_stat;
    double stop_pval;
    uint64_t best_rank;  // last rank attaining the maximum p in the segment (valid when there was no stop)
    double best_stat;
    double best_pval;
    int32_t stop_df;
    int32_t stop_power;
    int32_t best_df;
    int32_t pad;
    uint64_t evaluated;
};

// per-pool staging: own stream + events so that two pools can be in flight (host merges one while the GPU runs the other)
// significant level-0 pairs as the kernels leave them in device memory (i < j; statistic in Float64 or Float32)
struct FwL0Dev {
    const int32_t *i = nullptr, *j = nullptr;
    const double *stat64 = nullptr;
    const float *stat32 = nullptr;
    const double *pval = nullptr;
    size_t k = 0;
};

struct FwPoolBuf {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the segment kernel(s) of a launch
    hipEvent_t evd = nullptr;                 // after the device-to-host copy of the launch's results
    hipStream_t launch_stream = nullptr;      // stream the launches go to (= stream)
    FwPinned h_in, h_out;  // h_in = [FwSeg x ns | accepted ints]
    FwDevBuf d_in, d_out;
    FwDevBuf d_acc;  // arena of the live jobs' accepted lists (uploaded once per job, not once per round)
};

struct FwComm;  // library-side RCCL communicator state (fw_rccl.cpp)

struct fw_ctx {
    fw_params P{};
    FwComm *comm = nullptr;  // fw_comm_init: the library's own communicator for target-sharded runs
    int64_t n_obs_min_eff = 0;
    mutable std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // ---- continuous (FW_FZ) ----
    float *d_data = nullptr;  // n x p column-major, as uploaded
    float *d_xc = nullptr;    // centred columns, [p_pad][n_pad], zero padded
    float *d_sd = nullptr;    // sqrt(sum xc^2) per column, [p_pad]
    float *d_cor = nullptr;   // p x p (symmetric)
    bool cor_external = false;  // d_cor is caller-owned device memory (fw_use_cor_buffer): never freed here
    int64_t cor_capacity = 0;   // ... and holds this many floats
    int cor_rows_rank = -1, cor_rows_world = 0;  // what the last fw_compute_cor_mat_rows computed: rank / world / rows per rank
    int64_t cor_rows_per_rank = 0;               // (fw_cor_mat_allgather_comm checks its arguments against them)
    double *d_thr = nullptr;  // |r| significance thresholds of the segment kernel (fz_thresholds_kernel)
    double *d_fzs_stat = nullptr;  // recursive_pcor = 0: per column {mean, sum of squared deviations} in Float64 (fw_fzs.hip)
    uint64_t gram_epoch = 0;       // recursive_pcor = 0: bumped whenever the job-matrix arena loses its contents (new pool, reallocation, reset)
    bool have_fzs_stat = false;
    int n_pad = 0, p_pad = 0;
    bool have_data = false, have_cor = false;

    // ---- discrete (FW_MI / FW_MI_NZ) ----
    std::vector<int32_t> levels, max_vals;
    int L = 0;                   // maximum(max_vals) + 1
    int mi_view = 0;  // dense rules + mi_nz: test_subsets evaluates on the (T, candidate) row view of hiton.jl:41-50 (fw_set_row_views)
    int l0_rank = 0, l0_world = 1;  // fw_level0_sharded: this rank's share of the level-0 pair tiles (discrete kinds)
    double *d_gthr = nullptr;    // [df]: G^2 quantile with ccdf(Chisq(df), .) = alpha (significance without evaluating Q(a, x))
    int gthr_n = 0;
    int mi_nxy = 2;              // cells per stratum side of the conditional-test kernels (fw_mi_core.h): 3 only for "mi" on 3-valued data
    int W = 0;                   // 64-bit words per packed column
    uint64_t *d_nzbits = nullptr;  // [p][W] bit i of word w: sample 64w+i has value != 0
    uint64_t *d_hibits = nullptr;  // [p][W] value == 2 (second non-zero level); NULL when L == 2
    int32_t *d_levels = nullptr, *d_maxvals = nullptr;
    unsigned char *d_vals = nullptr;  // generic discrete form (a value above 2): one byte per (variable, sample), [p][n]; null otherwise
    bool mi_generic = false;
    // generic form, tables beyond the LDS of a wavefront (r05: more than 8 levels, or 4-8 levels at a max_k the LDS table does not hold):
    // one table per wavefront of a launch in device memory, launches cut so that the tables of one fit mig_gtab_bytes_max
    long long mig_gtab_words = 0;  // words of one table (0: the table lives in LDS)
    FwDevBuf d_mig_tab[3];  // [the engine's own stream, pool 0, pool 1]: two pools can be in flight
    bool fznz_lds_raised = false;  // fznz_submat_kernel's dynamic-LDS limit was raised on this context's device (a property of (function, device): one flag per context, set under the context's own launches)
    float *d_xlnx = nullptr;       // [x ln x | ln x] for x = 0..n (Float32 tables of the discrete level-0 screen)
    int32_t *d_firstnz = nullptr;  // per column: index of the first non-zero sample (n if none)

    // ---- level-0 result (host, CSR) ----
    bool have_level0 = false;
    std::vector<int64_t> nb_off;
    std::vector<int32_t> nb_idx;
    std::vector<double> nb_stat, nb_p;

    // ---- network result ----
    bool have_network = false;
    struct FwHostWorkers *host_workers = nullptr;  // host threads of the graph passes (fw_hiton.cpp), started at the first network that wants them
    std::vector<int32_t> e_src, e_dst;
    std::vector<double> e_w;
    std::vector<int64_t> pc_off;
    std::vector<int32_t> pc_idx;
    std::vector<double> pc_w, pc_p;

    fw_counters cnt{};

    // grow-only scratch
    FwDevBuf d_jobs, d_acc, d_out, d_tmp0, d_tmp1, d_tmp2, d_segs, d_segout, d_nzrecs, d_arena;
    FwDevBuf d_l0m_i, d_l0m_d;  // level-0 pairs merged over the ranks (fw_xchg.hip)
    FwDevBuf d_bh;  // scratch of the device-side BH / neighbour-list epilogue (fw_bh.hip)
    // device copies of the level-0 neighbour CSR (inside d_bh, valid until the next fw_level0); null after a host-side BH
    const long long *d_nb_off = nullptr;
    const int32_t *d_nb_idx = nullptr;
    const double *d_nb_stat = nullptr, *d_nb_p = nullptr;
    const int32_t *d_cand = nullptr;  // per variable: its neighbours in candidate order (ascending adjusted p, stable)
    unsigned long long l0_cap_hint = 0;  // most screened / significant level-0 pairs seen so far (sizes the next call's buffers)
    bool nb_host_valid = true;        // nb_idx / nb_stat / nb_p hold the current lists (nb_off always does)
#define FW_DH_MAX_CHAINS 4
    FwDevBuf d_dh[FW_DH_MAX_CHAINS];  // arenas of the device-resident HITON rounds (fw_devhiton.hip), one per concurrent chain
    FwPinned h_dh[FW_DH_MAX_CHAINS];  // their pinned flag pages
    hipStream_t dh_stream[FW_DH_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};  // streams of chains 1.. (chain 0: pb[0].stream)
    // r06: a high-priority stream per chain for the small kernels between two segment launches (step / compact / plan / fill), with the two events
    // that tie it to the chain's stream -- used where the segment launches are long (max_k > 3): a one-workgroup kernel then no longer queues behind
    // the thousands of pending workgroups of the OTHER chain's segment kernel
    hipStream_t dh_hp_stream[FW_DH_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t dh_hp_ev[FW_DH_MAX_CHAINS][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    FwPinned h_jobs, h_acc, h_out;
    FwPoolBuf pb[2];
};

// Tuning / test knobs (environment variables FW_*: DESIGN.md section 5 lists them) are read ONLY when FW_KNOBS=1 is set.  The
// compiled-in defaults are the product; a stray FW_* variable in a user's environment must not change what the library does.
// tests/conftest.py, bench.py (for its one-chain / host-seam passes) and the scripts under profiles/tools set FW_KNOBS=1.
inline const char *fw_knob(const char *name)
{
    const char *on = getenv("FW_KNOBS");
    return (on && on[0] == '1') ? getenv(name) : nullptr;
}

int fw_fail(const fw_ctx *ctx, int code, const char *fmt, ...);
int fw_dev_reserve(fw_ctx *ctx, FwDevBuf &b, size_t bytes);
int fw_pin_reserve(fw_ctx *ctx, FwPinned &b, size_t bytes);

#define FW_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fw_fail(ctx, FW_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                                  \
    } while (0)

// ---- fz (fw_fz.hip) ----
int fwi_fz_compute_cor(fw_ctx *ctx);
int fwi_fz_compute_cor_rows(fw_ctx *ctx, int rank, int world, int64_t *row0, int64_t *rows_per_rank);
int fwi_fz_level0(fw_ctx *ctx, std::vector<int32_t> &pi, std::vector<int32_t> &pj, std::vector<double> &stat,
                  std::vector<double> &pval, int64_t *m_reliable, FwL0Dev *dev);
int fwi_fz_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff,
                      const int32_t *zflat, fw_test_result *out);
int fwi_fz_segments(fw_ctx *ctx, int64_t nseg, int64_t nseg_tab, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb);

// ---- fz without a correlation matrix: streamed sample columns (fw_fzs.hip, fw_params.recursive_pcor = 0) ----
int fwi_fzs_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff, const int32_t *zflat,
                       fw_test_result *out);
int fwi_fzs_segments(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb,
                     const FwNzJob *recs_host, int64_t njobs, size_t arena_doubles);
int fwi_fzs_segments_nz(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb, int m_max);  // fz_nz, recursive_pcor = 0  // recs: one per launched job (FwSeg::pad indexes them); rec.nR != 0: matrix to be computed

// ---- HE-S / fz_nz (fw_fz.hip) ----
int fwi_fznz_upload(fw_ctx *ctx, const float *data);
int fwi_fznz_level0(fw_ctx *ctx, std::vector<int32_t> &pi, std::vector<int32_t> &pj, std::vector<double> &stat,
                    std::vector<double> &pval, int64_t *m_reliable, FwL0Dev *dev);
int fwi_fznz_submatrices(fw_ctx *ctx, int64_t njobs, const FwNzJob *recs_host, size_t arena_floats, const int32_t *d_acc,
                         hipStream_t stream, bool f64 = false);
void fwi_host_workers_free(fw_ctx *c);
#ifdef FW_FZ_FASTDBG
extern "C" void fwi_fz_fastdbg_print();  // fw_fz.hip: counters of the size-3 fast loop (profiling build only)
#endif
int fwi_mi_big_limits(fw_ctx *ctx, int k);  // FW_OK if discrete tests with k (6, 7) conditioning variables fit the large LDS table
int fwi_fznz_dev_limits(fw_ctx *ctx, int m_max);  // FW_OK if a job of m_max variables fits the sub-matrix kernel's LDS
int fwi_fznz_submatrices_dev(fw_ctx *ctx, int nslots, FwNzJob *d_recs, const int32_t *d_acc, float *d_arena, int m_max, bool any_long, hipStream_t stream);
int fwi_fznz_segments_dev(fw_ctx *ctx, unsigned grid, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, const unsigned *d_ns,
                          bool any_big, const unsigned *d_big, const FwNzJob *d_recs, const float *d_arena, hipStream_t stream);
int fwi_fznz_segments(fw_ctx *ctx, int64_t nseg, int64_t nseg_tab, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb);
int fwi_fznz_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff,
                        const int32_t *zflat, fw_test_result *out);

// ---- discrete (fw_mi.hip) ----
int fwi_mi_upload(fw_ctx *ctx, const int64_t *colptr, const int32_t *rowval, const int32_t *nzval);
int fwi_mi_level0(fw_ctx *ctx, std::vector<int32_t> &pi, std::vector<int32_t> &pj, std::vector<double> &stat,
                  std::vector<double> &pval, int64_t *m_reliable, FwL0Dev *dev);
int fwi_mi_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff,
                      const int32_t *zflat, fw_test_result *out);
int fwi_mi_segments(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb);

// algorithmic bytes of the first `evaluated` tests of a job with |accepted| = a (enumeration order: sizes max_k..1)
double fwi_alg_bytes(const fw_ctx *ctx, int a, int64_t evaluated);

// ---- asynchronous job pool (fw_core.cpp): every round evaluates one window of every live job in ONE launch ----
struct FwPoolJob {
    int32_t X = 0, Y = 0;
    int64_t tag = 0;     // caller's id (driver: target slot)
    int32_t aux = 0;     // driver: candidate index
    int32_t epoch = 0;   // driver: accepted-set epoch of the owner when the job was posted (stale jobs are dropped)
    bool hold = false;     // driver: do not launch further windows for now (speculative job past its first window)
    bool launched = false; // set by fwi_pool_launch for the jobs that are part of the pending window
    bool no_zs = false;    // the returned result has no conditioning set (fz_nz job without a test)
    std::vector<int32_t> acc;
    int64_t acc_dev_off = -1;  // offset (ints) of this job's accepted list in the pool buffer's device arena, -1 = not uploaded
    int64_t gram_off = -1;     // recursive_pcor = 0: offset (doubles) of the job's correlation matrix in ctx->d_arena, valid while
    uint64_t gram_epoch = 0;   // ... gram_epoch == ctx->gram_epoch (the arena was neither reallocated nor reset since)
    uint64_t N = 0, next = 0, width = 0;
    double best_p = -1.0, best_stat = 0.0;
    uint64_t best_rank = 0;
    int32_t best_df = 0;
    bool done = false;
    FwJobOut out{};
};
struct FwPool {
    std::vector<FwPoolJob> live;
    std::vector<int64_t> seg_job;
    std::vector<FwNzJob> nzrecs;  // fz_nz: job records of the pending launch
    int buf = 0;            // which ctx->pb[] this pool stages through
    bool want_zs = false;   // recover the conditioning set of each returned result from its rank (ABI path)
    const std::vector<int32_t> *owner_epoch = nullptr;  // if set: jobs whose epoch != (*owner_epoch)[tag] are cancelled
    int64_t dropped_evaluated = 0;                      // tests already evaluated for jobs cancelled before they finished
    double dropped_alg_bytes = 0.0;
    bool inflight = false;  // a window launch is pending (fwi_pool_launch without fwi_pool_collect)
    size_t ns = 0;
    uint64_t launched_ranks = 0;  // ranks of the pending / last launch (window-growth policy)
    size_t arena_top = 0;         // ints used in pb.d_acc (accepted lists stay resident while their job lives)
    size_t gram_top = 0;          // recursive_pcor = 0: doubles used in ctx->d_arena by this pool's job matrices (kept across rounds)
    uint64_t gram_epoch = 0;      // epoch of ctx->d_arena this pool's offsets belong to (0: none yet)
    double t_launch = 0.0;
};
int fwi_pool_launch(fw_ctx *ctx, FwPool &pool);                                    // asynchronous
int fwi_pool_collect(fw_ctx *ctx, FwPool &pool, std::vector<FwPoolJob> &finished);  // waits + merges
int fwi_pool_add(fw_ctx *ctx, FwPool &pool, int32_t X, int32_t Y, const int32_t *acc, int a, int64_t tag);
int fwi_pool_round(fw_ctx *ctx, FwPool &pool, std::vector<FwPoolJob> &finished);

// device-side Benjamini-Hochberg + neighbour CSR (fw_bh.hip); fills ctx->nb_off / nb_idx / nb_stat / nb_p
int fwi_bh_csr_device(fw_ctx *ctx, const FwL0Dev &in, int64_t m_reliable);
// device-resident all-gather of the ranks' significant level-0 pairs (fw_xchg.hip)
int fwi_l0_exchange_dev(fw_ctx *c, const fw_dev_exchange *x, int world, const FwL0Dev &local, int64_t m_local, FwL0Dev *merged, int64_t *m_sum);
int fwi_selftest_div(fw_ctx *ctx, unsigned long long cases, unsigned long long seed, unsigned long long *mismatches);  // fw_fz.hip
void fwi_comm_free(fw_ctx *ctx);  // fw_rccl.cpp
int fwi_nb_host_ensure(fw_ctx *ctx);  // download partners / statistics / adjusted p if only the device holds them

// ---- device-resident HITON rounds (fw_devhiton.hip, FW_FZ) ----
struct FwDhTarget {
    int32_t T = 0;
    std::vector<int32_t> cands;  // interleaving candidates in hiton.jl:211-217 order (unused when nc_dev >= 0)
    int32_t nc_dev = -1;         // >= 0: take the first nc_dev entries of ctx->d_cand at nb_off[T] (device-built order)
    const int32_t *wl = nullptr; // sorted whitelist (feed-forward), may be null
    int wl_n = 0;
};
struct FwDhResult {  // PC of one target in insertion order: entries [off, off + n) of the run's flat arrays
    int64_t off = 0;
    int32_t n = 0;
};
struct FwDhFlat {  // (one allocation per run instead of three per target: 150 000 small vectors were 10 ms of a cfg4 pass)
    std::vector<int32_t> key;
    std::vector<double> stat, pval;
};
int fwi_devhiton_run(fw_ctx *ctx, const std::vector<FwDhTarget> &in, std::vector<FwDhResult> &out, FwDhFlat &flat, int chain = 0);
// the whole feed-forward schedule of the discrete kinds on the device: whitelists built between the launches, one download (r05)
int fwi_devhiton_mi_schedule(fw_ctx *ctx, const int32_t *sched, int nt, int R, bool feed_forward, std::vector<int32_t> &all_t,
                             std::vector<int32_t> &all_u, std::vector<double> &all_s, std::vector<double> &all_p);
int fwi_mi_segments_dev(fw_ctx *ctx, unsigned grid, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, const unsigned *d_ns,
                        hipStream_t stream);
int fwi_fz_thresholds(fw_ctx *ctx, hipStream_t stream, double *zscale);  // ensures ctx->d_thr (fz_thresholds_kernel)
int fwi_fz_segments_dev(fw_ctx *ctx, unsigned grid, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, const unsigned *d_ns,
                        bool any_big, const unsigned *d_big, hipStream_t stream);

// ---- host driver (fw_hiton.cpp) ----
int fwi_subsets_dispatch(fw_ctx *ctx, int64_t m, const FwJob *jobs_host, const int32_t *acc_host, int64_t acc_total,
                         FwJobOut *out_host);
