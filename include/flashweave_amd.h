/*
 * flashweave_amd.h -- C ABI of the MI355X-native conditional-independence engine for FlashWeave.
 *
 * This is the drop-in boundary for ONE path of the reference (FlashWeave.jl, /root/reference): the
 * per-pair CI test batch + the level-0 all-pairs stage (src/tests.jl, src/contingency.jl, src/statfuns.jl).
 * The reference has no FFI for it; the seam is Julia multiple dispatch on `test_obj::AbstractTest`
 * (src/types.jl:57-59).  Each entry point below names the reference call edge it replaces, and
 * INTEGRATION.md shows the `ccall` methods a maintainer would add for a `GpuTest <: AbstractTest`.
 *
 * Conventions
 *   - plain C, no C++/torch types; all buffers are caller-owned host memory unless stated otherwise;
 *     the context owns every device allocation.
 *   - variable indices are 0-based on this ABI (the Julia shim subtracts 1).
 *   - data is n samples (rows) x p variables (columns), column-major, exactly as Julia holds it.
 *   - every function returns FW_OK (0) or a negative error code; fw_last_error() gives the message.
 *     Invalid input never aborts the process (reference: error()/@assert, e.g. src/learning.jl:72).
 *   - a context is used by one host thread at a time; calls are blocking; one context per GPU.
 *   - there is NO CPU fallback: if no gfx950 device is usable, fw_ctx_create fails with FW_ERR_DEVICE.
 */
#ifndef FLASHWEAVE_AMD_H
#define FLASHWEAVE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FW_ABI_VERSION 6 /* 2: fw_params.recursive_pcor, fw_level0_sharded; 3: device-resident exchange (fw_dev_exchange), sharded cor; 4: fw_params.no_cor_mat; 5: fw_selftest, fw_counters.gram_*, fw_comm_* (library-side RCCL); 6: FW_MAX_K 5 -> 7 (fw_subsets_result.zs grows) */

/* test kinds: src/types.jl:61-72 (test_name "mi" / "mi_nz" / "fz") */
#define FW_MI 0
#define FW_MI_NZ 1
#define FW_FZ 2
#define FW_FZ_NZ 3 /* "fz_nz": FlashWeaveHE-S, zero-ignoring Fisher-z tests */

#define FW_OK 0
#define FW_ERR_ARG (-1)     /* invalid argument */
#define FW_ERR_DEVICE (-2)  /* no usable gfx950 device / HIP runtime error */
#define FW_ERR_STATE (-3)   /* call order violated (e.g. level-0 before data upload) */
#define FW_ERR_NOBS (-4)    /* n_obs_min exceeds the number of samples (src/learning.jl:66-73) */
#define FW_ERR_LIMIT (-5)   /* a documented capacity limit was exceeded (max_k, table size) */
#define FW_ERR_NOMEM (-6)

#define FW_MAX_K 7 /* largest conditioning-set size (tests.jl:311-343 has no cap): up to 5 on the table / persistent kernels, 6 and 7 (ABI 6) on
                      general-form kernels through the host job pool; not with recursive_pcor = 0 */

typedef struct fw_ctx fw_ctx;

/* Engine parameters; defaults follow learn_network / LGL (src/learning.jl:203-214,466-473). */
typedef struct fw_params {
    int32_t kind;      /* FW_MI / FW_MI_NZ / FW_FZ / FW_FZ_NZ */
    int32_t n;         /* samples (rows) */
    int32_t p;         /* variables (columns) */
    int32_t device;    /* HIP device ordinal */
    int32_t max_k;     /* default 3; 0 = univariate network only */
    int32_t hps;       /* default 5 (heuristic power size, src/tests.jl:5-6) */
    int32_t fdr;       /* default 1: Benjamini-Hochberg on level-0 p-values (src/tests.jl:521-529) */
    int32_t dense_rules; /* FW_MI / FW_MI_NZ only.  0 (default): contingency tables follow the SparseMatrixCSC methods
                          * (src/contingency.jl:80-480), what learn_network uses (make_sparse = true).  1: the dense
                          * Matrix methods (src/contingency.jl:7-56 + level_map! src/misc.jl:162-184): every row is
                          * visited and levels_z = number of distinct Z keys over all rows.  The cell counts
                          * are the same; only levels_z (power verdict) can differ, and only for FW_MI_NZ.
                          * fw_learn_network with FW_MI_NZ + dense_rules tests on the per-(target, candidate) row views of
                          * src/hiton.jl:41-50 (see fw_set_row_views). */
    int64_t n_obs_min; /* default -1 = automatic (src/learning.jl:51-64, fires for every test kind) */
    int64_t max_tests; /* default 10_000_000 per (T, candidate) pair (src/learning.jl:205) */
    double alpha;      /* default 0.01 */
    int32_t recursive_pcor; /* FW_FZ and FW_FZ_NZ (r04: for FW_FZ_NZ the correlations of a job's ROW VIEW, hiton.jl:85, in Float64,
                             * conditioned as StatsBase.partialcor does, with the view's own sample size).  1 (default): conditional tests are recursive partial correlations on the resident
                             * Pearson matrix (pcor_rec, src/statfuns.jl:23-75).  0: no correlation matrix is used for them --
                             * every test streams its k + 2 sample columns from HBM and computes the partial correlation from
                             * the data (pcor -> StatsBase.partialcor, src/statfuns.jl:19-21; the FzTestCond with an empty
                             * cor_mat of src/tests.jl:253, learn_network(recursive_pcor = false)).  Level 0 keeps the matrix. */
    int32_t no_cor_mat;     /* FW_FZ with recursive_pcor = 0 only.  0 (default, the reference's dense_cor = true): level 0 reads the
                             * resident p x p Pearson matrix (src/learning.jl:42-45).  1 (dense_cor = false, src/tests.jl:118-147 with an
                             * empty cor_mat): no matrix exists at any time -- fw_level0 multiplies the centred columns tile by tile
                             * on the matrix cores and screens every tile in the epilogue (the same Float32 correlations and
                             * thresholds as the matrix path, hence the same network as no_cor_mat = 0), conditional tests come
                             * from the data (recursive_pcor = 0).  p is then bounded by the data (2 x n x p floats), not by p^2.
                             * fw_compute_cor_mat / fw_set_cor_mat / fw_get_cor_mat / fw_use_cor_buffer fail with FW_ERR_STATE. */
} fw_params;

/* One TestResult (src/types.jl:140-145) */
typedef struct fw_test_result {
    double stat;
    double pval;
    int32_t df;
    int32_t suff_power; /* 0 / 1 */
} fw_test_result;

/* Return of test_subsets (src/tests.jl:281-346): (TestResult, Zs, num_tests, fraction) */
#define FW_SUBSETS_EMPTY 0    /* Z_total was empty: sentinel (NaN, NaN, -1, true), (-1,), -1, NaN (src/tests.jl:285) */
#define FW_SUBSETS_STOPPED 1  /* returned at the first non-significant test or at max_tests (:326-336) */
#define FW_SUBSETS_ALL_SIG 2  /* every subset significant: the max-p result (:338-345) */
typedef struct fw_subsets_result {
    double stat;
    double pval;
    int32_t df;
    int32_t suff_power;
    int32_t status;          /* FW_SUBSETS_* */
    int32_t n_zs;            /* length of zs */
    int32_t zs[FW_MAX_K];    /* conditioning set of the returned result (variable ids) */
    int32_t reserved0;
    int64_t num_tests;       /* tests executed in the reference's sequential order (-1 for EMPTY) */
    double frac;             /* num_tests / total number of subsets */
} fw_subsets_result;

/* Counters the metric needs (the reference has none) */
typedef struct fw_counters {
    int64_t level0_tests;        /* p(p-1)/2 pair tests issued */
    int64_t cond_tests_ref;      /* sum of num_tests over all test_subsets calls (reference-equivalent) */
    int64_t cond_tests_evaluated;/* tests actually evaluated on the device (includes speculation) */
    int64_t subsets_calls;       /* number of (T, candidate) jobs */
    int64_t kernel_launches;
    int64_t subsets_launches;    /* launches of the test_subsets segment kernel */
    double t_level0_s;           /* wall seconds inside fw_level0 */
    double t_level0_host_s;      /* of which: host-side BH + neighbour-list construction */
    double t_cond_s;             /* wall seconds inside the conditional stage of fw_learn_network */
    double t_dev_subsets_s;      /* HIP-event seconds of the test_subsets kernels (sum) */
    double t_host_advance_s;     /* host: HITON-PC state machines + job posting */
    double t_host_build_s;       /* host: segment construction + staging */
    double t_host_launch_s;      /* host: enqueueing copies + kernel */
    double t_host_wait_s;        /* host: waiting for the device (copies + kernel + sync) */
    double t_host_merge_s;       /* host: in-order merge of segment outputs */
    double alg_bytes_subsets;    /* algorithmic bytes of the evaluated conditional tests:
                                    fz: 4*C(k+2,2)+32 per test of order k; discrete: (k+2)*n*b/8+32, b = 1 (mi) / 2 (mi_nz) */
    /* ABI 5 -- recursive_pcor = 0 with job-local correlation matrices (the default form of that variant): its own algorithmic unit.
     * A (T, candidate | subsets of a accepted variables) job computes ONE (a+2) x (a+2) matrix from its columns and every test
     * conditions a sub-matrix of it, so what must cross HBM is the job's columns once, and the contraction is 2 n C(a+2, 2) flops;
     * alg_bytes_subsets (columns per TEST) stays the unit of the streamed form (FW_FZS_GRAM=0, explicit test batches). */
    int64_t gram_jobs;           /* job matrices computed */
    double gram_alg_bytes;       /* sum over them of (a + 2) * n * 4 */
    double gram_alg_flops;       /* sum over them of 2 * n * C(a + 2, 2) */
    /* ABI 6 -- level 0 of the discrete kinds on the matrix cores (mi_level0_mfma_kernel): the binary Gram product, priced against the
     * dense fp4 peak */
    double l0_mfma_flops;        /* 2 x multiply-adds issued: tiles x 256 x 256 plane rows x 64 W samples (padded words included) */
    double t_l0_mfma_s;          /* HIP-event seconds of the kernel's launches */
} fw_counters;

/* ---- lifecycle -------------------------------------------------------------------------------- */

/* Fills *params with the reference defaults for `kind`. */
void fw_params_default(fw_params *params, int32_t kind, int32_t n, int32_t p);

/* replaces: make_test_object (src/misc.jl:34-45) + the per-worker data hand-over (src/interleaved.jl:90-93).
 * On failure *out is NULL and fw_last_error(NULL) describes the problem. */
int fw_ctx_create(const fw_params *params, fw_ctx **out);
int fw_ctx_destroy(fw_ctx *ctx);
const char *fw_last_error(const fw_ctx *ctx); /* ctx may be NULL: last creation error of this thread */
int fw_abi_version(void);

/* ---- data ------------------------------------------------------------------------------------- */

/* FW_FZ: the normalised dense matrix (Matrix{Float32}, n x p column-major) the reference hands to
 * cor() in prepare_lgl (src/learning.jl:42-45).  FW_FZ_NZ: the clr_nz matrix with zeros = absences (the reference
 * holds it as SparseMatrixCSC{Float32}; the Julia shim densifies it once). */
int fw_set_data_dense_f32(fw_ctx *ctx, const float *data);

/* FW_MI / FW_MI_NZ: SparseMatrixCSC{Int32,Int64} as produced by normalize_data (make_sparse = true).
 * colptr has p+1 entries; rowval is 0-based and sorted within each column; values are 1..61, stored
 * zeros are not allowed.  Values 1..2 everywhere (presence / absence, the two bins of binned_nz_clr): two bit planes per variable, the
 * fast kernels.  A value above 2 anywhere (meta variables with make_onehot = false, src/preprocessing.jl:42-117): the GENERIC form -- one
 * byte per value, tables of L x L x L^k cells as src/types.jl:98-117 sizes them, L = maximum + 1 <= 62; a test's table of L^max_k (L^2 + 1) words sits in LDS up to 3840
 * words and in device memory up to 64 M words (r05), else FW_ERR_LIMIT; same entry points and results, HITON-PC through the host job pool.
 * Also computes levels / max_vals (src/misc.jl:64-97). */
int fw_set_data_csc_i32(fw_ctx *ctx, const int64_t *colptr, const int32_t *rowval, const int32_t *nzval);

/* FW_MI / FW_MI_NZ, dense input (Matrix{Int32}, n x p column-major, values 0..61, see above); converted on the host to
 * the same packed device layout, i.e. evaluated with the SPARSE-path semantics (levels_z rules). */
int fw_set_data_dense_i32(fw_ctx *ctx, const int32_t *data);

int fw_get_levels(const fw_ctx *ctx, int32_t *levels, int32_t *max_vals); /* p entries each */

/* FW_FZ: supply a precomputed Pearson matrix (p x p, Float32) instead of computing it on the device --
 * mirrors `cor_mat` being an argument of pw_univar_neighbors / si_HITON_PC (src/tests.jl:441, src/hiton.jl:284). */
int fw_set_cor_mat(fw_ctx *ctx, const float *cor_mat);
/* replaces: cor(data_dense) -> Matrix{Float32} (src/learning.jl:44).  MFMA kernel; result stays resident. */
int fw_compute_cor_mat(fw_ctx *ctx);
int fw_get_cor_mat(const fw_ctx *ctx, float *cor_mat_out); /* p*p floats */

/* Exchange callback for target-sharded runs: called once per feed-forward round with this rank's newly found
 * directed neighbour entries (target, neighbour, weight-stat, p); must return the concatenation over all ranks
 * (rank order) through *out_* buffers allocated by the callee and valid until the next call.  NULL for
 * world_size = 1.  (RCCL all_gather in bench.py; gloo in the CPU tests.) */
typedef int (*fw_allgather_fn)(void *user, int64_t n_local, const int32_t *tgt, const int32_t *nbr, const double *stat,
                               const double *pval, int64_t *n_total, const int32_t **tgt_all, const int32_t **nbr_all,
                               const double **stat_all, const double **pval_all);

/* ---- level 0 ---------------------------------------------------------------------------------- */

/* replaces: pw_univar_neighbors (src/tests.jl:436-532) incl. the power/NaN rules and
 * benjamini_hochberg! (src/statfuns.jl:326-350).  Runs all p(p-1)/2 univariate tests on the device and
 * keeps the neighbour lists in the context.  *nnz_out = total number of (directed) neighbour entries. */
int fw_level0(fw_ctx *ctx, int64_t *nnz_out);
/* Same result, for target-sharded runs (one process per GPU): this rank screens only its share of the pair tiles (discrete
 * kinds; the Fisher-z kinds stay replicated) and the significant pairs are exchanged through the same callback type as the
 * per-round neighbour sets (fw_allgather_fn above: entries (i, j, stat, p)); Benjamini-Hochberg and the neighbour lists are
 * then built on every rank from the merged list.  The role of the reference's master process, which runs
 * pw_univar_neighbors once and ships the result to the workers (src/learning.jl:130-150, src/interleaved.jl:90-93). */
int fw_level0_sharded(fw_ctx *ctx, int32_t rank, int32_t world_size, fw_allgather_fn allgather, void *user, int64_t *nnz_out);
/* Device-resident exchange for target-sharded runs: the caller owns the communication buffers in DEVICE memory (torch tensors in
 * bench.py, all-gathered by RCCL over xGMI), the library packs into / unpacks from them with its own kernels -- no host copy of the
 * payload.  prepare(): every rank reports its record count and one auxiliary integer; the callee all-gathers both (counts[],
 * aux[], world_size entries each), makes room for cap = max(count) records of rec_bytes bytes in a send buffer and for world_size
 * blocks of cap records in a receive buffer, and returns their device addresses.  exchange(): all-gather the send buffers
 * (cap * rec_bytes bytes per rank) into the receive buffer in rank order; returns when the data is in place. */
typedef struct fw_dev_exchange {
    void *user;
    int (*prepare)(void *user, int64_t n_local, int64_t aux_local, int32_t rec_bytes, void **d_send, void **d_recv, int64_t *counts,
                   int64_t *aux, int64_t *cap_records);
    int (*exchange)(void *user);
} fw_dev_exchange;
/* fw_level0_sharded with the payload kept on the device: this rank screens its share of the pair tiles (discrete kinds), the
 * significant pairs are packed into the caller's send buffer, gathered, and Benjamini-Hochberg + the neighbour lists are built
 * from the gathered buffer on every rank.  (The Fisher-z kinds stay replicated: see fw_compute_cor_mat_rows for their share.) */
int fw_level0_sharded_dev(fw_ctx *ctx, int32_t rank, int32_t world_size, const fw_dev_exchange *exchange, int64_t *nnz_out);

/* FW_FZ, row-block sharding of cor(data_dense) (src/learning.jl:44) over the ranks: fw_use_cor_buffer makes the context keep its
 * p x p matrix in caller-owned device memory (capacity in floats >= world_size * rows_per_rank * p, see below);
 * fw_compute_cor_mat_rows computes the rows [*row0, *row0 + *rows_per_rank) of this rank (whole rows, no mirrored writes:
 * rows_per_rank = 128 * ceil(ceil(p / 128) / world_size), the same on every rank) and leaves the rest untouched; the caller then
 * all-gathers the row blocks in place (contiguous: rows_per_rank * p floats per rank) and calls fw_cor_mat_ready. */
int fw_use_cor_buffer(fw_ctx *ctx, void *d_cor, int64_t capacity_floats);
int fw_compute_cor_mat_rows(fw_ctx *ctx, int32_t rank, int32_t world_size, int64_t *row0, int64_t *rows_per_rank);
int fw_cor_mat_ready(fw_ctx *ctx);

/* Neighbour lists as CSR: off[p+1]; idx/stat/adj_p have nnz entries, partners ascending per variable;
 * adj_p is the BH-adjusted p-value when fdr = 1 (src/tests.jl:372-388). */
int fw_level0_get(const fw_ctx *ctx, int64_t *off, int32_t *idx, double *stat, double *adj_p);

/* ---- the per-pair test batch ------------------------------------------------------------------- */

/* replaces: test(X, Y, Zs, data, test_obj, ...) (src/tests.jl:28,108 for empty Zs; :184,:250 otherwise).
 * Test i is (X[i], Y[i] | zflat[zoff[i] .. zoff[i+1])), Z order preserved.  zoff has m+1 entries. */
int fw_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff,
                  const int32_t *zflat, fw_test_result *out);

/* replaces: test_subsets(T, candidate, accepted, data, test_obj, max_k, alpha; hps, n_obs_min, max_tests)
 * (src/tests.jl:281-346) for m independent (T, candidate, accepted) jobs -- the call edge hiton.jl:100.
 * Job i conditions on accflat[accoff[i] .. accoff[i+1]) (acceptance order preserved, duplicates allowed).
 * Enumeration order, early exit and the `>=` max-p rule are the reference's; num_tests is the count the
 * sequential reference would have executed. */
int fw_test_subsets_batch(fw_ctx *ctx, int64_t m, const int32_t *T, const int32_t *cand, const int64_t *accoff,
                          const int32_t *accflat, fw_subsets_result *out);

/* FW_MI_NZ with dense_rules = 1 only.  hiton.jl hands test_subsets not the data but a row VIEW of it: the rows where T is
 * non-zero if T has more than two levels, and likewise for the candidate (prepare_nzdata, src/hiton.jl:41-50,85,193 ->
 * needs_nz_view, src/misc.jl:103-107).  on = 1: fw_test_subsets_batch evaluates job (T, candidate, .) on that view (what a
 * GpuTest shim called from hiton.jl:100 needs); on = 0 (default): on every row of the uploaded matrix (what a direct
 * test_subsets(T, candidate, Z, data, ...) call on the full matrix computes).  fw_learn_network always uses the views. */
int fw_set_row_views(fw_ctx *ctx, int32_t on);

/* ---- host driver: the caller side, for hosts without Julia ----------------- */

typedef struct fw_learn_opts {
    int32_t feed_forward;  /* default 1 (src/learning.jl:469) */
    int32_t round_size;    /* targets per feed-forward round; 1 = the reference's deterministic single_il schedule;
                              0 = one round (no whitelist can form: identical to feed_forward = 0 / parallel="single") */
    int32_t rank;          /* this process' rank in a target-sharded run (0 for single GPU) */
    int32_t world_size;    /* number of ranks; the targets of a round are dealt by estimated work (heaviest first to the least loaded rank) */
    int32_t max_targets;   /* > 0: stop after this many targets of the schedule (sampling; 0 = all) */
    int32_t reserved0;
} fw_learn_opts;

/* replaces: LGL minus normalisation (src/learning.jl:203-279): level 0 (if not yet run), target ordering,
 * HITON-PC per target (src/hiton.jl:283-400) in level-synchronous batches over fw_test_subsets_batch,
 * feed-forward rounds (src/interleaved.jl:112-183), make_weights + make_symmetric_graph (src/misc.jl:137-272).
 * *n_edges_out = number of undirected edges; fetch them with fw_network_get. */
int fw_learn_network(fw_ctx *ctx, const fw_learn_opts *opts, fw_allgather_fn allgather, void *user,
                     int64_t *n_edges_out);
/* The same with the per-round exchange through a fw_dev_exchange (see fw_level0_sharded_dev): the round's directed entries are packed
 * into 24-byte records (int32 target, int32 neighbour, Float64 statistic, Float64 p) by the library, copied into the caller's device
 * send buffer, all-gathered by the caller's collective and unpacked from the gathered buffer -- the host language only runs the
 * collective (r02's callback packed and unpacked in numpy: ~0.9 ms per round of a cfg3 pass). */
int fw_learn_network_dev(fw_ctx *ctx, const fw_learn_opts *opts, const fw_dev_exchange *exchange, int64_t *n_edges_out);
/* ---- library-side collectives (ABI 5): the same exchanges on a communicator the LIBRARY owns ------------------------------------
 * replaces: the master <-> worker message loop of src/interleaved.jl:112-183 for one process per GPU.  RCCL (librccl.so.1, reached
 * through dlopen: single-GPU users never load it) on the context's stream; the host language only carries the 128-byte rendezvous
 * id from rank 0 to the others (torch.distributed / MPI / Distributed.jl) and never sees a payload.
 *   fw_comm_unique_id   rank 0: ncclGetUniqueId
 *   fw_comm_init        every rank, after fw_ctx_create on ITS device: ncclCommInitRank (one rank per device)
 *   fw_level0_comm      = fw_level0_sharded_dev with the library's all-gather (discrete kinds; Fisher-z kinds: plain fw_level0)
 *   fw_cor_mat_allgather_comm  the in-place all-gather of the row blocks of fw_compute_cor_mat_rows (then fw_cor_mat_ready)
 *   fw_learn_network_comm      = fw_learn_network_dev with the library's all-gather per feed-forward round; opts->rank /
 *                              world_size are taken from the communicator
 *   fw_comm_stats       exchanges so far: calls, collectives, directed entries, gathered bytes, seconds inside them */
#define FW_COMM_ID_BYTES 128
int fw_comm_unique_id(uint8_t *id128);
int fw_comm_init(fw_ctx *ctx, const uint8_t *id128, int32_t rank, int32_t world_size);
int fw_comm_destroy(fw_ctx *ctx);
int fw_comm_stats(const fw_ctx *ctx, int64_t *calls, int64_t *collectives, int64_t *entries, int64_t *bytes, double *seconds);
int fw_level0_comm(fw_ctx *ctx, int64_t *nnz_out);
int fw_cor_mat_allgather_comm(fw_ctx *ctx, int64_t rows_per_rank);
int fw_learn_network_comm(fw_ctx *ctx, const fw_learn_opts *opts, int64_t *n_edges_out);

int fw_network_get(const fw_ctx *ctx, int32_t *src, int32_t *dst, double *weight); /* src < dst */
/* directed per-target results (state_results of every HitonState): CSR over targets */
int fw_network_get_directed(const fw_ctx *ctx, int64_t *off, int32_t *idx, double *weight, double *pval);

/* ---- normalisation front-end on the device -------------------------------------- */

/* replaces: normalize_data / preprocess_data (src/preprocessing.jl:412-563) for a count table without meta variables:
 * filter_by_variance (:367-409), then by kind  FW_FZ: clr_adapt (:133-214)   FW_FZ_NZ: clr_nz (:192-207)   FW_MI: binary (:475-490)
 * FW_MI_NZ: binned_nz_clr (clr_nz, then per column the tied ranks of the non-zero entries in two bins, :217-291,492-521; the
 * column sort runs in LDS up to 16 384 samples and through device memory beyond).  counts: n x p column-major Int32.  Outputs (host buffers sized for n x p): out_f32 (FW_FZ / FW_FZ_NZ) or
 * out_i32 (FW_MI / FW_MI_NZ), column-major *n_out x *p_out; row_mask[n] / col_mask[p] = kept samples / variables.  No context
 * needed: create one with the resulting shape afterwards.  Meta variables (a handful of columns) are prepared by the caller
 * (one-hot, discretisation: preprocess.py) and appended. */
int fw_normalize_counts(int32_t device, int32_t kind, int32_t n, int32_t p, const int32_t *counts, float *out_f32, int32_t *out_i32,
                        uint8_t *row_mask, uint8_t *col_mask, int32_t *n_out, int32_t *p_out);

/* Device self-test of hand-written arithmetic sequences that replace a compiler-generated IEEE sequence and must return the
 * same bits.  which = FW_SELFTEST_DIV: the unscaled Float64 division of the NaN-free partial-correlation path (statfuns.jl:44-62
 * `/`) against the compiler's division on `cases` hashed operand pairs of that path's ranges; *mismatches = pairs that differ. */
#define FW_SELFTEST_DIV 1
int fw_selftest(fw_ctx *ctx, int which, uint64_t cases, uint64_t seed, uint64_t *mismatches);

int fw_get_counters(const fw_ctx *ctx, fw_counters *out);
int fw_reset_counters(fw_ctx *ctx);
/* effective n_obs_min after the automatic rule */
int64_t fw_effective_n_obs_min(const fw_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* FLASHWEAVE_AMD_H */
