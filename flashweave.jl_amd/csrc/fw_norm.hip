// Normalisation front-end on the device: the count matrix goes to HBM once and comes out normalised,
// ready for fw_set_data_*.  Mirrors preprocess_data for a table without meta variables (reference src/preprocessing.jl):
//   filter_by_variance                  :367-409   zero-variance columns, then samples without reads
//   "fz"     clr_adapt                  :133-214   adaptive pseudo-counts (adaptive_pseudocount!) + centred log-ratio
//   "fz_nz"  clr_nz                     :192-207   log(x / geometric mean of the row's non-zeros), zeros stay zeros
//   "mi"     binary                     :475-490   presence / absence, columns with exactly two levels
//   "mi_nz"  binned_nz_clr              :217-291,492-521  clr_nz, then per column the tied (average) ranks of the non-zero entries,
//                                       rank / max rank, bin = floor(. / (1/2 + 1e-5)) + 1 (two bins, disc_method "median"), zeros stay 0;
//                                       columns whose non-zeros show exactly two bins
// Arithmetic is Float64 like the reference (clrnorm converts to Matrix{Float64}); the continuous modes return Float32
// (convert_to_target_prec with prec = 32).  Layout: n samples x p variables, column-major, as Julia holds it; one thread
// per sample row and column chunk (adjacent lanes = adjacent samples: coalesced), chunk partials reduced in a fixed order.
// HBM-bound elementwise work: 4 B read per count and pass (3 passes) + 4 B written.
#include <cmath>
#include <vector>

#include "fw_internal.h"

namespace {

#define NORM_CHUNKS 64

// per column: min and max over the samples (variance > 0 <=> min != max)
__global__ __launch_bounds__(256) void norm_col_minmax_kernel(const int32_t *__restrict__ x, int n, int p, int32_t *__restrict__ cmin,
                                                              int32_t *__restrict__ cmax)
{
    __shared__ int32_t s_lo[256], s_hi[256];
    const int j = blockIdx.x;
    int32_t lo = 0x7fffffff, hi = (int32_t)0x80000000;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int32_t v = x[(size_t)j * n + i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            s_lo[threadIdx.x] = s_lo[threadIdx.x + o] < s_lo[threadIdx.x] ? s_lo[threadIdx.x + o] : s_lo[threadIdx.x];
            s_hi[threadIdx.x] = s_hi[threadIdx.x + o] > s_hi[threadIdx.x] ? s_hi[threadIdx.x + o] : s_hi[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cmin[j] = s_lo[0];
        cmax[j] = s_hi[0];
    }
}

// per (row, column chunk) over the kept columns: read sum, number of zeros, sum of log over the non-zeros, smallest non-zero
__global__ __launch_bounds__(256) void norm_row_stats_kernel(const int32_t *__restrict__ x, int n, const int32_t *__restrict__ cols, int pk,
                                                             double *__restrict__ rsum, int32_t *__restrict__ rzero,
                                                             double *__restrict__ rlog, int32_t *__restrict__ rmin)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = blockIdx.y;
    const int per = (pk + NORM_CHUNKS - 1) / NORM_CHUNKS;
    const int j0 = c * per, j1 = (j0 + per) < pk ? (j0 + per) : pk;
    double s = 0.0, sl = 0.0;
    int nz = 0, mn = 0x7fffffff;
    for (int q = j0; q < j1; ++q) {
        const int32_t v = x[(size_t)cols[q] * n + i];
        s += (double)v;
        if (v == 0) {
            ++nz;
        } else {
            sl += log((double)v);
            mn = v < mn ? v : mn;
        }
    }
    rsum[(size_t)c * n + i] = s;
    rzero[(size_t)c * n + i] = nz;
    rlog[(size_t)c * n + i] = sl;
    rmin[(size_t)c * n + i] = mn;
}

// out[r][q] for kept rows r and kept columns q (column-major n_out x p_out)
//   mode 0 (clr_adapt): log(x' / g_r), x' = x or the row's pseudo-count, g_r = exp(mean log x')
//   mode 1 (clr_nz):    x == 0 ? 0 : log(x / g_r), g_r = exp(mean log of the non-zeros)
__global__ __launch_bounds__(256) void norm_clr_out_kernel(const int32_t *__restrict__ x, int n, const int32_t *__restrict__ rows, int nk,
                                                           const int32_t *__restrict__ cols, int pk, const double *__restrict__ pseudo,
                                                           const double *__restrict__ gmean, int mode, float *__restrict__ out)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nk) return;
    for (int q = blockIdx.y; q < pk; q += gridDim.y) {  // (grid.y is capped at NORM_GRID_Y: tables with more kept columns loop)
        const int32_t v = x[(size_t)cols[q] * n + rows[r]];
        double o;
        if (mode == 0)
            o = log((v == 0 ? pseudo[r] : (double)v) / gmean[r]);
        else
            o = v == 0 ? 0.0 : log((double)v / gmean[r]);
        out[(size_t)q * nk + r] = (float)o;
    }
}

__global__ __launch_bounds__(256) void norm_binary_out_kernel(const int32_t *__restrict__ x, int n, const int32_t *__restrict__ rows, int nk,
                                                              const int32_t *__restrict__ cols, int pk, int32_t *__restrict__ out)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nk) return;
    for (int q = blockIdx.y; q < pk; q += gridDim.y) out[(size_t)q * nk + r] = x[(size_t)cols[q] * n + rows[r]] != 0 ? 1 : 0;
}

// per kept column over the kept rows: does it hold both a zero and a non-zero?
__global__ __launch_bounds__(256) void norm_col_levels_kernel(const int32_t *__restrict__ x, int n, const int32_t *__restrict__ rows, int nk,
                                                              const int32_t *__restrict__ cols, int pk, int32_t *__restrict__ two)
{
    __shared__ int s_z, s_nz;
    const int q = blockIdx.x;
    if (threadIdx.x == 0) s_z = s_nz = 0;
    __syncthreads();
    int z = 0, nzz = 0;
    for (int r = threadIdx.x; r < nk; r += 256) {
        const int32_t v = x[(size_t)cols[q] * n + rows[r]];
        z |= v == 0;
        nzz |= v != 0;
    }
    if (z) s_z = 1;
    if (nzz) s_nz = 1;
    __syncthreads();
    if (threadIdx.x == 0) two[q] = s_z && s_nz;
}

// binned_nz_clr, one workgroup per kept column: the clr_nz values log(x / g_row) of the column's non-zero entries (kept rows) go
// to LDS and are sorted there (bitonic, +inf padding); an entry's tied rank is then (#smaller) + (#equal + 1) / 2 from two binary
// searches, the largest rank m - (e_max - 1) / 2, and bin = floor((rank / max rank) / (1/2 + 1e-5)) + 1 exactly as discretize()
// computes it in Float64 (preprocessing.jl:238-265 with n_bins - 1 = 2 for the non-zeros, :267-291).  two[q] = the non-zeros show
// both bins.  The keys live in LDS, 8 bytes per padded entry, up to 16 384 kept rows; beyond (r04: preprocessing.jl:217-291 has no
// bound) in a slice of device memory per workgroup (gkeys, M doubles each: the same bitonic network, its passes through L2
// instead of LDS -- the stores of a pass are visible to the workgroup's other wavefronts behind the barrier, one L1 per CU), the
// workgroups striding over the columns so that the scratch stays at gridDim.x * M doubles.
#define NORM_BIN_MAX 16384
__global__ __launch_bounds__(1024) void norm_binned_kernel(const int32_t *__restrict__ x, int n, const int32_t *__restrict__ rows, int nk,
                                                           const int32_t *__restrict__ cols, int pk, const double *__restrict__ gmean,
                                                           int32_t *__restrict__ out, int32_t *__restrict__ two, int M, double *gkeys)
{
    extern __shared__ double s_key_lds[];
    __shared__ int s_cnt, s_b1, s_b2;
    const int tid = threadIdx.x;
    double *s_key = gkeys ? gkeys + (size_t)blockIdx.x * (size_t)M : s_key_lds;
    // keys in device memory: relaxed agent-scope accesses (they bypass the per-CU vector cache, whose lines another wavefront's
    // store does not refresh) and a device-scope fence in front of every barrier; in LDS plain accesses
    const bool glob = gkeys != nullptr;
    auto KLD = [&](int i) -> double {
        return glob ? __longlong_as_double(__hip_atomic_load((const long long *)&s_key[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : s_key[i];
    };
    auto KST = [&](int i, double v) {
        if (glob)
            __hip_atomic_store((long long *)&s_key[i], __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            s_key[i] = v;
    };
#define NB_SYNC()                 \
    {                             \
        if (glob) __threadfence(); \
        __syncthreads();          \
    }
  for (int q = blockIdx.x; q < pk; q += gridDim.x) {
    const int32_t *col = x + (size_t)cols[q] * n;
    __syncthreads();  // (the previous column's last readers of s_key / s_b1 / s_b2)
    if (tid == 0) s_cnt = s_b1 = s_b2 = 0;
    __syncthreads();
    for (int r = tid; r < nk; r += 1024) {
        const int32_t v = col[rows[r]];
        if (v != 0) KST(atomicAdd(&s_cnt, 1), log((double)v / gmean[r]));  // order does not matter: sorted below
    }
    NB_SYNC()
    const int m = s_cnt;
    for (int i = m + tid; i < M; i += 1024) KST(i, INFINITY);
    NB_SYNC()
    for (int k = 2; k <= M; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < M; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const double a = KLD(i), b = KLD(l);
                    const bool up = (i & k) == 0;
                    if (up ? (a > b) : (a < b)) {
                        KST(i, b);
                        KST(l, a);
                    }
                }
            }
            NB_SYNC()
        }
    double rmax = 1.0;
    if (m > 0) {
        const double top = KLD(m - 1);
        int lo = 0, hi = m;  // first index with key >= top
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (KLD(mid) < top)
                lo = mid + 1;
            else
                hi = mid;
        }
        rmax = (double)m - ((double)(m - lo) - 1.0) / 2.0;
    }
    const double step = (1.0 / 2.0) + 1e-5;
    int b1 = 0, b2 = 0;
    for (int r = tid; r < nk; r += 1024) {
        const int32_t v = col[rows[r]];
        int bin = 0;
        if (v != 0) {
            const double c = log((double)v / gmean[r]);  // the same operations as above: the same bits
            int lo = 0, hi = m;
            while (lo < hi) {  // #smaller
                const int mid = (lo + hi) >> 1;
                if (KLD(mid) < c)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            int lo2 = lo, hi2 = m;
            while (lo2 < hi2) {  // first index with key > c
                const int mid = (lo2 + hi2) >> 1;
                if (KLD(mid) <= c)
                    lo2 = mid + 1;
                else
                    hi2 = mid;
            }
            const double rank = (double)lo + ((double)(lo2 - lo) + 1.0) / 2.0;
            bin = (int)floor((rank / rmax) / step) + 1;
            b1 |= bin == 1;
            b2 |= bin == 2;
        }
        out[(size_t)q * nk + r] = bin;
    }
    if (b1) s_b1 = 1;
    if (b2) s_b2 = 1;
    __syncthreads();
    if (tid == 0) two[q] = s_b1 && s_b2;
  }
#undef NB_SYNC
}

// dst column q2 = src column sel[q2] (nk entries each)
__global__ __launch_bounds__(256) void norm_gather_cols_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ sel, int nk, int pk,
                                                               int32_t *__restrict__ dst)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nk) return;
    for (int q = blockIdx.y; q < pk; q += gridDim.y) dst[(size_t)q * nk + r] = src[(size_t)sel[q] * nk + r];
}

#define NORM_GRID_Y 65535  // HIP's limit on grid.y: the per-column output kernels loop beyond it

#define NHIP(call)                                                                                           \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess) {                                                                             \
            rc = fw_fail(nullptr, FW_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            goto done;                                                                                       \
        }                                                                                                    \
    } while (0)

}  // namespace

extern "C" int fw_normalize_counts(int32_t device, int32_t kind, int32_t n, int32_t p, const int32_t *counts, float *out_f32,
                                   int32_t *out_i32, uint8_t *row_mask, uint8_t *col_mask, int32_t *n_out, int32_t *p_out)
{
    if (!counts || !row_mask || !col_mask || !n_out || !p_out || n <= 0 || p <= 0)
        return fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: invalid argument");
    if (kind != FW_FZ && kind != FW_FZ_NZ && kind != FW_MI && kind != FW_MI_NZ)
        return fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: unknown kind %d", kind);
    const bool discrete = kind == FW_MI || kind == FW_MI_NZ;
    if (discrete ? !out_i32 : !out_f32) return fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: missing output buffer");
    int rc = FW_OK;
    int32_t *d_x = nullptr, *d_cmin = nullptr, *d_cmax = nullptr, *d_cols = nullptr, *d_rows = nullptr, *d_rzero = nullptr, *d_rmin = nullptr,
            *d_two = nullptr, *d_oi = nullptr, *d_tmp = nullptr, *d_sel = nullptr;
    double *d_rsum = nullptr, *d_rlog = nullptr, *d_pseudo = nullptr, *d_g = nullptr, *d_keys = nullptr;
    float *d_of = nullptr;
    std::vector<int32_t> cmin((size_t)p), cmax((size_t)p), cols, rows, rzero, rmin;
    std::vector<double> rsum, rlog;
    int pk = 0, nk = 0;
    {
        NHIP(hipSetDevice(device));
        NHIP(hipMalloc((void **)&d_x, sizeof(int32_t) * (size_t)n * p));
        NHIP(hipMemcpy(d_x, counts, sizeof(int32_t) * (size_t)n * p, hipMemcpyHostToDevice));
        NHIP(hipMalloc((void **)&d_cmin, sizeof(int32_t) * p));
        NHIP(hipMalloc((void **)&d_cmax, sizeof(int32_t) * p));
        hipLaunchKernelGGL(norm_col_minmax_kernel, dim3(p), dim3(256), 0, 0, d_x, n, p, d_cmin, d_cmax);
        NHIP(hipGetLastError());
        NHIP(hipMemcpy(cmin.data(), d_cmin, sizeof(int32_t) * p, hipMemcpyDeviceToHost));
        NHIP(hipMemcpy(cmax.data(), d_cmax, sizeof(int32_t) * p, hipMemcpyDeviceToHost));
        for (int j = 0; j < p; ++j) {
            if (cmin[j] < 0) {
                rc = fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: negative count in column %d", j);
                goto done;
            }
            col_mask[j] = cmin[j] != cmax[j];  // var(data, dims=1) .> 0
            if (col_mask[j]) cols.push_back(j);
        }
        pk = (int)cols.size();
        if (pk == 0) {
            rc = fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: every column is constant");
            goto done;
        }
        NHIP(hipMalloc((void **)&d_cols, sizeof(int32_t) * pk));
        NHIP(hipMemcpy(d_cols, cols.data(), sizeof(int32_t) * pk, hipMemcpyHostToDevice));
        const size_t cn = (size_t)NORM_CHUNKS * n;
        NHIP(hipMalloc((void **)&d_rsum, sizeof(double) * cn));
        NHIP(hipMalloc((void **)&d_rlog, sizeof(double) * cn));
        NHIP(hipMalloc((void **)&d_rzero, sizeof(int32_t) * cn));
        NHIP(hipMalloc((void **)&d_rmin, sizeof(int32_t) * cn));
        hipLaunchKernelGGL(norm_row_stats_kernel, dim3((n + 255) / 256, NORM_CHUNKS), dim3(256), 0, 0, d_x, n, d_cols, pk, d_rsum, d_rzero,
                           d_rlog, d_rmin);
        NHIP(hipGetLastError());
        rsum.resize(cn);
        rlog.resize(cn);
        rzero.resize(cn);
        rmin.resize(cn);
        NHIP(hipMemcpy(rsum.data(), d_rsum, sizeof(double) * cn, hipMemcpyDeviceToHost));
        NHIP(hipMemcpy(rlog.data(), d_rlog, sizeof(double) * cn, hipMemcpyDeviceToHost));
        NHIP(hipMemcpy(rzero.data(), d_rzero, sizeof(int32_t) * cn, hipMemcpyDeviceToHost));
        NHIP(hipMemcpy(rmin.data(), d_rmin, sizeof(int32_t) * cn, hipMemcpyDeviceToHost));
        // chunk partials -> per-row totals, in chunk order (deterministic)
        std::vector<double> S((size_t)n, 0.0), SL((size_t)n, 0.0);
        std::vector<int64_t> NZ((size_t)n, 0);
        int32_t min_abund = 0x7fffffff;
        for (int c = 0; c < NORM_CHUNKS; ++c)
            for (int i = 0; i < n; ++i) {
                S[i] += rsum[(size_t)c * n + i];
                SL[i] += rlog[(size_t)c * n + i];
                NZ[i] += rzero[(size_t)c * n + i];
            }
        for (int i = 0; i < n; ++i) {
            row_mask[i] = S[i] > 0.0;  // sum(data, dims=2) .> 0
            if (row_mask[i]) rows.push_back(i);
        }
        for (int c = 0; c < NORM_CHUNKS; ++c)
            for (int i : rows) min_abund = std::min(min_abund, rmin[(size_t)c * n + i]);
        nk = (int)rows.size();
        if (nk == 0) {
            rc = fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: no sample has reads");
            goto done;
        }
        std::vector<double> pseudo((size_t)nk, 0.0), g((size_t)nk, 1.0);
        if (kind == FW_FZ) {  // adaptive_pseudocount! (:157-190): row of maximal depth as the anchor
            int md = rows[0];
            for (int i : rows)
                if (S[i] > S[md]) md = i;
            const double base = min_abund >= 1 ? 1.0 : (double)min_abund / 10.0;
            const double k = (double)NZ[md], P = (double)pk, Nprod1 = SL[md];
            std::vector<int32_t> rows2;
            std::vector<double> ps2;
            for (int i : rows) {
                const double nz = (double)NZ[i];
                if (!(nz < P && k < P)) {
                    rc = fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: samples with all zero abundances are not allowed");
                    goto done;
                }
                const double ps = std::exp((1.0 / (nz - P)) * ((k - P) * std::log(base) + Nprod1 - SL[i]));
                if (ps != 0.0) {
                    rows2.push_back(i);
                    ps2.push_back(ps);
                } else {
                    row_mask[i] = 0;
                }
            }
            rows.swap(rows2);
            nk = (int)rows.size();
            pseudo.assign(ps2.begin(), ps2.end());
            g.resize((size_t)nk);
            for (int r = 0; r < nk; ++r) {  // clr!(pseudo_count = 0): geometric mean of the filled row
                const int i = rows[r];
                g[r] = std::exp((SL[i] + (double)NZ[i] * std::log(pseudo[r])) / P);
            }
        } else if (kind == FW_FZ_NZ || kind == FW_MI_NZ) {  // geometric mean of the non-zeros
            g.resize((size_t)nk);
            for (int r = 0; r < nk; ++r) {
                const int i = rows[r];
                const double cnt = (double)pk - (double)NZ[i];
                g[r] = cnt > 0 ? std::exp(SL[i] / cnt) : 1.0;
            }
        }
        NHIP(hipMalloc((void **)&d_rows, sizeof(int32_t) * nk));
        NHIP(hipMemcpy(d_rows, rows.data(), sizeof(int32_t) * nk, hipMemcpyHostToDevice));
        if (kind == FW_MI) {
            // presabs_norm! + columns with exactly two levels among the kept samples
            NHIP(hipMalloc((void **)&d_two, sizeof(int32_t) * pk));
            hipLaunchKernelGGL(norm_col_levels_kernel, dim3(pk), dim3(256), 0, 0, d_x, n, d_rows, nk, d_cols, pk, d_two);
            NHIP(hipGetLastError());
            std::vector<int32_t> two((size_t)pk);
            NHIP(hipMemcpy(two.data(), d_two, sizeof(int32_t) * pk, hipMemcpyDeviceToHost));
            std::vector<int32_t> cols2;
            for (int q = 0; q < pk; ++q) {
                if (two[q])
                    cols2.push_back(cols[q]);
                else
                    col_mask[cols[q]] = 0;
            }
            cols.swap(cols2);
            pk = (int)cols.size();
            if (pk == 0) {
                rc = fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: no column with two levels");
                goto done;
            }
            NHIP(hipMemcpy(d_cols, cols.data(), sizeof(int32_t) * pk, hipMemcpyHostToDevice));
            NHIP(hipMalloc((void **)&d_oi, sizeof(int32_t) * (size_t)nk * pk));
            hipLaunchKernelGGL(norm_binary_out_kernel, dim3((nk + 255) / 256, std::min(pk, NORM_GRID_Y)), dim3(256), 0, 0, d_x, n, d_rows, nk, d_cols, pk, d_oi);
            NHIP(hipGetLastError());
            NHIP(hipMemcpy(out_i32, d_oi, sizeof(int32_t) * (size_t)nk * pk, hipMemcpyDeviceToHost));
        } else if (kind == FW_MI_NZ) {
            int M = 2;
            while (M < nk) M <<= 1;
            const bool keys_in_lds = M <= NORM_BIN_MAX;  // beyond: one slice of device memory per workgroup
            const int bgrid = keys_in_lds ? pk : std::min(pk, 1024);
            if (!keys_in_lds) NHIP(hipMalloc((void **)&d_keys, sizeof(double) * (size_t)M * (size_t)bgrid));
            NHIP(hipMalloc((void **)&d_g, sizeof(double) * nk));
            NHIP(hipMemcpy(d_g, g.data(), sizeof(double) * nk, hipMemcpyHostToDevice));
            NHIP(hipMalloc((void **)&d_two, sizeof(int32_t) * pk));
            NHIP(hipMalloc((void **)&d_tmp, sizeof(int32_t) * (size_t)nk * pk));
            NHIP(hipFuncSetAttribute((const void *)norm_binned_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * NORM_BIN_MAX)));
            hipLaunchKernelGGL(norm_binned_kernel, dim3(bgrid), dim3(1024), keys_in_lds ? sizeof(double) * (size_t)M : 0, 0, d_x, n, d_rows, nk, d_cols, pk,
                               d_g, d_tmp, d_two, M, d_keys);
            NHIP(hipGetLastError());
            std::vector<int32_t> two((size_t)pk), sel;
            NHIP(hipMemcpy(two.data(), d_two, sizeof(int32_t) * pk, hipMemcpyDeviceToHost));
            for (int q = 0; q < pk; ++q) {
                if (two[q])
                    sel.push_back(q);
                else
                    col_mask[cols[q]] = 0;
            }
            pk = (int)sel.size();
            if (pk == 0) {
                rc = fw_fail(nullptr, FW_ERR_ARG, "fw_normalize_counts: no column whose non-zero abundances fall into two bins");
                goto done;
            }
            NHIP(hipMalloc((void **)&d_sel, sizeof(int32_t) * pk));
            NHIP(hipMemcpy(d_sel, sel.data(), sizeof(int32_t) * pk, hipMemcpyHostToDevice));
            NHIP(hipMalloc((void **)&d_oi, sizeof(int32_t) * (size_t)nk * pk));
            hipLaunchKernelGGL(norm_gather_cols_kernel, dim3((nk + 255) / 256, std::min(pk, NORM_GRID_Y)), dim3(256), 0, 0, d_tmp, d_sel, nk, pk, d_oi);
            NHIP(hipGetLastError());
            NHIP(hipMemcpy(out_i32, d_oi, sizeof(int32_t) * (size_t)nk * pk, hipMemcpyDeviceToHost));
        } else {
            NHIP(hipMalloc((void **)&d_pseudo, sizeof(double) * nk));
            NHIP(hipMalloc((void **)&d_g, sizeof(double) * nk));
            NHIP(hipMemcpy(d_pseudo, pseudo.data(), sizeof(double) * nk, hipMemcpyHostToDevice));
            NHIP(hipMemcpy(d_g, g.data(), sizeof(double) * nk, hipMemcpyHostToDevice));
            NHIP(hipMalloc((void **)&d_of, sizeof(float) * (size_t)nk * pk));
            hipLaunchKernelGGL(norm_clr_out_kernel, dim3((nk + 255) / 256, std::min(pk, NORM_GRID_Y)), dim3(256), 0, 0, d_x, n, d_rows, nk, d_cols, pk, d_pseudo, d_g,
                               kind == FW_FZ ? 0 : 1, d_of);
            NHIP(hipGetLastError());
            NHIP(hipMemcpy(out_f32, d_of, sizeof(float) * (size_t)nk * pk, hipMemcpyDeviceToHost));
        }
        *n_out = nk;
        *p_out = pk;
    }
done:
    void *ptrs[] = {d_x, d_cmin, d_cmax, d_cols, d_rows, d_rzero, d_rmin, d_two, d_oi, d_tmp, d_sel, d_rsum, d_rlog, d_pseudo, d_g, d_of, d_keys};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    return rc;
}
