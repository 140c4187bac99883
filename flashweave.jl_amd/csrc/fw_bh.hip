// Level-0 epilogue on the device: benjamini_hochberg! (statfuns.jl:326-350) on the p < alpha subset and the neighbour
// lists of condensed_stats_to_dict (tests.jl:372-388) as a CSR with ascending partners.  Input: the compacted
// significant pairs the level-0 kernels leave in device memory (i < j, statistic, raw p); output: ctx->nb_* on the host.
//   1. sort (p, index) descending by p (radix sort; p >= 0, no NaN in the subset)
//   2. adj_desc[t] = p * m / (k - t)  (= p * m / rank, rank ascending 1-based), first element clamped to 1
//   3. inclusive min-scan over the descending order = the reference's backward cumulative minimum
//   4. scatter back, keep adj < alpha, emit both directions keyed (src << 32 | dst), sort, cut rows by binary search
// Floating-point expressions are the host restatement's (p * m / rank, one multiplication then one division).
// The sorts and the scan are rocPRIM's device primitives called directly (ROCm's native library; r01-r03 went through the
// hipCUB compatibility layer): LSD radix sorts are stable, which step 4 and the candidate order rely on.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "fw_internal.h"

namespace {

__global__ void bh_iota_kernel(uint32_t *v, size_t k)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < k) v[t] = (uint32_t)t;
}

__global__ void bh_adj_kernel(const double *__restrict__ p_desc, double *__restrict__ adj, size_t k, double md)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    double a = p_desc[t] * md / (double)(k - t);
    if (t == 0) a = a < 1.0 ? a : 1.0;  // statfuns.jl:343: adj_last = min(p * m / n_filt, 1)
    adj[t] = a;
}

__global__ void bh_scatter_kernel(const double *__restrict__ scanned, const uint32_t *__restrict__ idx_desc,
                                  double *__restrict__ padj, size_t k)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < k) padj[idx_desc[t]] = scanned[t];
}

struct MinOp {
    __device__ __forceinline__ double operator()(const double &a, const double &b) const { return b < a ? b : a; }
};

// kept pairs -> two directed entries each; slots are handed out per wavefront
__global__ void bh_emit_kernel(const int32_t *__restrict__ pi, const int32_t *__restrict__ pj, const double *__restrict__ padj,
                               size_t k, double alpha, unsigned long long *__restrict__ counter,
                               unsigned long long *__restrict__ keys, uint32_t *__restrict__ pay)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool keep = t < k && padj[t] < alpha;
    const unsigned long long mask = __ballot(keep);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, 2ull * (unsigned long long)__popcll(mask));
    base = __shfl(base, leader);
    if (keep) {
        const unsigned long long slot = base + 2ull * (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        const unsigned long long i = (unsigned long long)(uint32_t)pi[t], j = (unsigned long long)(uint32_t)pj[t];
        keys[slot] = (i << 32) | j;
        pay[slot] = (uint32_t)t;
        keys[slot + 1] = (j << 32) | i;
        pay[slot + 1] = (uint32_t)t;
    }
}

__global__ void bh_rows_kernel(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ pay,
                               const double *__restrict__ stat64, const float *__restrict__ stat32,
                               const double *__restrict__ padj, size_t n2, int32_t *__restrict__ idx,
                               double *__restrict__ st, double *__restrict__ pv)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    const uint32_t t = pay[q];
    idx[q] = (int32_t)(keys[q] & 0xFFFFFFFFull);
    st[q] = stat64 ? stat64[t] : (double)stat32[t];
    pv[q] = padj[t];
}

__global__ void bh_offsets_kernel(const unsigned long long *__restrict__ keys, size_t n2, int p, long long *__restrict__ off,
                                  int *__restrict__ off32)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > p) return;
    const unsigned long long want = (unsigned long long)(uint32_t)v << 32;
    size_t lo = 0, hi = n2;  // first entry with key >= want
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (keys[mid] < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    off[v] = (long long)lo;
    off32[v] = (int)lo;
}

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

int fwi_bh_csr_device(fw_ctx *ctx, const FwL0Dev &in, int64_t m)
{
    const int p = ctx->P.p;
    const size_t k = in.k;
    ctx->d_nb_off = nullptr;
    ctx->d_nb_idx = nullptr;
    ctx->d_nb_stat = ctx->d_nb_p = nullptr;
    ctx->d_cand = nullptr;
    ctx->nb_host_valid = true;  // (empty lists until proven otherwise)
    ctx->nb_off.assign((size_t)p + 1, 0);
    ctx->nb_idx.clear();
    ctx->nb_stat.clear();
    ctx->nb_p.clear();
    if (k == 0) return FW_OK;
    if (k > 0xFFFFFFF0ull) return fw_fail(ctx, FW_ERR_LIMIT, "level-0: %zu significant pairs exceed the 32-bit index range", k);
    hipStream_t st = ctx->stream;
    // temp storage sizes of the three library calls
    size_t tb_sort1 = 0, tb_scan = 0, tb_sort2 = 0, tb_seg = 0;
    FW_HIP(ctx, (rocprim::radix_sort_pairs_desc(nullptr, tb_sort1, (const double *)nullptr, (double *)nullptr, (const uint32_t *)nullptr,
                                                (uint32_t *)nullptr, k, 0u, 64u, st)));
    FW_HIP(ctx, (rocprim::inclusive_scan(nullptr, tb_scan, (const double *)nullptr, (double *)nullptr, k, MinOp(), st)));
    FW_HIP(ctx, (rocprim::radix_sort_pairs(nullptr, tb_sort2, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                           (const uint32_t *)nullptr, (uint32_t *)nullptr, 2 * k, 0u, 64u, st)));
    FW_HIP(ctx, (rocprim::segmented_radix_sort_pairs(nullptr, tb_seg, (const double *)nullptr, (double *)nullptr, (const int32_t *)nullptr,
                                                     (int32_t *)nullptr, (unsigned int)(2 * k), (unsigned int)p, (const int *)nullptr,
                                                     (const int *)nullptr, 0u, 64u, st)));
    const size_t tb = std::max(std::max(tb_sort1, tb_seg), std::max(tb_scan, tb_sort2));
    // carve one scratch buffer
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += up256(bytes);
        return o;
    };
    const size_t o_tmp = take(tb), o_pdesc = take(k * 8), o_idesc = take(k * 4), o_iota = take(k * 4), o_adj = take(k * 8),
                 o_padj = take(k * 8), o_cnt = take(8), o_keys = take(2 * k * 8), o_pay = take(2 * k * 4),
                 o_keys2 = take(2 * k * 8), o_pay2 = take(2 * k * 4), o_idx = take(2 * k * 4), o_st = take(2 * k * 8),
                 o_pv = take(2 * k * 8), o_off = take(((size_t)p + 1) * 8), o_off32 = take(((size_t)p + 1) * 4),
                 o_cand = take(2 * k * 4), o_psort = take(2 * k * 8);
    int rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_bh, off))) return rc;
    char *B = (char *)ctx->d_bh.ptr;
    double *p_desc = (double *)(B + o_pdesc), *adj = (double *)(B + o_adj), *padj = (double *)(B + o_padj);
    uint32_t *i_desc = (uint32_t *)(B + o_idesc), *iota = (uint32_t *)(B + o_iota);
    unsigned long long *cnt = (unsigned long long *)(B + o_cnt), *keys = (unsigned long long *)(B + o_keys),
                       *keys2 = (unsigned long long *)(B + o_keys2);
    uint32_t *pay = (uint32_t *)(B + o_pay), *pay2 = (uint32_t *)(B + o_pay2);
    int32_t *idx = (int32_t *)(B + o_idx);
    double *sto = (double *)(B + o_st), *pvo = (double *)(B + o_pv);
    long long *offs = (long long *)(B + o_off);
    const unsigned gk = (unsigned)((k + 255) / 256);
    if (ctx->P.fdr) {
        hipLaunchKernelGGL(bh_iota_kernel, dim3(gk), dim3(256), 0, st, iota, k);
        size_t t1 = tb;
        FW_HIP(ctx, (rocprim::radix_sort_pairs_desc(B + o_tmp, t1, in.pval, p_desc, (const uint32_t *)iota, i_desc, k, 0u, 64u, st)));
        hipLaunchKernelGGL(bh_adj_kernel, dim3(gk), dim3(256), 0, st, (const double *)p_desc, adj, k, (double)m);
        size_t t2 = tb;
        FW_HIP(ctx, (rocprim::inclusive_scan(B + o_tmp, t2, (const double *)adj, adj, k, MinOp(), st)));
        hipLaunchKernelGGL(bh_scatter_kernel, dim3(gk), dim3(256), 0, st, (const double *)adj, (const uint32_t *)i_desc, padj, k);
    } else {
        FW_HIP(ctx, hipMemcpyAsync(padj, in.pval, k * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    FW_HIP(ctx, hipMemsetAsync(cnt, 0, 8, st));
    hipLaunchKernelGGL(bh_emit_kernel, dim3(gk), dim3(256), 0, st, in.i, in.j, (const double *)padj, k, ctx->P.alpha, cnt, keys, pay);
    FW_HIP(ctx, hipGetLastError());
    unsigned long long n2 = 0;
    FW_HIP(ctx, hipMemcpyAsync(&n2, cnt, 8, hipMemcpyDeviceToHost, st));
    FW_HIP(ctx, hipStreamSynchronize(st));
    ctx->cnt.kernel_launches += 6;
    if (n2 == 0) return FW_OK;
    int bits = 1;
    while ((1ll << bits) < (long long)p) ++bits;
    size_t t3 = tb;
    FW_HIP(ctx, (rocprim::radix_sort_pairs(B + o_tmp, t3, (const unsigned long long *)keys, keys2, (const uint32_t *)pay, pay2, (size_t)n2, 0u,
                                           (unsigned int)(32 + bits), st)));
    hipLaunchKernelGGL(bh_rows_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, (const unsigned long long *)keys2,
                       (const uint32_t *)pay2, in.stat64, in.stat32, (const double *)padj, (size_t)n2, idx, sto, pvo);
    int *off32 = (int *)(B + o_off32);
    int32_t *cand = (int32_t *)(B + o_cand);
    double *psort = (double *)(B + o_psort);
    hipLaunchKernelGGL(bh_offsets_kernel, dim3((unsigned)((p + 1 + 255) / 256)), dim3(256), 0, st,
                       (const unsigned long long *)keys2, (size_t)n2, p, offs, off32);
    // candidate order of every variable (hiton.jl:211-217): its neighbours by ascending adjusted p, ties by ascending
    // index = a STABLE sort of each row by p (radix sort is stable; rows are already in ascending partner order)
    size_t t4 = tb;
    FW_HIP(ctx, (rocprim::segmented_radix_sort_pairs(B + o_tmp, t4, (const double *)pvo, psort, (const int32_t *)idx, cand, (unsigned int)n2,
                                                     (unsigned int)p, (const int *)off32, (const int *)off32 + 1, 0u, 64u, st)));
    FW_HIP(ctx, hipGetLastError());
    // only the row offsets go to the host now; partners / statistics / p-values follow on demand (fwi_nb_host_ensure):
    // the device-resident rounds never need them there (58 MB at cfg3)
    FW_HIP(ctx, hipMemcpyAsync(ctx->nb_off.data(), offs, ((size_t)p + 1) * 8, hipMemcpyDeviceToHost, st));
    FW_HIP(ctx, hipStreamSynchronize(st));
    ctx->cnt.kernel_launches += 4;
    ctx->nb_host_valid = false;
    ctx->d_cand = cand;
    ctx->d_nb_off = offs;  // stay valid until the next level-0 (the device HITON rounds read them)
    ctx->d_nb_idx = idx;
    ctx->d_nb_stat = sto;
    ctx->d_nb_p = pvo;
    return FW_OK;
}

// Host copies of the neighbour lists (partners, statistics, adjusted p) on demand.
int fwi_nb_host_ensure(fw_ctx *ctx)
{
    if (ctx->nb_host_valid) return FW_OK;
    const size_t n2 = (size_t)ctx->nb_off[ctx->P.p];
    ctx->nb_idx.resize(n2);
    ctx->nb_stat.resize(n2);
    ctx->nb_p.resize(n2);
    if (n2) {
        if (!ctx->d_nb_idx) return fw_fail(ctx, FW_ERR_STATE, "level-0 neighbour lists are neither on the host nor on the device");
        FW_HIP(ctx, hipMemcpy(ctx->nb_idx.data(), ctx->d_nb_idx, n2 * 4, hipMemcpyDeviceToHost));
        FW_HIP(ctx, hipMemcpy(ctx->nb_stat.data(), ctx->d_nb_stat, n2 * 8, hipMemcpyDeviceToHost));
        FW_HIP(ctx, hipMemcpy(ctx->nb_p.data(), ctx->d_nb_p, n2 * 8, hipMemcpyDeviceToHost));
    }
    ctx->nb_host_valid = true;
    return FW_OK;
}
