// Device-resident exchange of level-0 results between the ranks of a target-sharded run (fw_level0_sharded_dev): the significant
// pairs a rank's share of the pair tiles produced are packed into 24-byte records straight inside a send buffer the CALLER owns
// (torch tensors in bench.py: RCCL all-gathers them over xGMI without a host copy), and the gathered, rank-padded buffer is
// compacted back into the structure-of-arrays form the BH / neighbour-list epilogue (fw_bh.hip) reads.  r02 sent the same data
// through pinned host memory, numpy and a C callback: 8.8 s per cfg4 pass over gloo, which made level-0 sharding unusable.
#include "fw_internal.h"

namespace {

struct FwL0Rec {  // wire format: 24 bytes
    int32_t i, j;
    double stat, pval;
};
static_assert(sizeof(FwL0Rec) == 24, "level-0 exchange record");

__global__ __launch_bounds__(256) void l0_pack_kernel(const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                      const double *__restrict__ s, const double *__restrict__ p, long long k,
                                                      FwL0Rec *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= k) return;
    FwL0Rec r;
    r.i = i[t];
    r.j = j[t];
    r.stat = s[t];
    r.pval = p[t];
    out[t] = r;
}

// recv: world blocks of cap records, block r holds counts[r] of them; off[r] = exclusive prefix sum of counts
__global__ __launch_bounds__(256) void l0_unpack_kernel(const FwL0Rec *__restrict__ recv, long long cap, int world,
                                                        const long long *__restrict__ off, long long total, int32_t *__restrict__ i,
                                                        int32_t *__restrict__ j, double *__restrict__ s, double *__restrict__ p)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    int r = 0;
    while (r + 1 < world && off[r + 1] <= t) ++r;  // world <= a few dozen
    const FwL0Rec v = recv[(long long)r * cap + (t - off[r])];
    i[t] = v.i;
    j[t] = v.j;
    s[t] = v.stat;
    p[t] = v.pval;
}

}  // namespace

int fwi_l0_exchange_dev(fw_ctx *c, const fw_dev_exchange *x, int world, const FwL0Dev &local, int64_t m_local, FwL0Dev *merged,
                        int64_t *m_sum)
{
    std::vector<int64_t> counts((size_t)world, 0), aux((size_t)world, 0);
    void *d_send = nullptr, *d_recv = nullptr;
    int64_t cap = 0;
    int rc = x->prepare(x->user, (int64_t)local.k, m_local, (int32_t)sizeof(FwL0Rec), &d_send, &d_recv, counts.data(), aux.data(), &cap);
    if (rc) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded_dev: exchange.prepare failed (%d)", rc);
    if ((int64_t)local.k > cap || (!d_send && local.k) || !d_recv) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded_dev: exchange.prepare returned no room (cap %lld for %zu records)", (long long)cap, local.k);
    if (local.k) {
        hipLaunchKernelGGL(l0_pack_kernel, dim3((unsigned)((local.k + 255) / 256)), dim3(256), 0, c->stream, local.i, local.j, local.stat64,
                           local.pval, (long long)local.k, (FwL0Rec *)d_send);
        FW_HIP(c, hipGetLastError());
    }
    FW_HIP(c, hipStreamSynchronize(c->stream));  // the caller's collective runs on its own stream
    rc = x->exchange(x->user);
    if (rc) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded_dev: exchange.exchange failed (%d)", rc);
    std::vector<long long> off((size_t)world + 1, 0);
    int64_t msum = 0;
    for (int r = 0; r < world; ++r) {
        if (counts[r] < 0 || counts[r] > cap) return fw_fail(c, FW_ERR_ARG, "fw_level0_sharded_dev: rank %d reports %lld records (cap %lld)", r, (long long)counts[r], (long long)cap);
        off[r + 1] = off[r] + counts[r];
        msum += aux[r];
    }
    const size_t total = (size_t)off[world];
    if ((rc = fw_dev_reserve(c, c->d_l0m_i, (total + 1) * 2 * sizeof(int32_t)))) return rc;
    if ((rc = fw_dev_reserve(c, c->d_l0m_d, (total + 1) * 2 * sizeof(double) + ((size_t)world + 1) * sizeof(long long)))) return rc;
    int32_t *oi = (int32_t *)c->d_l0m_i.ptr, *oj = oi + total;
    double *os = (double *)c->d_l0m_d.ptr, *op = os + total;
    long long *d_off = (long long *)(op + total + 1);
    FW_HIP(c, hipMemcpyAsync(d_off, off.data(), sizeof(long long) * ((size_t)world + 1), hipMemcpyHostToDevice, c->stream));
    if (total) {
        hipLaunchKernelGGL(l0_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, (const FwL0Rec *)d_recv, (long long)cap,
                           world, (const long long *)d_off, (long long)total, oi, oj, os, op);
        FW_HIP(c, hipGetLastError());
    }
    FW_HIP(c, hipStreamSynchronize(c->stream));
    *merged = FwL0Dev{};
    merged->i = oi;
    merged->j = oj;
    merged->stat64 = os;
    merged->pval = op;
    merged->k = total;
    *m_sum = msum;
    return FW_OK;
}
