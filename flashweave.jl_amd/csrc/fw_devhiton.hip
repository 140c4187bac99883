// Device-resident HITON-PC rounds (FW_FZ, FW_MI, FW_MI_NZ).  The host driver of fw_hiton.cpp pays one host round trip per pool round
// (collect -> merge -> advance -> build -> launch, ~140 us at cfg3, a third of the pass); here the per-target state
// machines, the in-rank-order merge of the segment results and the construction of the next launch live in device
// memory and run as three small kernels between two launches of the segment kernel:
//     seg kernel -> dh_step_kernel (merge, commit, advance: one wavefront per target)
//                -> dh_plan_kernel (segment length, per-target segment counts, exclusive scan: one workgroup)
//                -> dh_fill_kernel (one thread per segment record) -> seg kernel -> ...
// (plus dh_compact_kernel once per batch: the three kernels walk a list of the targets that still have work).
// The host only enqueues batches of rounds and looks at a pinned "done" flag between batches.  Semantics are those
// of the host driver (hiton.jl:109-149 interleaving / elimination, check_candidate! :80-107, update_PC_dict! :249-256,
// tests.jl:326-345 merge rules); windows follow the same growth policy.  Speculation here is the look-ahead described
// at dh_step_kernel (elimination phase: next members against the pools they will see if every earlier one is kept;
// interleaving phase: first windows of the next candidates), switched on only while launches are small.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include "fw_internal.h"
#define MI_MAX_K FW_MAX_K_FAST  // (conditioning sets of 6 and 7 variables: host job pool over the segment kernels of fw_mi.hip)
#include "fw_mi_core.h"
#include "fw_fz_core.h"
#include "fw_unrank.h"

namespace {

struct DhTgt {
    int32_t T, phase, pos, nc;  // nc = candidates of the current phase
    int32_t na, ntpc, npc, wl_n;
    long long co;  // offset of this target's arrays (capacity cap each; accepted list: 2 * co, 2 * cap)
    long long cand_off;  // offset of its interleaving candidates in DhArrays::cand0
    long long wl_off, nb_off;
    int32_t nb_n, cap;
    int32_t jactive, nsp;  // nsp: look-ahead jobs of the elimination phase in the coming launch (see dh_step_kernel)
    int32_t cur, pad0;     // accepted-list buffer in use (0 .. spec_depth)
    unsigned long long jN, jnext, jwidth, jwin, jevaluated;
    unsigned long long c_eval_short;  // executed tests of jobs with at most FW_HK_A accepted variables (FW_TRACE_HOST)
    int na_max, wl_used;              // longest accepted list of any job of this target; whitelisted candidates appended in phase 0
    double jbest_p, jbest_stat;
    unsigned long long c_ref, c_calls, c_eval;  // per-target totals (summed on the host: no same-address atomics)
    double c_alg;
    unsigned int r_first0, r_more0, r_first1, r_more1;  // FW_TRACE_HOST: rounds this target spent on first / later windows of interleaving (0) and elimination (1) jobs
    // look-ahead jobs of the coming launch: spmode 0 = same accepted list / elimination pools, window = the job's own (jwin2 = jwin);
    // spmode 1 = interleaving "assume the current candidate is accepted": list = accepted + [candidate], own first window jwin2 of
    // an enumeration of jN2 subsets (see dh_step_kernel)
    int32_t spmode, wl_unsorted;  // wl_unsorted: the whitelist was appended to on the device (dh_wl_append_kernel): no order, no early exit
    unsigned long long jwin2, jN2;
    long long tm_off;  // fz rounds (r06): offset (floats) of the target's local correlation matrix in DhArrays::tmat, -1: none
};

struct DhGlobal {
    unsigned long long launched_ranks, next_ranks;
    unsigned int n_live_prev, n_live_next;
    unsigned int ns, seglen, done, rounds, rounds_nonempty, max_a;  // max_a: longest accepted list of any job so far
    unsigned long long cond_tests_ref, subsets_calls, evaluated;
    double alg_bytes;
    unsigned int n_act, act_sel;  // unfinished targets: act[act_sel * ntg + 0 .. n_act) (dh_compact_kernel)
    unsigned int pad_b, max_ab;     // max_ab: largest (accepted + whitelisted neighbours still to come) of any target so far: what a
                                    // list can reach without tested acceptances
    unsigned int ns_ring[64];  // segments of the last 64 planned launches (the host reads the record once per batch)
    // words that many workgroups touch: a cache line each
    unsigned int pad_l0[32];
    unsigned int step_ticket;  // (unused: the r05 cooperative round's barrier word)
    unsigned int pad_l1[31];
    unsigned int any_big;      // the coming launch holds a segment with |accepted| > FW_TAB_A (set by the fill): the in-lane variant
                               // of the fz segment kernel leaves at once when it does not
    unsigned int pad_l2[31];
};

// accepted-list buffer b of a target: (spec_depth + 1) buffers of 2 * cap entries each
#define DH_ACC_OFF(x, b, d1) (2ll * (x).co * (long long)(d1) + 2ll * (long long)(b) * (long long)(x).cap)

struct DhArrays {
    const int32_t *cand0;  // interleaving candidates (hiton.jl:211-217 order)
    int32_t *tpc_key, *pc_key, *acc;
    double *tpc_stat, *tpc_p, *pc_stat, *pc_p;
    const int32_t *wl;        // sorted whitelists (feed-forward)
    const unsigned int *wl_cnt;  // device-built whitelists (fwi_devhiton_mi_schedule): entries of variable v's list so far, else null
    const long long *nb_off;  // level-0 neighbour lists (for the empty-pool case, hiton.jl:57-59)
    const int32_t *nb_idx;
    const double *nb_stat, *nb_p;
    const float *tmat;  // fz rounds (r06): the targets' local correlation matrices (DhTgt::tm_off), else null
};

struct DhParams {
    double alpha;
    int max_k;
    long long max_tests;
    unsigned long long small_launch, w0_big, w0_small;  // window policy (fw_core.cpp: fwi_pool_add / fw_window_growth)
    unsigned int seg_q, seg_min;                        // segment length granularity / minimum (fz 256 / 256, discrete 4 / 8)
    double disc_bytes_per_col;                          // discrete kinds: n * b / 8 (algorithmic bytes per column), else 0
    int elim_full;                                      // elimination-phase jobs start with the full enumeration as their window
    unsigned long long growth_small, growth, growth_busy;  // window growth: launch below small_launch / default / many jobs
    unsigned int busy_jobs;
    int spec_depth;  // elimination-phase look-ahead: candidates tested ahead of the current one per target (0 = off)
    unsigned long long spec_below;  // ... only while the last launch held fewer ranks than this
    unsigned int mi_seq, mi_win0, mi_chunk_div, mi_chunk_min, mi_chunk_max, mi_help_jobs;  // dh_mi_target_kernel (env knobs)
    unsigned int mi_elim_min;  // elimination-phase jobs with more ranks than this open a board for the whole enumeration at once (0: off)
    unsigned int mi_heavy;     // targets with at least this many candidates never work on other targets' boards (0: off)
    unsigned int mi_seq_heavy; // ... and run this many first tests of a job alone (instead of mi_seq) before they open a board
    unsigned int mi_seq_tail;  // ... and every target this many once the target list is exhausted (idle wavefronts are waiting for work)
    unsigned int mi_chunk_tail, mi_win0_tail;  // ranks per record / first window once the launch is in its tail (idle wavefronts waiting)
    unsigned int mi_ahead;     // R4 kernels: first tests of up to four interleaving candidates in one step (mi_first4); 0: off
    unsigned int mi_team;      // the first mi_team targets of the (heaviest-first) list are run by a whole workgroup each (dh_mi_team)
    unsigned int mi_team_tail;   // team targets publish tail-mode boards (short records, wide first window) from the start
    unsigned int mi_team_steps;  // lock-step rounds of a team job (4 wavefronts x 1 or 4 ranks each) before its enumeration goes to a board
    unsigned int mi_trace;       // take the per-job / per-target clock reads (FW_TRACE_HOST, FW_MI_TICKS builds)
    int spec0_depth;                // interleaving-phase look-ahead (first windows of the next candidates)
    unsigned long long spec0_below;
    unsigned int spec0_jobs;  // ... and fewer live jobs than this
    int spec0_depth_light;          // ... and this many candidates while the last launch held fewer than spec0_light_below ranks
    unsigned long long spec0_light_below;
    int spec1_depth;          // interleaving-phase look-ahead behind a candidate that is about to be accepted (same two conditions)
};

// Is v one of the target's whitelisted neighbours?  Called by the whole wavefront with a uniform v: the lanes read 64 entries of
// the (sorted) list at a time and vote -- ONE load latency per 64 entries.  (r02: a binary search, i.e. log2(n) + 1 DEPENDENT global
// loads per call, up to four calls per target and round: a third of dh_step_kernel's 19 us.)
__device__ __forceinline__ bool dh_in_wl(const DhTgt &x, const DhArrays &A, int32_t v)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int32_t *w = A.wl + x.wl_off;
    for (int base = 0; base < x.wl_n; base += 64) {
        const int32_t e = base + lane < x.wl_n ? w[base + lane] : -1;
        if (__ballot(e == v) != 0ull) return true;
        if (!x.wl_unsorted && __shfl(e, 63) > v) return false;  // sorted: nothing further on can match (an invalid last lane reads -1: the loop ends anyway)
    }
    return false;
}

// algorithmic bytes of the first `evaluated` ranks of a job over `a` accepted variables (fwi_alg_bytes, fz form)
__device__ __forceinline__ double dh_alg_bytes(int a, unsigned long long evaluated, int max_k, double disc_bytes_per_col)
{
    double bytes = 0.0, left = (double)evaluated;
    for (int s = max_k; s >= 1 && left > 0.0; --s) {
        double b = 1.0;
        for (int i = 1; i <= s; ++i) b = b * (double)(a - s + i) / (double)i;
        if (a < s) b = 0.0;
        const double cnt = left < b ? left : b;
        bytes += cnt * (disc_bytes_per_col > 0.0 ? (double)(s + 2) * disc_bytes_per_col + 32.0
                                                 : 4.0 * (double)((s + 2) * (s + 1) / 2) + 32.0);
        left -= cnt;
    }
    return bytes;
}

// number of subsets of sizes max_k .. 1 of a accepted variables, capped by max_tests (tests.jl:300-311): 32-bit binomials (no
// 64-bit division, no Float64 estimate) where every term fits -- all of cfg2 / cfg4; the persistent discrete kernel paid for
// three fw_binom_u64 per job
__device__ __forceinline__ unsigned long long dh_enum_size(int a, int max_k, long long max_tests)
{
    unsigned long long N = 0ull;
    if (max_k <= 3 && a <= FW_UNRANK32_A) {
        for (int s = max_k; s >= 1; --s) N += (unsigned long long)fw_binom32(a, s);
    } else {
        for (int s = max_k; s >= 1; --s) {
            N += fw_binom_u64(a, s);
            if (N > (1ull << 62)) N = 1ull << 62;
        }
    }
    if (max_tests > 0 && (unsigned long long)max_tests < N) N = (unsigned long long)max_tests;
    return N;
}

// dh_alg_bytes for the discrete kinds with max_k <= 3 and short lists: the same sum in integers (bytes per test of size s are
// (s + 2) * bytes_per_col + 32, a multiple of 1/8 at most -> exact in Float64 either way)
__device__ __forceinline__ double dh_alg_bytes_disc32(int a, unsigned long long evaluated, int max_k, double disc_bytes_per_col)
{
    double bytes = 0.0;
    unsigned long long left = evaluated;
    for (int s = max_k; s >= 1 && left > 0ull; --s) {
        const unsigned long long b = (unsigned long long)fw_binom32(a, s), cnt = left < b ? left : b;
        bytes += (double)cnt * ((double)(s + 2) * disc_bytes_per_col + 32.0);
        left -= cnt;
    }
    return bytes;
}

// Advance a target until it needs a device test (returns true; the job is (T, cands[pos], acc[0..na))) or finishes.
// Executed by the whole wavefront of the target: every lane holds the same copy of x and takes the same branches
// (stores of one lane are made visible to the others by workgroup-scope fences: one L1 per CU, no cache maintenance);
// the loops over the target's arrays (removing the candidate from the pool, the phase switch, update_PC_dict!) are
// spread over the lanes -- a single lane walking 250 dependent global loads per candidate was 100 us per round.
__device__ bool dh_advance(DhTgt &x, const DhArrays &A, int lane, int d1)
{
    for (;;) {
        int32_t *acc = A.acc + DH_ACC_OFF(x, x.cur, d1);
        if (x.phase == 2) return false;
        const int32_t *cands = x.phase == 0 ? A.cand0 + x.cand_off : A.tpc_key + x.co;
        int32_t *dkey = (x.phase == 0 ? A.tpc_key : A.pc_key) + x.co;
        double *dstat = (x.phase == 0 ? A.tpc_stat : A.pc_stat) + x.co;
        double *dp = (x.phase == 0 ? A.tpc_p : A.pc_p) + x.co;
        int32_t &dn = x.phase == 0 ? x.ntpc : x.npc;
        while (x.pos < x.nc) {
            const int32_t cand = cands[x.pos];
            if (x.wl_n > 0 && dh_in_wl(x, A, cand)) {  // hiton.jl:20-30
                if (lane == 0) {
                    acc[x.na] = cand;
                    dkey[dn] = cand;
                    dstat[dn] = NAN;
                    dp[dn] = NAN;
                }
                ++x.na;
                x.wl_used += x.phase == 0 ? 1 : 0;
                ++dn;
                ++x.pos;
                continue;
            }
            if (x.phase == 1) {  // hiton.jl:134-136: the candidate leaves the conditioning pool while it is tested
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                int w = 0;  // stable in-place compaction, 64 entries at a time (writes never pass the read front)
                for (int base = 0; base < x.na; base += 64) {
                    const int q = base + lane;
                    const int32_t v = q < x.na ? acc[q] : cand;
                    const bool keep = q < x.na && v != cand;
                    const unsigned long long m = __ballot(keep);
                    if (keep) acc[w + __popcll(m & ((1ull << lane) - 1ull))] = v;
                    w += __popcll(m);
                }
                x.na = w;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            if (x.na == 0) {  // tests.jl:285 sentinel + hiton.jl:57-59
                double s = NAN, p = NAN;
                if (x.phase == 0) {
                    const int32_t *b = A.nb_idx + x.nb_off;
                    int lo = 0, hi = x.nb_n;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (b[mid] < cand)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    s = A.nb_stat[x.nb_off + lo];
                    p = A.nb_p[x.nb_off + lo];
                } else {
                    for (int q = 0; q < x.ntpc; ++q)
                        if (A.tpc_key[x.co + q] == cand) {
                            s = A.tpc_stat[x.co + q];
                            p = A.tpc_p[x.co + q];
                            break;
                        }
                }
                if (lane == 0) {
                    acc[x.na] = cand;
                    dkey[dn] = cand;
                    dstat[dn] = s;
                    dp[dn] = p;
                }
                ++x.na;
                ++dn;
                ++x.pos;
                continue;
            }
            return true;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (x.phase == 0) {  // hiton.jl:242: elimination over keys(TPC) in insertion order
            x.phase = 1;
            x.nc = x.ntpc;
            for (int q = lane; q < x.ntpc; q += 64) acc[q] = A.tpc_key[x.co + q];
            x.na = x.ntpc;
            x.pos = 0;
        } else {  // hiton.jl:249-256 update_PC_dict!
            for (int i = lane; i < x.npc; i += 64) {
                const int32_t k = A.pc_key[x.co + i];
                for (int q = 0; q < x.ntpc; ++q)
                    if (A.tpc_key[x.co + q] == k) {
                        const double tp = A.tpc_p[x.co + q], pp = A.pc_p[x.co + i];
                        if (tp > pp || isnan(pp)) {
                            A.pc_stat[x.co + i] = A.tpc_stat[x.co + q];
                            A.pc_p[x.co + i] = tp;
                        }
                        break;
                    }
            }
            x.phase = 2;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// ceil(w / seglen) for w < 2^62: double quotient + exact fix-up (a 64-bit integer division costs ~100 instructions)
__device__ __forceinline__ unsigned int dh_ceil_div(unsigned long long w, unsigned long long seglen, double inv)
{
    if (w == 0ull) return 0u;
    unsigned long long q = (unsigned long long)((double)w * inv);
    while (q * seglen < w) ++q;
    while (q > 0ull && (q - 1ull) * seglen >= w) --q;
    return (unsigned int)q;
}


// In-rank-order merge of the nseg segment records of one job by the 64 lanes of a wavefront (tests.jl:326-345): the
// first stop (smallest segment index) ends the job; otherwise the lexicographic maximum of (p, segment index), i.e.
// "later wins ties" (tests.jl:338).  Every lane returns the same values.
#define DH_MAX_SPEC 8  // look-ahead jobs per target (register arrays in dh_step_kernel)

struct DhMerge {
    bool stop;
    double stat, p;  // of the stopping test (stop) or of the maximum-p test (!stop; p = -2 if no segment had one)
    int pow;
    unsigned long long nt;  // stop: tests up to and including the stopping one
    unsigned long long ev;  // tests executed by the segments
    double g;               // mi_merge, !stop: G^2 and df of the maximum-p test (the seed of later records of the job)
    int df;
};

__device__ __forceinline__ DhMerge dh_merge(const FwSegOut *__restrict__ so, long long base, int nseg, int lane)
{
    unsigned long long ev = 0ull;
    int my_stop = 0x7fffffff;  // smallest segment index of this lane that reports a stop
    double st_stat = 0.0, st_p = 0.0;
    int st_pow = 0, st_df = 0;
    unsigned long long st_rank = 0ull;
    double bp = -2.0, bs = 0.0;  // lane best over its segments (increasing index, `>=`)
    int bi = -1;
    for (int sg = lane; sg < nseg; sg += 64) {
        const FwSegOut o = so[base + sg];
        ev += o.evaluated;
        if (o.stop_rank != FW_RANK_NONE) {
            if (my_stop == 0x7fffffff) {
                my_stop = sg;
                st_stat = o.stop_stat;
                st_p = o.stop_pval;
                st_pow = o.stop_power;
                st_rank = o.stop_rank;
                st_df = o.stop_df;
            }
        } else if (o.best_pval >= bp) {
            bp = o.best_pval;
            bs = o.best_stat;
            bi = sg;
        }
    }
    int first = my_stop;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ev += __shfl_xor(ev, o);
        const int f2 = __shfl_xor(first, o);
        first = f2 < first ? f2 : first;
        const double p2 = __shfl_xor(bp, o), s2 = __shfl_xor(bs, o);
        const int i2 = __shfl_xor(bi, o);
        if (p2 > bp || (p2 == bp && i2 > bi)) {
            bp = p2;
            bs = s2;
            bi = i2;
        }
    }
    DhMerge M;
    M.ev = ev;
    if (first != 0x7fffffff) {  // segments after the first stop were speculative
        const int owner = first & 63;
        M.stop = true;
        M.stat = __shfl(st_stat, owner);
        M.p = __shfl(st_p, owner);
        M.pow = __shfl(st_pow, owner);
        M.nt = __shfl(st_rank, owner) + 1ull;
        // fz_nz: too few rows with T != 0 and candidate != 0 -> (0, 1, 0, false) with ZERO tests (tests.jl:294-296; marker of fz_seg_body)
        if (__shfl(st_df, owner) == -2) M.nt = 0ull;
    } else {
        M.stop = false;
        M.stat = bs;
        M.p = bi >= 0 ? bp : -2.0;
        M.pow = 1;
        M.nt = 0ull;
    }
    M.g = 0.0;
    M.df = 0;
    return M;
}

// issig (tests.jl:1-3) -> hiton.jl:61-63: the candidate at x.pos joins the accepted list (buffer x.cur) and TPC / PC,
// or is dropped.  Every lane computes the same state; lane 0 writes.
__device__ __forceinline__ bool dh_commit(DhTgt &x, const DhArrays &A, int lane, int d1, double r_stat, double r_p, int r_pow,
                                          double alpha)
{
    const int32_t *cands = x.phase == 0 ? A.cand0 + x.cand_off : A.tpc_key + x.co;
    const int32_t cand = cands[x.pos];
    ++x.pos;
    if (!(r_p < alpha && r_pow)) return false;
    if (lane == 0) {
        A.acc[DH_ACC_OFF(x, x.cur, d1) + x.na] = cand;
        if (x.phase == 0) {
            A.tpc_key[x.co + x.ntpc] = cand;
            A.tpc_stat[x.co + x.ntpc] = r_stat;
            A.tpc_p[x.co + x.ntpc] = r_p;
        } else {
            A.pc_key[x.co + x.npc] = cand;
            A.pc_stat[x.co + x.npc] = r_stat;
            A.pc_p[x.co + x.npc] = r_p;
        }
    }
    ++x.na;
    if (x.phase == 0)
        ++x.ntpc;
    else
        ++x.npc;
    return true;
}

// ---- discrete kinds: persistent wavefronts, one target at a time, big enumerations shared through a board ---------------
// A discrete (T, candidate) job is a handful of tests (cfg4: 3.3 on average, 786 000 jobs per pass) and a test is a few
// thousand instructions (fw_mi_core.h), so the level-synchronous rounds above (segment kernel -> step -> plan -> fill,
// ~550 dependent rounds per chain at cfg4, a 16-rank speculative window per job) cost more than the tests themselves.
// Here the launch holds as many wavefronts as the GPU keeps resident; each takes the next target of a heaviest-first list
// (one atomic) and runs its whole HITON-PC: dh_advance / dh_commit are the state machine of the rounds (hiton.jl:109-149,
// :53-78, :249-256), test_subsets (tests.jl:281-346) runs sequentially in the reference's own order, so a job that stops
// after a few tests -- nearly all of them -- costs exactly those tests: no window, no speculation, no round trip.
// A job that survives its first mi_seq tests is an enumeration that will probably run to the end (cfg4: one target owns
// a chain of 90 000 tests).  Its owner publishes the following ranks window by window on a BOARD in device memory: chunk
// records any wavefront can claim with one atomic.  Wavefronts look at the boards before every job of their own and when
// they run out of targets, so a big enumeration gets the whole GPU while its owner waits; the owner claims chunks of its
// own board too, which is why nothing ever waits on a wavefront that is not running.  Chunk results are merged in rank
// order exactly like segment records (dh_merge): first stop wins, otherwise the (p, rank) maximum with "later wins
// ties"; a stop found by one chunk cancels the later chunks of the board (stop_min).  num_tests is the reference's count.
// (the thresholds live in DhParams: mi_seq = 48 tests of a job its owner runs alone before it opens a board, mi_win0 = 128 ranks in the
// first board window, later windows grow x8 -- set where the launch parameters are built, with the sweeps that chose them)
#define MI_BOARD_CAP (1u << 18)
#define MI_REC_CAP (1u << 20)

// one LDS table [stratum][cell] per wavefront (fw_mi_core.h); module scope so that the called test routine addresses it as LDS
__shared__ unsigned short dh_mi_tab[4][2 * MI_TAB16];  // (twice the 16-bit table: the 32-bit form of more than 65 535 samples)

__shared__ int32_t dh_mi_acc[4][1024];  // a helper's copy of the accepted list of the board it works on (MI_ACC_LDS)
__shared__ DhTgt dh_mi_x[4];            // the state of the target each wavefront of dh_mi_target_kernel is working on
__shared__ MiBest dh_mi_seed[4];        // the seed a wavefront hands to the test routine (mi_run_ranks)
// Everything the out-of-line routines of the persistent kernel need about the launch -- device arrays, data description, parameters --
// sits in LDS, written once by the kernel, and the test routine leaves its record in its wavefront's LDS slot.  r02 / r03 passed
// MiDev by value and DhArrays / DhParams by reference and returned FwSegOut by value: by-value structs cross a call through
// private memory (100 + 72 bytes PER LANE written and read back per call, 1 M calls per cfg4 pass), and a by-reference kernel
// argument forces the kernel to keep a stack copy that every use in the callee loads from -- the r03 counters showed twice as many
// bytes written as read by this kernel, and the r04 tick profile 9 us per call outside the tests (profiles/r04_cfg4_attribution.json).
struct DhMiCtx {
    DhArrays A;
    MiDev M;
    DhParams P;
};
__shared__ DhMiCtx dh_mi_ctx;
__shared__ FwSegOut dh_mi_out[4];  // the record of the last mi_run_ranks / mi_run_ranks4 call of each wavefront
#ifdef FW_MI_TICKS
__shared__ unsigned long long dh_mi_ticks[4][12];  // per wavefront: calls, tests, ticks in the prologue / the test core / the accounting (Q) / sizes; [6..11]: ticks in dh_advance, job set-up (enumeration size, tail look), commit + counters, publish, wait for own board (idle), merge
#endif

// Everything two wavefronts share travels as write-through messages: the producer stores with sc1 (relaxed agent-scope atomic
// stores: the line leaves its XCD's L2), drains them with `s_waitcnt vmcnt(0)` (inline asm: the compiler drops the builtin
// form after a release fence on ROCm 7.2, MI355X_MICROARCH.md "compiler hazard") and then raises the flag / counter; the
// consumer polls with ONE relaxed sc1 load and reads the payload with sc1 loads.  No acquire / release fences: on this
// multi-XCD part an agent acquire invalidates the reader's L2 -- i.e. the bit planes every test reads -- and the first
// version of this kernel, which fenced per record, ran 15x slower with 2-rank records than with 8-rank ones.
struct MiBoard {
    unsigned long long tc;        // T | cand << 32
    unsigned long long ac;        // a | chunk << 32            (chunk = ranks per record)
    unsigned long long acc_off;   // accepted list of the job: write-through copy in MiShared::bacc
    unsigned long long start, end;  // ranks [start, end) of the window
    unsigned long long nr;        // nch | res_off << 32        (records of the window: res[res_off .. res_off + nch))
    unsigned long long stop_min;  // smallest stopping rank found so far (FW_RANK_NONE: none)
    unsigned int next_chunk, done;  // claimed / finished records
    unsigned int ready, seed_df;
    double seed_p, seed_g;        // the job's maximum-p test so far (seed_p < 0: none): records skip the Q(a, x) of tests it dominates
};

struct alignas(16) MiQueue {
    unsigned int next_target, targets_done, n_boards, hint, res_top, bacc_top, pad[1], next_team;  // pad[0]: watchdog code
    unsigned long long t_body, t_ctl, t_sleep, n_seg;  // dh_mi_target_kernel: 100 MHz ticks summed over wavefronts (FW_TRACE_HOST; see the end of the kernel)
    unsigned long long t_total;  // dh_mi_target_kernel: ticks until the wavefront ran out of targets (the fields above: see its end)
    unsigned long long tick[12];  // FW_MI_TICKS builds: calls of the test routine, tests, ticks before / in the test core / in the accounting, sum of set sizes; [6..11] see dh_mi_ticks
    unsigned long long tm_run, tm_wait, tm_steps, tm_tests;  // dh_mi_team: ticks inside the test routine / at the barrier behind it, lock-step rounds, tests in them (all wavefronts)
};

#define MI_BACC_CAP (1u << 22)  // ints of accepted-list copies per launch
#define MI_ACC_LDS 1024         // accepted-list entries a helper stages in LDS (longer lists are read through sc1 loads)

__device__ __forceinline__ unsigned int mi_ld_u32(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long mi_ld_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mi_st_u64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mi_st_u32(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mi_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the four hot words of MiQueue (next_target, targets_done, n_boards, hint: its first 16 bytes) in ONE sc1 round trip -- the owner
// of a target looks at them before every fourth job, and four dependent relaxed loads were four round trips of ~2 us
typedef unsigned int mi_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mi_u32x4 mi_ld_u128(const void *p)
{
    mi_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
#ifndef MI_NAP_WAIT
#define MI_NAP_WAIT 2  // x 128 cycles between two looks at a board the wavefront waits for
#endif
#ifndef MI_NAP_TAIL
#define MI_NAP_TAIL 32  // (1 / 1: cfg4 159.8 ms, 2 / 32: 156.2 -- and the idle wavefronts stop inflating the instruction counters) x 512 cycles between two looks at the boards once the wavefront has run out of targets
#endif
__device__ __forceinline__ void mi_nap_wait()
{
#pragma unroll
    for (int i = 0; i < MI_NAP_WAIT; ++i) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void mi_nap_tail()
{
#pragma unroll
    for (int i = 0; i < MI_NAP_TAIL; ++i) __builtin_amdgcn_s_sleep(8);
}
// Watchdog of the waits of the persistent kernel: a wavefront that waits for another one (records of its board, the end of the launch)
// gives up when the word it watches has not CHANGED for MI_WD_TICKS of the 100 MHz clock (2 s: a record is at most mi_chunk_max = 64
// tests) -- the launch then ends with a code in MiQueue::pad[0] and the host call fails with FW_ERR_DEVICE instead of hanging the GPU.
// The clock (a scalar memory round trip) is read every 256th look only.
#ifndef MI_WD_TICKS
#define MI_WD_TICKS 200000000ull
#endif
struct MiWatch {
    unsigned long long t0;
    unsigned int last, n;
};
__device__ __forceinline__ void mi_watch_begin(MiWatch &w, unsigned int word)
{
    w.t0 = 0ull;
    w.last = word;
    w.n = 0u;
}
__device__ __forceinline__ bool mi_watch_expired(MiWatch &w, unsigned int word)
{
    if (word != w.last) {
        w.last = word;
        w.t0 = 0ull;
        w.n = 0u;
        return false;
    }
    if ((++w.n & 255u) != 0u) return false;
    const unsigned long long now = wall_clock64();
    if (w.t0 == 0ull) {
        w.t0 = now;
        return false;
    }
    return now - w.t0 > MI_WD_TICKS;
}
// lane 0 performs the atomic, every lane gets the value
__device__ __forceinline__ unsigned int mi_wave_add(unsigned int *p, unsigned int v, int lane)
{
    unsigned int r = 0u;
    if (lane == 0) r = atomicAdd(p, v);
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)r);
}

// 64-bit v_readlane
__device__ __forceinline__ unsigned long long mi_rfl_lane64(unsigned long long v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// ranks [r0, r1) of the job (T, cand | subsets of acc[0..a)) in enumeration order, one test after the other (whole wavefront).
// Called, not inlined: the kernel reaches it from four places (own jobs, own board, other boards while waiting / between
// jobs / at the end) and four copies of the test made 168 KB of code -- against a 64 KB instruction cache.
template <int L, int NXY, int PRE>
__device__ __noinline__ void mi_run_ranks(int T, int cand, const int32_t *__restrict__ acc_in, int a, int max_k,
                                          long long max_tests, unsigned long long r0, unsigned long long r1,
                                          const unsigned long long *stop_min_in, int remote_acc, const MiBest *seed_in)
{
    unsigned short *tab = dh_mi_tab[threadIdx.x >> 6];
    const MiDev M = mi_uniform(dh_mi_ctx.M);
    T = __builtin_amdgcn_readfirstlane(T);
    cand = __builtin_amdgcn_readfirstlane(cand);
    a = __builtin_amdgcn_readfirstlane(a);
    max_k = __builtin_amdgcn_readfirstlane(max_k);
    max_tests = (long long)mi_rfl64((unsigned long long)max_tests);
    r0 = mi_rfl64(r0);
    r1 = mi_rfl64(r1);
    const int32_t *acc = (const int32_t *)mi_rfl64((unsigned long long)acc_in);
    const unsigned long long *stop_min = (const unsigned long long *)mi_rfl64((unsigned long long)stop_min_in);
    remote_acc = __builtin_amdgcn_readfirstlane(remote_acc);
    FwSegOut o;
    o.stop_rank = FW_RANK_NONE;
    o.stop_stat = o.stop_pval = 0.0;
    o.best_rank = 0ull;
    o.best_stat = 0.0;
    o.best_pval = -3.0;  // "no test": dh_merge ignores it
    o.stop_df = o.stop_power = o.best_df = o.pad = 0;
    o.evaluated = 0ull;
    int s = max_k;
    // the positions of the running subset: only ever indexed with compile-time constants (unrolled loops), so they live in scalar
    // registers.  r05 indexed them with run-time values (the unranking loop, the step to the next subset): a private array on the
    // stack -- five dword stores per call and two or three DEPENDENT scratch loads per test in the stepping code, and, with the
    // callee-saved registers this out-of-line routine has to park, the 3.65 GB the kernel wrote per cfg4 pass for 86 MB of results
    // (profiles/r06_discrete_kernel_writes.txt)
    int pos[MI_MAX_K];
#pragma unroll
    for (int q = 0; q < MI_MAX_K; ++q) pos[q] = 0;
    if (max_k <= 3 && a <= FW_UNRANK32_A) {
        // every rank of the job fits 28 bits: 32-bit binomials, divisions by constants (fw_unrank.h: ~100 instructions instead of
        // ~1 500 -- the r03 trace of cfg4 showed 9.3 us of prologue per call of this routine, as much as a test)
        uint32_t rem = (uint32_t)r0;
        while (s > 1 && rem >= fw_binom32(a, s)) {
            rem -= fw_binom32(a, s);
            --s;
        }
        int prev = -1;  // fw_unrank_comb32 with static position indices (s <= 3 here)
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < s) {
                const int t = s - d, n = a - 1 - prev;
                const uint32_t tot = fw_binom32(n, t);
                const int m = fw_inv_binom32(tot - rem, t, n);
                rem -= tot - fw_binom32(m, t);
                pos[d] = a - m;
                prev = a - m;
            }
    } else {
        unsigned long long rem = r0;
        while (s > 1 && rem >= fw_binom_u64(a, s)) {
            rem -= fw_binom_u64(a, s);
            --s;
        }
        int pm[MI_MAX_K];  // (the general form keeps its array: lists beyond 1 024 entries or max_k 4-5, never at the benchmark sizes)
#pragma unroll
        for (int q = 0; q < MI_MAX_K; ++q) pm[q] = 0;
        fw_unrank_comb(rem, a, s, pm);
#pragma unroll
        for (int q = 0; q < MI_MAX_K; ++q) pos[q] = pm[q];
    }
    // the maximum-p record the job holds so far (seed: what earlier ranks of the job found -- LDS, every lane reads the same words).
    // A test the seed dominates (mi_account: df <= and G^2 >) skips its Q(a, x); without a seed every record pays one Q for its
    // first significant test -- 18-25 us at cfg4, three tests' worth.  A record reports a maximum only if it improved on the seed.
    const MiBest *seed = (const MiBest *)mi_rfl64((unsigned long long)seed_in);
    MiBest mb;
    mb.p = -3.0;
    mb.stat = mb.g = 0.0;
    mb.df = 0;
    if (seed) {
        mb.p = seed->p;
        mb.stat = seed->stat;
        mb.g = seed->g;
        mb.df = seed->df;
    }
#ifdef FW_MI_TICKS
    unsigned long long *tkw = dh_mi_ticks[threadIdx.x >> 6];
    unsigned long long tka = wall_clock64();
    if ((threadIdx.x & 63) == 0) tkw[0] += 1ull;
#endif
    for (unsigned long long r = r0; r < r1; ++r) {
        if (stop_min && mi_ld_u64(stop_min) < r) break;  // an earlier rank already ended the job
        MiZs zs;
#pragma unroll
        for (int q = 0; q < MI_MAX_K; ++q)  // remote_acc: another wavefront's list, read where it was written through (sc1)
            zs.v[q] = (q < s) ? (remote_acc ? __hip_atomic_load(&acc[pos[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : acc[pos[q]]) : 0;
#ifdef FW_MI_TICKS
        const unsigned long long tkb = wall_clock64();
#endif
        MiRes t = mi_test_core<L, NXY, PRE == 1, PRE == 2>(M, T, cand, zs, s, tab);  // PRE: 0 words loaded per batch, 1 register-resident (n <= 6144), 2 32-bit counts (n > 65 535)
        ++o.evaluated;
#ifdef FW_MI_TICKS
        const unsigned long long tkc = wall_clock64();
#endif
        const int ev = mi_account(M, t, max_tests > 0 && r + 1ull >= (unsigned long long)max_tests, mb, false);  // tests.jl:326-341
#ifdef FW_MI_TICKS
        {
            const unsigned long long tkd = wall_clock64();
            if ((threadIdx.x & 63) == 0) {
                tkw[1] += 1ull;
                tkw[2] += tkb - tka;
                tkw[3] += tkc - tkb;
                tkw[4] += tkd - tkc;
                tkw[5] += (unsigned long long)s;
            }
            tka = tkd;
        }
#endif
        if (ev == 1) {
            o.stop_rank = r;
            o.stop_stat = t.stat;
            o.stop_pval = t.pval;
            o.stop_df = t.df;
            o.stop_power = t.power;
            break;
        }
        if (ev == 2) {
            o.best_pval = mb.p;
            o.best_stat = mb.stat;
            o.best_rank = r;
            o.best_df = mb.df;
            o.stop_stat = mb.g;  // (records without a stop: G^2 of the maximum-p test, the seed of later records)
        }
        // next subset of this size in lexicographic order of the positions, then the next size down (static indices: see above)
        int i = -1, base = 0;  // i: the last position that can still move; base: its new value
#pragma unroll
        for (int q = 0; q < MI_MAX_K; ++q)
            if (q < s && pos[q] != a - s + q) {
                i = q;
                base = pos[q] + 1;
            }
        if (i < 0) {
            --s;
            if (s < 1) break;
#pragma unroll
            for (int q = 0; q < MI_MAX_K; ++q) pos[q] = q;
        } else {
#pragma unroll
            for (int q = 0; q < MI_MAX_K; ++q)
                if (q >= i && q < s) pos[q] = base + (q - i);
        }
    }
    if ((threadIdx.x & 63) == 0) dh_mi_out[threadIdx.x >> 6] = o;  // (every lane holds the same record)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The same enumeration, four subsets per step (mi_test_core4: one DPP row of 16 lanes per test; n <= MI4_N, max_k <= 3, 2 x 2 cells
// per stratum): the four tests of a step are ranks r .. r + 3 of the same size; they are accounted for in rank order exactly as the
// sequential loop would (first stop wins, `>=` maximum before it), so the tests behind a stop inside a step are the only
// speculation (counted in `evaluated`).
template <int L>
__device__ __noinline__ void mi_run_ranks4(int T, int cand, const int32_t *__restrict__ acc_in, int a, int max_k,
                                           long long max_tests, unsigned long long r0, unsigned long long r1,
                                           const unsigned long long *stop_min_in, int remote_acc, const MiBest *seed_in)
{
    unsigned short *tab = dh_mi_tab[threadIdx.x >> 6];
    const MiDev M = mi_uniform(dh_mi_ctx.M);
    T = __builtin_amdgcn_readfirstlane(T);
    cand = __builtin_amdgcn_readfirstlane(cand);
    a = __builtin_amdgcn_readfirstlane(a);
    max_k = __builtin_amdgcn_readfirstlane(max_k);
    max_tests = (long long)mi_rfl64((unsigned long long)max_tests);
    r0 = mi_rfl64(r0);
    r1 = mi_rfl64(r1);
    const int32_t *acc = (const int32_t *)mi_rfl64((unsigned long long)acc_in);
    const unsigned long long *stop_min = (const unsigned long long *)mi_rfl64((unsigned long long)stop_min_in);
    remote_acc = __builtin_amdgcn_readfirstlane(remote_acc);
    const int row = (threadIdx.x & 63) >> 4;
    FwSegOut o;
    o.stop_rank = FW_RANK_NONE;
    o.stop_stat = o.stop_pval = 0.0;
    o.best_rank = 0ull;
    o.best_stat = 0.0;
    o.best_pval = -3.0;
    o.stop_df = o.stop_power = o.best_df = o.pad = 0;
    o.evaluated = 0ull;
    int s = max_k;
    int p0 = 0, p1 = 0, p2 = 0;  // positions of the running subset (s <= 3), scalar
    {
        int pos[MI_MAX_K];
#pragma unroll
        for (int q = 0; q < MI_MAX_K; ++q) pos[q] = 0;
        if (a <= FW_UNRANK32_A) {  // (max_k <= 3 here: 32-bit binomials, see mi_run_ranks)
            uint32_t rem = (uint32_t)r0;
            while (s > 1 && rem >= fw_binom32(a, s)) {
                rem -= fw_binom32(a, s);
                --s;
            }
            fw_unrank_comb32(rem, a, s, pos);
        } else {
            unsigned long long rem = r0;
            while (s > 1 && rem >= fw_binom_u64(a, s)) {
                rem -= fw_binom_u64(a, s);
                --s;
            }
            fw_unrank_comb(rem, a, s, pos);
        }
        p0 = pos[0];
        p1 = pos[1];
        p2 = pos[2];
    }
    // the maximum-p record the job holds so far (seed: what earlier ranks of the job found -- LDS, every lane reads the same words).
    // A test the seed dominates (mi_account: df <= and G^2 >) skips its Q(a, x); without a seed every record pays one Q for its
    // first significant test -- 18-25 us at cfg4, three tests' worth.  A record reports a maximum only if it improved on the seed.
    const MiBest *seed = (const MiBest *)mi_rfl64((unsigned long long)seed_in);
    MiBest mb;
    mb.p = -3.0;
    mb.stat = mb.g = 0.0;
    mb.df = 0;
    if (seed) {
        mb.p = seed->p;
        mb.stat = seed->stat;
        mb.g = seed->g;
        mb.df = seed->df;
    }
    unsigned long long r = r0;
    while (r < r1 && s >= 1) {
        if (stop_min && mi_ld_u64(stop_min) < r) break;  // an earlier rank already ended the job
        // up to four subsets of size s: row t takes the t-th; rows beyond the step's count repeat the first (ignored)
        int zrow[3] = {0, 0, 0};
        int nb = 0;
        const int s_step = s;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool have = r + (unsigned long long)t < r1 && s == s_step;  // uniform
            if (have) {
                ++nb;
                const int q0 = p0, q1 = s_step >= 2 ? p1 : p0, q2 = s_step >= 3 ? p2 : p0;
                const int v0 = remote_acc ? __hip_atomic_load(&acc[q0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : acc[q0];
                const int v1 = remote_acc ? __hip_atomic_load(&acc[q1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : acc[q1];
                const int v2 = remote_acc ? __hip_atomic_load(&acc[q2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : acc[q2];
                if (row == t || t == 0) {  // t == 0 first: the default of every row
                    zrow[0] = v0;
                    zrow[1] = v1;
                    zrow[2] = v2;
                }
                // next subset in lexicographic order of the positions, then the next size down (scalar)
                if (s_step == 1) {
                    if (p0 < a - 1) ++p0; else s = 0;
                } else if (s_step == 2) {
                    if (p1 < a - 1) {
                        ++p1;
                    } else if (p0 < a - 2) {
                        ++p0;
                        p1 = p0 + 1;
                    } else {
                        s = 1;
                        p0 = 0;
                    }
                } else {
                    if (p2 < a - 1) {
                        ++p2;
                    } else if (p1 < a - 2) {
                        ++p1;
                        p2 = p1 + 1;
                    } else if (p0 < a - 3) {
                        ++p0;
                        p1 = p0 + 1;
                        p2 = p0 + 2;
                    } else {
                        s = 2;
                        p0 = 0;
                        p1 = 1;
                    }
                }
            }
        }
        MiRes mine = mi_test_core4<L>(M, T, cand, zrow, s_step, tab, cand);
        o.evaluated += (unsigned long long)nb;
        bool stopped = false;
        for (int t = 0; t < nb; ++t) {  // rank order (uniform: every lane reads row t's result)
            MiRes tt;
            const int src = 16 * t;
            tt.stat = __longlong_as_double((long long)mi_rfl_lane64((unsigned long long)__double_as_longlong(mine.stat), src));
            tt.g = __longlong_as_double((long long)mi_rfl_lane64((unsigned long long)__double_as_longlong(mine.g), src));
            tt.pval = __longlong_as_double((long long)mi_rfl_lane64((unsigned long long)__double_as_longlong(mine.pval), src));
            tt.df = __builtin_amdgcn_readlane(mine.df, src);
            tt.power = __builtin_amdgcn_readlane(mine.power, src);
            tt.n_obs = (long long)mi_rfl_lane64((unsigned long long)mine.n_obs, src);
            const unsigned long long rr = r + (unsigned long long)t;
            const int ev = mi_account(M, tt, max_tests > 0 && rr + 1ull >= (unsigned long long)max_tests, mb, false);  // tests.jl:326-341
            if (ev == 1) {
                o.stop_rank = rr;
                o.stop_stat = tt.stat;
                o.stop_pval = tt.pval;
                o.stop_df = tt.df;
                o.stop_power = tt.power;
                stopped = true;
                break;
            }
            if (ev == 2) {
                o.best_pval = mb.p;
                o.best_stat = mb.stat;
                o.best_rank = rr;
                o.best_df = mb.df;
                o.stop_stat = mb.g;  // (records without a stop: G^2 of the maximum-p test, the seed of later records)
            }
        }
        if (stopped) break;
        r += (unsigned long long)nb;
    }
    if ((threadIdx.x & 63) == 0) dh_mi_out[threadIdx.x >> 6] = o;  // (every lane holds the same record)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- first tests of four candidates at once (R4 kernels) ------------------------------------------------------------------------
// Most interleaving jobs end with their FIRST test (cfg2: 2.6 tests per job, a target with 190 candidates is a chain of ~200 jobs of
// ~20 us).  While a candidate is rejected the accepted list does not change, so the first test of the next candidates -- (T, c | the
// first subset of the same list) -- can be evaluated before the current one is decided: one mi_test_core4 step takes candidates
// pos .. pos + 3 on its four rows (same X, same Z, the row's own Y).  The results are kept per wavefront and consumed in
// candidate order as long as (target, accepted length) still match; the first accepted candidate invalidates the rest.  A
// candidate whose first test is significant continues with rank 1 of its own job, seeded with that test.
struct MiAhead {
    int T, pos0, na, n;  // results for candidates pos0 .. pos0 + n - 1 of target T's interleaving list, accepted length na
    int stop[4], df[4], power[4];
    double stat[4], pval[4], g[4];  // stop: the test ended the job (stat / pval / df / power of it); else its p is materialised
};
__shared__ MiAhead dh_mi_ahead[4];

// candidates cands[0 .. nb) that can share ONE mi_test_core4 step with the first of them: the uniform decisions of the test core read
// levels / maxv > 1 of Y, so only candidates that agree with cands[0] on both ride along (uniform: every lane computes the same count)
__device__ __forceinline__ int mi_first_run(const MiDev &M, const int32_t *__restrict__ cands, int nb)
{
    const int c0 = cands[0];
    int n = 1;
    for (int t = 1; t < nb; ++t) {
        const int ct = cands[t];
        if (M.levels[ct] != M.levels[c0] || (M.maxv[ct] > 1) != (M.maxv[c0] > 1)) break;
        ++n;
    }
    return n;
}

template <int L>
__device__ __noinline__ void mi_first4(int T, int pos0, const int32_t *__restrict__ cands_in, int nb, const int32_t *__restrict__ acc_in,
                                       int a, int max_k, long long max_tests)
{
    unsigned short *tab = dh_mi_tab[threadIdx.x >> 6];
    MiAhead &H = dh_mi_ahead[threadIdx.x >> 6];
    const MiDev M = mi_uniform(dh_mi_ctx.M);
    T = __builtin_amdgcn_readfirstlane(T);
    pos0 = __builtin_amdgcn_readfirstlane(pos0);
    nb = __builtin_amdgcn_readfirstlane(nb);
    a = __builtin_amdgcn_readfirstlane(a);
    max_k = __builtin_amdgcn_readfirstlane(max_k);
    max_tests = (long long)mi_rfl64((unsigned long long)max_tests);
    const int32_t *cands = (const int32_t *)mi_rfl64((unsigned long long)cands_in);
    const int32_t *acc = (const int32_t *)mi_rfl64((unsigned long long)acc_in);
    const int lane = threadIdx.x & 63, row = lane >> 4;
    const int s = max_k < a ? max_k : a;  // rank 0: the first subset of the largest size the list allows (tests.jl:300-311)
    int zrow[3];
    zrow[0] = acc[0];
    zrow[1] = acc[s >= 2 ? 1 : 0];
    zrow[2] = acc[s >= 3 ? 2 : 0];
    const int c0 = cands[0];
    const int n = mi_first_run(M, cands, nb);
    const int my_c = cands[row < n ? row : 0];
    MiRes mine = mi_test_core4<L>(M, T, c0, zrow, s, tab, my_c);
    for (int t = 0; t < n; ++t) {
        MiRes tt;
        const int src = 16 * t;
        tt.stat = __longlong_as_double((long long)mi_rfl_lane64((unsigned long long)__double_as_longlong(mine.stat), src));
        tt.g = __longlong_as_double((long long)mi_rfl_lane64((unsigned long long)__double_as_longlong(mine.g), src));
        tt.pval = __longlong_as_double((long long)mi_rfl_lane64((unsigned long long)__double_as_longlong(mine.pval), src));
        tt.df = __builtin_amdgcn_readlane(mine.df, src);
        tt.power = __builtin_amdgcn_readlane(mine.power, src);
        tt.n_obs = (long long)mi_rfl_lane64((unsigned long long)mine.n_obs, src);
        MiBest mb;
        mb.p = -3.0;
        mb.stat = mb.g = 0.0;
        mb.df = 0;
        const int ev = mi_account(M, tt, max_tests == 1, mb, false);  // tests.jl:326-341 for rank 0
        if (lane == 0) {
            H.stop[t] = ev == 1;
            H.stat[t] = tt.stat;
            H.pval[t] = ev == 1 ? tt.pval : mb.p;
            H.g[t] = tt.g;
            H.df[t] = tt.df;
            H.power[t] = tt.power;
        }
    }
    if (lane == 0) {
        H.T = T;
        H.pos0 = pos0;
        H.na = a;
        H.n = n;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// R4: the four-subsets-per-step form (host: n <= MI4_N, max_k <= 3, 2 x 2 cells); one of the two routines per instantiation
template <int L, int NXY, int PRE, bool R4>
__device__ __forceinline__ FwSegOut mi_run_ranks_sel(int T, int cand, const int32_t *__restrict__ acc, int a, int max_k,
                                                     long long max_tests, unsigned long long r0, unsigned long long r1,
                                                     const unsigned long long *stop_min, int remote_acc, const MiBest *seed)
{
    if constexpr (R4)
        mi_run_ranks4<L>(T, cand, acc, a, max_k, max_tests, r0, r1, stop_min, remote_acc, seed);
    else
        mi_run_ranks<L, NXY, PRE>(T, cand, acc, a, max_k, max_tests, r0, r1, stop_min, remote_acc, seed);
    return dh_mi_out[threadIdx.x >> 6];  // (inlined: the record comes back through LDS, not through a by-value return)
}

// one record = 9 64-bit words (FwSegOut), written through / read back word by word
__device__ __forceinline__ void mi_record_store(FwSegOut *dst, const FwSegOut &o)
{
    unsigned long long w[9];
    __builtin_memcpy(w, &o, sizeof(w));
    unsigned long long *d = (unsigned long long *)dst;
#pragma unroll
    for (int q = 0; q < 9; ++q) mi_st_u64(d + q, w[q]);
}
__device__ __forceinline__ FwSegOut mi_record_load(const FwSegOut *src)
{
    unsigned long long w[9];
    const unsigned long long *d = (const unsigned long long *)src;
#pragma unroll
    for (int q = 0; q < 9; ++q) w[q] = mi_ld_u64(d + q);
    FwSegOut o;
    __builtin_memcpy(&o, w, sizeof(w));
    return o;
}
static_assert(sizeof(FwSegOut) == 72, "FwSegOut is nine 64-bit words");

// dh_merge over records another wavefront wrote through (sc1 loads)
__device__ __forceinline__ DhMerge mi_merge(const FwSegOut *__restrict__ so, long long base, int nseg, int lane)
{
    unsigned long long ev = 0ull;
    int my_stop = 0x7fffffff;
    double st_stat = 0.0, st_p = 0.0;
    int st_pow = 0;
    unsigned long long st_rank = 0ull;
    double bp = -2.0, bs = 0.0, bg = 0.0;
    int bi = -1, bdf = 0;
    for (int sg = lane; sg < nseg; sg += 64) {
        const FwSegOut o = mi_record_load(so + base + sg);
        ev += o.evaluated;
        if (o.stop_rank != FW_RANK_NONE) {
            if (my_stop == 0x7fffffff) {
                my_stop = sg;
                st_stat = o.stop_stat;
                st_p = o.stop_pval;
                st_pow = o.stop_power;
                st_rank = o.stop_rank;
            }
        } else if (o.best_pval >= bp) {
            bp = o.best_pval;
            bs = o.best_stat;
            bg = o.stop_stat;  // (records without a stop carry the G^2 of their maximum here)
            bdf = o.best_df;
            bi = sg;
        }
    }
    int first = my_stop;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ev += __shfl_xor(ev, o);
        const int f2 = __shfl_xor(first, o);
        first = f2 < first ? f2 : first;
        const double p2 = __shfl_xor(bp, o), s2 = __shfl_xor(bs, o), g2 = __shfl_xor(bg, o);
        const int i2 = __shfl_xor(bi, o), d2 = __shfl_xor(bdf, o);
        if (p2 > bp || (p2 == bp && i2 > bi)) {
            bp = p2;
            bs = s2;
            bg = g2;
            bdf = d2;
            bi = i2;
        }
    }
    DhMerge M;
    M.ev = ev;
    if (first != 0x7fffffff) {
        const int owner = first & 63;
        M.stop = true;
        M.stat = __shfl(st_stat, owner);
        M.p = __shfl(st_p, owner);
        M.pow = __shfl(st_pow, owner);
        M.nt = __shfl(st_rank, owner) + 1ull;
    } else {
        M.stop = false;
        M.stat = bs;
        M.p = bi >= 0 ? bp : -2.0;
        M.pow = 1;
        M.nt = 0ull;
    }
    M.g = bg;
    M.df = bdf;
    return M;
}

// claim and evaluate one record of board b (if any is left); true if a record was processed.  acc_own: the caller published the board (its accepted list is at hand); otherwise the list is
// staged from the board's write-through copy.
template <int L, int NXY, int PRE, bool R4>
__device__ __forceinline__ bool mi_board_work(MiBoard *__restrict__ b, FwSegOut *__restrict__ res, const int32_t *__restrict__ bacc,
                                              const int32_t *acc_own, int lane)
{
    const DhParams &P = dh_mi_ctx.P;
    const unsigned long long nr = mi_ld_u64(&b->nr);
    const unsigned int nch = (unsigned int)nr, res_off = (unsigned int)(nr >> 32);
    if (mi_ld_u32(&b->next_chunk) >= nch) return false;
    const unsigned int c = mi_wave_add(&b->next_chunk, 1u, lane);
    if (c >= nch) return false;
    const unsigned long long tc = mi_ld_u64(&b->tc), ac = mi_ld_u64(&b->ac);
    const int a = (int)(unsigned int)ac, chunk = (int)(unsigned int)(ac >> 32);
    const unsigned long long start = mi_ld_u64(&b->start), end = mi_ld_u64(&b->end);
    const unsigned long long r0 = start + (unsigned long long)c * (unsigned long long)chunk;
    unsigned long long r1 = r0 + (unsigned long long)chunk;
    if (r1 > end) r1 = end;
    FwSegOut o;
    if (mi_ld_u64(&b->stop_min) < r0) {  // cancelled: an earlier rank stopped the job
        o.stop_rank = FW_RANK_NONE;
        o.stop_stat = o.stop_pval = 0.0;
        o.best_rank = 0ull;
        o.best_stat = 0.0;
        o.best_pval = -3.0;
        o.stop_df = o.stop_power = o.best_df = o.pad = 0;
        o.evaluated = 0ull;
    } else {
        const int32_t *acc = acc_own;
        int remote = 0;
        if (!acc_own) {
            const int32_t *src = bacc + mi_ld_u64(&b->acc_off);
            if (a <= MI_ACC_LDS) {  // one coalesced sc1 sweep into this wavefront's LDS slot
                int32_t *slot = dh_mi_acc[threadIdx.x >> 6];
                for (int i = lane; i < a; i += 64) slot[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                acc = slot;
            } else {
                acc = src;
                remote = 1;
            }
        }
        const MiBest *seed = nullptr;
        const double sp = __longlong_as_double((long long)mi_ld_u64((const unsigned long long *)&b->seed_p));
        if (sp >= 0.0) {
            MiBest &sd = dh_mi_seed[threadIdx.x >> 6];
            sd.p = sp;
            sd.stat = 0.0;  // (never reported: a record reports a maximum only if one of its own tests reaches the seed's p)
            sd.g = __longlong_as_double((long long)mi_ld_u64((const unsigned long long *)&b->seed_g));
            sd.df = (int)mi_ld_u32(&b->seed_df);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            seed = &sd;
        }
        o = mi_run_ranks_sel<L, NXY, PRE, R4>((int)(unsigned int)tc, (int)(unsigned int)(tc >> 32), acc, a, P.max_k, P.max_tests, r0, r1,
                                      &b->stop_min, remote, seed);
        if (o.stop_rank != FW_RANK_NONE && lane == 0) atomicMin(&b->stop_min, o.stop_rank);
    }
    if (lane == 0) {
        mi_record_store(res + res_off + c, o);
        mi_drain();  // the record before the count
        atomicAdd(&b->done, 1u);
    }
    return true;
}

// look for an open board (from the first one that still has unclaimed records) and work on one record of it; true if
// something was done.  (A global FIFO of records with a compare-and-swap head was tried instead of the scan: 10x slower --
// a thousand wavefronts polling and swapping the same two words.)
template <int L, int NXY, int PRE, bool R4>
__device__ __noinline__ bool mi_help_scan(MiQueue *__restrict__ Q, MiBoard *__restrict__ boards, FwSegOut *__restrict__ res,
                                          const int32_t *__restrict__ bacc, int lane, unsigned int nb, unsigned int i)
{
    for (; i < nb; ++i) {
        MiBoard *b = boards + i;
        if (mi_ld_u32(&b->ready) == 0u) return false;  // reserved, not yet filled
        if (mi_ld_u32(&b->next_chunk) < (unsigned int)mi_ld_u64(&b->nr)) {
            if (mi_board_work<L, NXY, PRE, R4>(b, res, bacc, nullptr, lane)) return true;
        } else if (i == mi_ld_u32(&Q->hint) && lane == 0) {
            atomicMax(&Q->hint, i + 1u);  // every record of this board is taken: later scans start behind it
        }
    }
    return false;
}
// The look that finds nothing -- no board behind the hint: what a wavefront that has run out of targets sees thousands of times while
// the last heavy targets finish -- stays in the caller.  r05 made the call first: the out-of-line scan keeps ~38 values in callee-saved
// registers across its own call of the test routine and parks them on the stack in its prologue, so every empty poll wrote and read
// back 38 x 256 bytes of scratch (profiles/r06_discrete_kernel_writes.txt).
template <int L, int NXY, int PRE, bool R4>
__device__ __forceinline__ bool mi_help(MiQueue *__restrict__ Q, MiBoard *__restrict__ boards, FwSegOut *__restrict__ res,
                                        const int32_t *__restrict__ bacc, int lane)
{
    unsigned int nb = mi_ld_u32(&Q->n_boards);
    if (nb > MI_BOARD_CAP) nb = MI_BOARD_CAP;
    const unsigned int i = mi_ld_u32(&Q->hint);
    if (i >= nb) return false;
    if (mi_ld_u32(&boards[i].ready) == 0u) return false;  // reserved, not yet filled (the scan would return at once)
    return mi_help_scan<L, NXY, PRE, R4>(Q, boards, res, bacc, lane, nb, i);
}

// ---- heavy targets: one WORKGROUP per target -------------------------------------------------------------------------------------
// cfg4's last feed-forward round (r03 trace, `FW_TRACE_HOST`): the six targets that finish last (126-196 candidates, ~200 jobs each)
// are taken at t = 0 and end at 39-49 ms of a 48 ms launch, 21 ms of it in the owner's sequential prefixes (16 tests x 7 us, one after
// the other) and 26 ms in board phases of ~40 tests in which helpers arrive late (every other wavefront is inside a prefix of its own).
// The chain of such a target is the launch.  So the first mi_team targets of the heaviest-first list get the four wavefronts of a
// workgroup: the leader (wavefront 0) runs the state machine; every job starts with lock-step rounds in which wavefront w evaluates
// rank(s) base + w (the records meet in LDS and are merged in rank order exactly like board records: first stop wins, else the
// `>=` maximum), and what is left of a long enumeration goes to a board as before -- with three helpers that are there at once.
struct MiTeamJob {
    int go, cand, a, tail;
    int hw;                   // R4: the MiAhead slot that already holds this job's first test (-1: none) -- decided by the leader alone
    long long acc_off;
    unsigned long long N;
    unsigned int board;       // board of the current window (MI_BOARD_CAP: none, the leader ran the window alone)
    DhMerge mg;               // merged outcome of the current window
};
__shared__ MiTeamJob dh_mi_tj;
__shared__ FwSegOut dh_mi_trec[2][4];

// allocate + publish a board for ranks [next, next + W) of the job; returns false if the launch is out of board / record space
__device__ __forceinline__ bool mi_publish(MiQueue *__restrict__ Q, MiBoard *__restrict__ boards, int32_t *__restrict__ bacc, unsigned int &bacc_off,
                                           const int32_t *__restrict__ acc, int a, int T, int cand, unsigned long long next, unsigned long long W,
                                           unsigned long long chunk, unsigned int nch, int lane, unsigned int &bi, unsigned int &ro,
                                           double seed_p, double seed_g, int seed_df)
{
    bi = MI_BOARD_CAP;
    ro = MI_REC_CAP;
    if (bacc_off == MI_BACC_CAP && mi_ld_u32(&Q->bacc_top) + (unsigned int)a <= MI_BACC_CAP) {
        bacc_off = mi_wave_add(&Q->bacc_top, (unsigned int)a, lane);
        if (bacc_off + (unsigned int)a > MI_BACC_CAP) {
            bacc_off = MI_BACC_CAP;
        } else {
            for (int i = lane; i < a; i += 64) __hip_atomic_store(bacc + bacc_off + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bacc_off != MI_BACC_CAP && mi_ld_u32(&Q->n_boards) < MI_BOARD_CAP && mi_ld_u32(&Q->res_top) + nch <= MI_REC_CAP) {
        ro = mi_wave_add(&Q->res_top, nch, lane);
        if (ro + nch <= MI_REC_CAP) bi = mi_wave_add(&Q->n_boards, 1u, lane);
    }
    if (bi >= MI_BOARD_CAP) return false;
    MiBoard *b = boards + bi;
    mi_drain();  // the copy of the accepted list (every lane's stores) ...
    if (lane == 0) {
        mi_st_u64(&b->tc, (unsigned long long)(unsigned int)T | ((unsigned long long)(unsigned int)cand << 32));
        mi_st_u64(&b->ac, (unsigned long long)(unsigned int)a | ((unsigned long long)chunk << 32));
        mi_st_u64(&b->acc_off, (unsigned long long)bacc_off);
        mi_st_u64(&b->start, next);
        mi_st_u64(&b->end, next + W);
        mi_st_u64(&b->nr, (unsigned long long)nch | ((unsigned long long)ro << 32));
        mi_st_u64(&b->stop_min, FW_RANK_NONE);  // next_chunk / done are zero from the launch's memset
        mi_st_u64((unsigned long long *)&b->seed_p, (unsigned long long)__double_as_longlong(seed_p > 1e-290 ? seed_p : -1.0));  // (p = 0 seeds nothing: ties)
        mi_st_u64((unsigned long long *)&b->seed_g, (unsigned long long)__double_as_longlong(seed_g));
        mi_st_u32(&b->seed_df, (unsigned int)seed_df);
        mi_drain();  // ... and the board before the flag
        mi_st_u32(&b->ready, 1u);
    }
    return true;
}

// the whole HITON-PC of target t on the four wavefronts of this workgroup (every wavefront calls it; uniform control flow
// between the barriers: every decision is taken from LDS values all four read)
// the persistent kernel's own clock reads (per job: phase times for FW_TRACE_HOST): a wall_clock64() is a scalar memory round trip,
// four of them per job were ~5 % of a light job -- taken only when the host asks for the trace (DhParams::mi_trace)
#define MI_CLK() (P.mi_trace ? wall_clock64() : 0ull)
template <int L, int NXY, int PRE, bool R4>
__device__ __noinline__ void dh_mi_team(DhTgt *__restrict__ tg, int ntg, int t, MiQueue *__restrict__ Q,
                                        MiBoard *__restrict__ boards, FwSegOut *__restrict__ res, int32_t *__restrict__ bacc)
{
    const DhArrays &A = dh_mi_ctx.A;
    const DhParams &P = dh_mi_ctx.P;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    DhTgt &x = dh_mi_x[0];
    MiTeamJob &J = dh_mi_tj;
    if (wave == 0 && lane == 0) {
        x = tg[t];
        if (A.wl_cnt) x.wl_n = (int32_t)A.wl_cnt[x.T];  // device-built whitelist: what the earlier launches of the schedule appended
        x.r_first0 = (unsigned int)MI_CLK();
    }
    __syncthreads();
    const unsigned long long per = R4 ? 4ull : 1ull;  // ranks per wavefront and lock-step round
    for (;;) {
        if (wave == 0) {
            const bool more = dh_advance(x, A, lane, 1);
            if (lane == 0) {
                J.go = more ? 1 : 0;
                if (more) {
                    const int32_t *cands = x.phase == 0 ? A.cand0 + x.cand_off : A.tpc_key + x.co;
                    J.cand = cands[x.pos];
                    J.acc_off = DH_ACC_OFF(x, x.cur, 1);
                    J.a = x.na;
                    J.N = dh_enum_size(x.na, P.max_k, P.max_tests);
                    J.tail = (P.mi_team_tail || (mi_ld_u32(&Q->next_target) + P.mi_team >= (unsigned int)ntg &&
                                                 ((unsigned int)ntg - mi_ld_u32(&Q->targets_done)) * 8u <= gridDim.x * 4u)) ? 1 : 0;
                    // Which look-ahead slot (if any) holds this job's first test is decided HERE, by the leader, between the barrier that
                    // ends the previous job and the one that starts this one: no wavefront writes a slot in that interval.  (r05 let every
                    // wavefront scan the slots inside the job, while wavefront 0 could already be rewriting its own in mi_first4: a late
                    // wavefront then saw the NEW slot 0, skipped the compute branch and its barrier, and ran one barrier behind the others
                    // from there on -- one of the two races behind the pass-to-pass differences of profiles/r05_discrete_pass_to_pass.txt.)
                    int hw = -1;
                    if (R4 && P.mi_ahead && x.phase == 0 && x.na >= 1 && J.N >= 1ull) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const MiAhead &Hq = dh_mi_ahead[w];
                            if (Hq.n > 0 && Hq.T == x.T && Hq.na == x.na && x.pos >= Hq.pos0 && x.pos < Hq.pos0 + Hq.n) hw = w;
                        }
                    }
                    J.hw = hw;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (wave == 0 && J.go && J.a <= MI_ACC_LDS) {  // the job's accepted list: one LDS copy for the four wavefronts (the leader wrote the
            const int32_t *src = A.acc + J.acc_off;    // global one itself: the others must not meet it in a cache of their own)
            for (int i = lane; i < J.a; i += 64) dh_mi_acc[0][i] = src[i];
        }
        __syncthreads();
        if (!J.go) break;
        const int cand = J.cand, a = J.a, T = x.T;
        const int32_t *acc = a <= MI_ACC_LDS ? (const int32_t *)dh_mi_acc[0] : A.acc + J.acc_off;
        const unsigned long long N = J.N;
        const unsigned long long tk1 = MI_CLK();
        // lock-step rounds: wavefront w takes ranks next + w * per ...
        unsigned long long next = 0ull, ev = 0ull, nt = 0ull;
        bool stopped = false;
        double r_stat = 0.0, r_p = 0.0, best_p = -1.0, best_stat = 0.0, best_g = 0.0;
        int r_pow = 1, best_df = 0;
        unsigned int n_rounds = 0u;
        if constexpr (R4) {
            // First tests of up to SIXTEEN candidates in one round (r05; the one-wavefront form of this is mi_first4 in the kernel below).
            // A heavy target of cfg2 is a chain of ~400 jobs of which ~390 end with their first test: while candidates are rejected the
            // accepted list does not change, so the first test of the next candidates -- (T, c | the first subset of the same list) --
            // does not depend on the earlier verdicts.  The four wavefronts take consecutive slices of the candidate list (a slice: up to
            // four candidates that can share a mi_test_core4 step, cut in front of a whitelisted one: it joins without a test and changes
            // the list), the results stay in the wavefronts' MiAhead slots and are consumed in candidate order while (target, accepted
            // length) still match; an accepted candidate invalidates the rest.  r04: every job of a team target started with a
            // lock-step round of 16 ranks of ITS OWN enumeration, 15 of them behind the stop (cfg2: 946 000 executed tests for 287 000).
            if (P.mi_ahead && x.phase == 0 && a >= 1 && N >= 1ull) {
                const int32_t *cands = A.cand0 + x.cand_off;
                int hw = J.hw;  // (the leader's decision: see where J is written)
                unsigned long long computed = 0ull;
                if (hw < 0 && x.nc - x.pos >= 2) {  // (uniform: x and J do not change between the job's first barrier and the one in front of its commit)
                    const MiDev Mu = mi_uniform(dh_mi_ctx.M);
                    int start = x.pos, my_start = x.pos, my_n = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int nb = x.nc - start < 4 ? x.nc - start : 4;
                        if (nb <= 0) break;
                        int n = mi_first_run(Mu, cands + start, nb);  // (a cut by the levels only ends the slice: the next wavefront goes on)
                        bool wl_cut = false;
                        if (x.wl_n > 0)  // a whitelisted candidate ends the chain of first tests (the candidate at x.pos itself is not one: dh_advance)
                            for (int q = (start == x.pos ? 1 : 0); q < n; ++q)
                                if (dh_in_wl(x, A, cands[start + q])) {
                                    n = q;
                                    wl_cut = true;
                                    break;
                                }
                        if (w == wave) {
                            my_start = start;
                            my_n = n;
                        }
                        start += n;
                        if (wl_cut) break;
                    }
                    computed = (unsigned long long)(start - x.pos);
                    if (my_n > 0)
                        mi_first4<L>(T, my_start, cands + my_start, my_n, acc, a, P.max_k, P.max_tests);
                    else if (lane == 0)
                        dh_mi_ahead[wave].n = 0;
                    __syncthreads();
                    hw = 0;  // (wavefront 0's slice starts at x.pos and holds at least that candidate)
                }
                if (hw >= 0) {
                    const MiAhead &Hq = dh_mi_ahead[hw];
                    const int t = x.pos - Hq.pos0;
                    ev += computed;
                    if (Hq.stop[t]) {
                        stopped = true;
                        r_stat = Hq.stat[t];
                        r_p = Hq.pval[t];
                        r_pow = Hq.power[t];
                        nt = 1ull;
                    } else {  // significant: the job goes on at rank 1, seeded with its first test
                        best_p = Hq.pval[t];
                        best_stat = Hq.stat[t];
                        best_g = Hq.g[t];
                        best_df = Hq.df[t];
                        next = 1ull;
                    }
                }
            }
        }
        for (unsigned int step = 0u; !stopped && next < N && step < P.mi_team_steps; ++step) {
            ++n_rounds;
            const unsigned long long r0 = next + (unsigned long long)wave * per;
            unsigned long long r1 = r0 + per;
            if (r1 > N) r1 = N;
            FwSegOut o;
            const unsigned long long tq0 = MI_CLK();
            if (r0 < N) {
                const MiBest *seed = nullptr;
                if (best_p > 1e-290) {  // what the earlier rounds found: the tests it dominates skip their Q(a, x)
                    MiBest &sd = dh_mi_seed[wave];
                    sd.p = best_p;
                    sd.stat = best_stat;
                    sd.g = best_g;
                    sd.df = best_df;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    seed = &sd;
                }
                o = mi_run_ranks_sel<L, NXY, PRE, R4>(T, cand, acc, a, P.max_k, P.max_tests, r0, r1, nullptr, 0, seed);
            } else {
                o.stop_rank = FW_RANK_NONE;
                o.stop_stat = o.stop_pval = 0.0;
                o.best_rank = 0ull;
                o.best_stat = 0.0;
                o.best_pval = -3.0;
                o.stop_df = o.stop_power = o.best_df = o.pad = 0;
                o.evaluated = 0ull;
            }
            if (lane == 0) dh_mi_trec[step & 1u][wave] = o;
            const unsigned long long tq1 = MI_CLK();
            __syncthreads();  // (two buffers: the records of round s are rewritten in round s + 2, behind the barrier of round s + 1)
#ifdef FW_MI_TICKS
            if (lane == 0) {
                atomicAdd(&Q->tm_run, tq1 - tq0);
                atomicAdd(&Q->tm_wait, MI_CLK() - tq1);
                atomicAdd(&Q->tm_steps, 1ull);
                atomicAdd(&Q->tm_tests, o.evaluated);
            }
#else
            (void)tq0;
            (void)tq1;
#endif
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const FwSegOut &q = dh_mi_trec[step & 1u][w];
                ev += q.evaluated;
                if (stopped) continue;
                if (q.stop_rank != FW_RANK_NONE) {
                    stopped = true;
                    r_stat = q.stop_stat;
                    r_p = q.stop_pval;
                    r_pow = q.stop_power;
                    nt = q.stop_rank + 1ull;
                } else if (q.best_pval >= 0.0 && q.best_pval >= best_p) {
                    best_p = q.best_pval;
                    best_stat = q.best_stat;
                    best_g = q.stop_stat;
                    best_df = q.best_df;
                }
            }
            next += 4ull * per;
            if (next > N) next = N;
        }
        const unsigned long long tk2 = MI_CLK();
        // ... the rest of a long enumeration: boards (the leader publishes and merges, all four work on the records)
        const bool tail = J.tail != 0;
        unsigned long long width = (unsigned long long)(tail ? P.mi_win0_tail : P.mi_win0);
        const unsigned long long chunk_min = tail ? P.mi_chunk_tail : P.mi_chunk_min;
        unsigned int bacc_off = MI_BACC_CAP;
        bool boarded = false;
        while (!stopped && next < N) {
            boarded = true;
            const unsigned long long W = (N - next) < width ? (N - next) : width;
            unsigned long long chunk = W / (unsigned long long)P.mi_chunk_div;
            chunk = chunk < chunk_min ? chunk_min : (chunk > P.mi_chunk_max ? P.mi_chunk_max : chunk);
            const unsigned int nch = (unsigned int)((W + chunk - 1ull) / chunk);
            if (wave == 0) {
                unsigned int bi, ro;
                const bool ok = mi_publish(Q, boards, bacc, bacc_off, acc, a, T, cand, next, W, chunk, nch, lane, bi, ro, best_p, best_g, best_df);
                if (!ok) {  // out of board space (never at the benchmark sizes): the leader carries on alone
                    const FwSegOut q = mi_run_ranks_sel<L, NXY, PRE, R4>(T, cand, acc, a, P.max_k, P.max_tests, next, next + W, nullptr, 0, nullptr);
                    if (lane == 0) {
                        J.mg.g = q.stop_stat;
                        J.mg.df = q.best_df;
                        J.mg.stop = q.stop_rank != FW_RANK_NONE;
                        J.mg.stat = J.mg.stop ? q.stop_stat : q.best_stat;
                        J.mg.p = J.mg.stop ? q.stop_pval : (q.best_pval < 0.0 ? -2.0 : q.best_pval);
                        J.mg.pow = q.stop_power;
                        J.mg.nt = q.stop_rank + 1ull;
                        J.mg.ev = q.evaluated;
                    }
                }
                if (lane == 0) J.board = ok ? bi : MI_BOARD_CAP;
            }
            __syncthreads();
            const unsigned int bi = J.board;
            if (bi < MI_BOARD_CAP) {
                MiBoard *b = boards + bi;
                while (mi_board_work<L, NXY, PRE, R4>(b, res, bacc, acc, lane)) {
                }
                if (wave == 0) {
                    MiWatch wd;
                    unsigned int dn = mi_ld_u32(&b->done);
                    mi_watch_begin(wd, dn);
                    while (dn < nch) {  // records claimed by other wavefronts: they are running
                        mi_nap_wait();
                        dn = mi_ld_u32(&b->done);
                        if (mi_watch_expired(wd, dn)) {
                            if (lane == 0) atomicExch(&Q->pad[0], 3u);
                            break;
                        }
                    }
                    const DhMerge mg = mi_merge(res, (long long)(unsigned int)(mi_ld_u64(&b->nr) >> 32), (int)nch, lane);
                    if (lane == 0) J.mg = mg;
                }
            }
            __syncthreads();
            const DhMerge mg = J.mg;
            ev += mg.ev;
            if (mg.stop) {
                stopped = true;
                r_stat = mg.stat;
                r_p = mg.p;
                r_pow = mg.pow;
                nt = mg.nt;
            } else if (mg.p != -2.0 && mg.p >= best_p) {
                best_p = mg.p;
                best_stat = mg.stat;
                best_g = mg.g;
                best_df = mg.df;
            }
            next += W;
            width *= 8ull;
        }
        if (!stopped) {  // every subset significant: the maximum-p result (tests.jl:338-345)
            r_stat = best_stat;
            r_p = best_p < 0.0 ? 0.0 : best_p;
            r_pow = 1;
            nt = N;
        }
        // The job's verdict changes the target state in LDS (dh_commit: x.pos, x.na, ...) and every wavefront reads that state while it
        // works on the job (x.pos, x.nc, x.phase, the whitelist, the look-ahead slot of x.pos): nobody may still be reading when the leader
        // commits.  r05 had this barrier BEHIND the commit only; a job that ended with a cached first test had no barrier at all between
        // its start and the commit, a late wavefront read the advanced x.pos, took another branch with a barrier of its own and stayed one
        // barrier behind (wrong records in its hands, one target's PC list differing about once in a thousand passes, or a hang).
        // Behind the commit the other wavefronts read nothing until the next job's first barrier, so this is the only one needed.
        __syncthreads();
        if (wave == 0) {
            const unsigned long long tk3 = MI_CLK();
            if (lane == 0) {
                x.r_first1 += (unsigned int)(tk2 - tk1);
                x.r_more1 += (unsigned int)(tk3 - tk2);
                if (boarded) x.c_eval_short += 1ull;
                x.c_eval_short += (unsigned long long)n_rounds << 32;  // (trace: lock-step rounds of the target in the high word)
            }
            x.c_ref += nt;
            x.c_calls += 1ull;
            x.c_eval += ev;
            x.c_alg += (P.max_k <= 3 && a <= FW_UNRANK32_A) ? dh_alg_bytes_disc32(a, ev, P.max_k, P.disc_bytes_per_col)
                                                            : dh_alg_bytes(a, ev, P.max_k, P.disc_bytes_per_col);
            dh_commit(x, A, lane, 1, r_stat, r_p, r_pow, P.alpha);
        }
    }
    if (wave == 0 && lane == 0) {
        x.r_more0 = (unsigned int)MI_CLK();
        tg[t] = x;
        atomicAdd(&Q->targets_done, 1u);
    }
    __syncthreads();
}

#ifdef FW_MI_TICKS
#define MI_TICK(slot, t0) do { if ((threadIdx.x & 63) == 0) dh_mi_ticks[threadIdx.x >> 6][slot] += wall_clock64() - (t0); } while (0)
#else
#define MI_TICK(slot, t0) do { (void)(t0); } while (0)
#endif
#ifndef DH_MI_OCC
#define DH_MI_OCC 1  // workgroups per CU the register budget is sized for
#endif
template <int L, int NXY, int PRE, bool R4>
__global__ __launch_bounds__(256, DH_MI_OCC) void dh_mi_target_kernel(DhTgt *__restrict__ tg, int ntg, const int32_t *__restrict__ order,
                                                           DhArrays A, MiDev M, DhParams P, MiQueue *__restrict__ Q,
                                                           MiBoard *__restrict__ boards, FwSegOut *__restrict__ res,
                                                           int32_t *__restrict__ bacc)
{
    const int lane = threadIdx.x & 63;
    // 100 MHz ticks per wavefront, summed into MiQueue at the end (FW_TRACE_HOST prints the averages): own first tests, board
    // phases (own records + waiting / helping), helping before jobs, the tail after the last target
    unsigned long long tk_seq = 0ull, tk_board = 0ull, tk_help = 0ull, tk_tail = 0ull;
#ifdef FW_MI_TICKS
    if (lane < 12) dh_mi_ticks[threadIdx.x >> 6][lane] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
    if (lane == 0) dh_mi_ahead[threadIdx.x >> 6].n = 0;  // (no look-ahead results yet)
    if (threadIdx.x == 0) {  // the launch context of the out-of-line routines (see DhMiCtx)
        dh_mi_ctx.A = A;
        dh_mi_ctx.M = M;
        dh_mi_ctx.P = P;
    }
    __syncthreads();
    const unsigned long long tk_begin = MI_CLK();
    // the heaviest targets (the first mi_team of the list): a workgroup each, all four wavefronts on it (dh_mi_team), taken in list
    // order by whichever workgroup is free; the list proper starts behind them
    for (;;) {
        if (threadIdx.x == 0) dh_mi_tj.go = (int)atomicAdd(&Q->next_team, 1u);
        __syncthreads();
        const unsigned int ts = (unsigned int)dh_mi_tj.go;
        __syncthreads();  // (dh_mi_team rewrites the descriptor)
        if (ts >= P.mi_team) break;
        dh_mi_team<L, NXY, PRE, R4>(tg, ntg, order[ts], Q, boards, res, bacc);
    }
    unsigned int jobctr = 0u;
    bool tail_seen = false;
    for (;;) {
        const unsigned int slot = P.mi_team + mi_wave_add(&Q->next_target, 1u, lane);
        if (slot >= (unsigned int)ntg) break;
        const int t = order[slot];
        // the target's state lives in LDS while its jobs run: every lane holds the same copy, and ~50 registers of it live across
        // the out-of-line test routine were part of what kept this kernel at one wavefront per SIMD
        DhTgt &x = dh_mi_x[threadIdx.x >> 6];
        if (lane == 0) {
            x = tg[t];
            if (A.wl_cnt) x.wl_n = (int32_t)A.wl_cnt[x.T];  // device-built whitelist: what the earlier launches of the schedule appended
            x.r_first0 = (unsigned int)MI_CLK();  // FW_TRACE_HOST: when the target was taken / finished (100 MHz ticks, low word)
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            const unsigned long long tka0 = MI_CLK();
            const bool more_jobs = dh_advance(x, A, lane, 1);
            MI_TICK(6, tka0);
            if (!more_jobs) break;
            // other targets' big enumerations first: they are the critical path of the pass
            const unsigned long long tk0 = MI_CLK();
            // ... unless this target is one of the heavy ones: its own chain of jobs IS the critical path (cfg2 / cfg4: the
            // launch ended when the heaviest target did, 13 ms after the average wavefront had run out of targets)
            const bool heavy = P.mi_heavy > 0u && (unsigned int)x.nc >= P.mi_heavy;
            // the shared words (boards, target counters) are looked at before every FOURTH job: each look is two to four sc1 round
            // trips of ~2 us, and a chain of jobs that end with their first test (or cost nothing at all: mi_first4) paid more for
            // them than for its tests (cfg2 trace: 18.8 us per job against a 12 us test)
            const bool poll = (jobctr++ & 3u) == 0u;
            mi_u32x4 q4 = {0u, 0u, 0u, 0u};  // {next_target, targets_done, n_boards, hint}
            if (poll) q4 = mi_ld_u128(Q);
            if (P.mi_help_jobs && !heavy && poll && q4.z > q4.w)
                while (mi_help<L, NXY, PRE, R4>(Q, boards, res, bacc, lane) && mi_ld_u32(&Q->n_boards) > mi_ld_u32(&Q->hint)) {
                }
            const unsigned long long tk1 = MI_CLK();
            tk_help += tk1 - tk0;
            const int32_t *cands = x.phase == 0 ? A.cand0 + x.cand_off : A.tpc_key + x.co;
            const int32_t cand = cands[x.pos];
            const long long acc_off = DH_ACC_OFF(x, x.cur, 1);
            const int a = x.na;
            const unsigned long long N = dh_enum_size(a, P.max_k, P.max_tests);
            // the first tests: alone.  Elimination-phase jobs nearly always run to the end (the member passed every test against
            // almost this pool a moment ago): a big one goes to a board at once, whole enumeration in one window
            const bool elim_full = x.phase == 1 && P.mi_elim_min > 0u && N > (unsigned long long)P.mi_elim_min;
            // the owner's sequential prefix.  While every wavefront still has targets of its own, boards cost more than they give
            // (a prefix of 4 instead of 16: 55 -> 72 ms at cfg4 on one GPU); once the target list is exhausted the idle wavefronts
            // do nothing but poll, and the chains of the last heavy targets ARE the rest of the pass: hand over after mi_seq_tail
            // tests (one rank of eight, cfg4: 46.9 -> 37.2 ms of conditional stage)
            // (tail: no target left to claim AND fewer than one wavefront in eight still owns one -- cfg2 has as many targets as
            // the launch has wavefronts: "list exhausted" alone switched every job of the pass to the short prefix, 15 -> 22 ms)
            if (poll) tail_seen = q4.x + P.mi_team >= (unsigned int)ntg && ((unsigned int)ntg - q4.y) * 8u <= gridDim.x * 4u;
            const bool tail = tail_seen;
            const unsigned long long seq = tail ? P.mi_seq_tail : (heavy ? P.mi_seq_heavy : P.mi_seq);
            unsigned long long next = elim_full ? 0ull : (N < seq ? N : seq);
            MI_TICK(7, tk1);
            FwSegOut o;
            bool first_known = false;
            if constexpr (R4) {
                MiAhead &H = dh_mi_ahead[threadIdx.x >> 6];
                if (P.mi_ahead && x.phase == 0 && a >= 1 && next >= 1ull) {
                    bool hit = H.T == x.T && H.na == a && x.pos >= H.pos0 && x.pos < H.pos0 + H.n;
                    unsigned long long computed = 0ull;
                    if (!hit && x.nc - x.pos >= 2) {
                        const int nb = x.nc - x.pos < 4 ? x.nc - x.pos : 4;
                        mi_first4<L>(x.T, x.pos, cands + x.pos, nb, A.acc + acc_off, a, P.max_k, P.max_tests);
                        computed = (unsigned long long)H.n;
                        hit = true;
                    }
                    if (hit) {
                        const int t = x.pos - H.pos0;
                        first_known = true;
                        o.stop_rank = FW_RANK_NONE;
                        o.stop_stat = o.stop_pval = 0.0;
                        o.best_rank = 0ull;
                        o.best_stat = 0.0;
                        o.best_pval = -3.0;
                        o.stop_df = o.stop_power = o.best_df = o.pad = 0;
                        o.evaluated = 0ull;
                        if (H.stop[t]) {
                            o.stop_rank = 0ull;
                            o.stop_stat = H.stat[t];
                            o.stop_pval = H.pval[t];
                            o.stop_df = H.df[t];
                            o.stop_power = H.power[t];
                        } else {
                            if (next > 1ull) {  // ranks 1 .. of this candidate's own job, seeded with its first test
                                MiBest &sd = dh_mi_seed[threadIdx.x >> 6];
                                if (lane == 0) {
                                    sd.p = H.pval[t];
                                    sd.stat = H.stat[t];
                                    sd.g = H.g[t];
                                    sd.df = H.df[t];
                                }
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                                o = mi_run_ranks_sel<L, NXY, PRE, R4>(x.T, cand, A.acc + acc_off, a, P.max_k, P.max_tests, 1ull, next, nullptr, 0,
                                                                      H.pval[t] > 1e-290 ? &sd : nullptr);
                            }
                            if (o.stop_rank == FW_RANK_NONE && !(o.best_pval >= H.pval[t])) {  // the first test is still the maximum
                                o.best_pval = H.pval[t];
                                o.best_stat = H.stat[t];
                                o.best_df = H.df[t];
                                o.stop_stat = H.g[t];
                            }
                        }
                        o.evaluated += computed;
                    }
                }
            }
            if (!first_known)
                o = mi_run_ranks_sel<L, NXY, PRE, R4>(x.T, cand, A.acc + acc_off, a, P.max_k, P.max_tests, 0ull, next, nullptr, 0, nullptr);
            unsigned long long ev = o.evaluated, nt = 0ull;
            bool stopped = o.stop_rank != FW_RANK_NONE;
            double r_stat = stopped ? o.stop_stat : 0.0, r_p = stopped ? o.stop_pval : 0.0;
            int r_pow = stopped ? o.stop_power : 1;
            double best_p = o.best_pval, best_stat = o.best_stat, best_g = stopped ? 0.0 : o.stop_stat;  // (+ G^2, df: the seed of the boards)
            int best_df = o.best_df;
            if (stopped) nt = o.stop_rank + 1ull;
            // (in the tail the records are short and the first window wide: the wavefronts that poll outnumber the records of a window)
            unsigned long long width = elim_full ? N : (unsigned long long)(tail ? P.mi_win0_tail : P.mi_win0);
            const unsigned long long chunk_min = tail ? P.mi_chunk_tail : P.mi_chunk_min;
            unsigned int bacc_off = MI_BACC_CAP;  // this job's write-through copy of its accepted list (made with its first board)
            const unsigned long long tk2 = MI_CLK();
            tk_seq += tk2 - tk1;
            if (lane == 0) x.r_first1 += (unsigned int)(tk2 - tk1);  // FW_TRACE_HOST: this target's ticks in sequential prefixes / board phases
            while (!stopped && next < N) {
                const unsigned long long W = (N - next) < width ? (N - next) : width;
                unsigned long long chunk = W / (unsigned long long)P.mi_chunk_div;
                chunk = chunk < chunk_min ? chunk_min : (chunk > P.mi_chunk_max ? P.mi_chunk_max : chunk);
                const unsigned int nch = (unsigned int)((W + chunk - 1ull) / chunk);
                unsigned int bi, ro;
                const unsigned long long tkp0 = MI_CLK();
                const bool published = mi_publish(Q, boards, bacc, bacc_off, A.acc + acc_off, a, x.T, cand, next, W, chunk, nch, lane, bi, ro,
                                                  best_p, best_g, best_df);
                MI_TICK(9, tkp0);
                DhMerge mg;
                if (!published) {  // out of board space (never at the benchmark sizes): the owner carries on alone
                    const FwSegOut q = mi_run_ranks_sel<L, NXY, PRE, R4>(x.T, cand, A.acc + acc_off, a, P.max_k, P.max_tests, next, next + W, nullptr, 0, nullptr);
                    mg.stop = q.stop_rank != FW_RANK_NONE;
                    mg.stat = mg.stop ? q.stop_stat : q.best_stat;
                    mg.p = mg.stop ? q.stop_pval : (q.best_pval < 0.0 ? -2.0 : q.best_pval);
                    mg.pow = q.stop_power;
                    mg.nt = q.stop_rank + 1ull;
                    mg.ev = q.evaluated;
                    mg.g = q.stop_stat;
                    mg.df = q.best_df;
                } else {
                    MiBoard *b = boards + bi;
                    while (mi_board_work<L, NXY, PRE, R4>(b, res, bacc, A.acc + acc_off, lane)) {
                    }
                    MiWatch wd;
                    unsigned int dn = mi_ld_u32(&b->done);
                    mi_watch_begin(wd, dn);
                    while (dn < nch) {  // records claimed by other wavefronts: they are running
                        if (heavy || !mi_help<L, NXY, PRE, R4>(Q, boards, res, bacc, lane)) {
                            const unsigned long long tkw0 = MI_CLK();
                            mi_nap_wait();
                            MI_TICK(10, tkw0);
                            if (mi_watch_expired(wd, mi_ld_u32(&b->done))) {  // 2 s without a record finishing: a logic error, not a workload -- report instead of hanging the GPU
                                if (lane == 0) atomicExch(&Q->pad[0], 1u);
                                break;
                            }
                        } else {
                            mi_watch_begin(wd, dn);  // (helped another board meanwhile: the clock starts again)
                        }
                        dn = mi_ld_u32(&b->done);
                    }
                    const unsigned long long tkm0 = MI_CLK();
                    mg = mi_merge(res, (long long)ro, (int)nch, lane);
                    MI_TICK(11, tkm0);
                }
                ev += mg.ev;
                if (mg.stop) {
                    stopped = true;
                    r_stat = mg.stat;
                    r_p = mg.p;
                    r_pow = mg.pow;
                    nt = mg.nt;
                } else if (mg.p != -2.0 && mg.p >= best_p) {
                    best_p = mg.p;
                    best_stat = mg.stat;
                    best_g = mg.g;
                    best_df = mg.df;
                }
                next += W;
                width *= 8ull;
            }
            {
                const unsigned long long tk3 = MI_CLK();
                tk_board += tk3 - tk2;
                if (lane == 0) {
                    x.r_more1 += (unsigned int)(tk3 - tk2);
                    if (next > (elim_full ? 0ull : (N < seq ? N : seq))) x.c_eval_short += 1ull;  // jobs that went to a board
                }
            }
            if (!stopped) {  // every subset significant: the maximum-p result (tests.jl:338-345)
                r_stat = best_stat;
                r_p = best_p < 0.0 ? 0.0 : best_p;
                r_pow = 1;
                nt = N;
            }
            const unsigned long long tkc0 = MI_CLK();
            x.c_ref += nt;
            x.c_calls += 1ull;
            x.c_eval += ev;
            x.c_alg += (P.max_k <= 3 && a <= FW_UNRANK32_A) ? dh_alg_bytes_disc32(a, ev, P.max_k, P.disc_bytes_per_col)
                                                            : dh_alg_bytes(a, ev, P.max_k, P.disc_bytes_per_col);
            dh_commit(x, A, lane, 1, r_stat, r_p, r_pow, P.alpha);
            MI_TICK(8, tkc0);
        }
        if (lane == 0) {
            x.r_more0 = (unsigned int)MI_CLK();
            tg[t] = x;
            atomicAdd(&Q->targets_done, 1u);  // (a termination count, not a hand-off: the host reads tg after the kernel)
        }
    }
    // no targets left to start: work on boards until every target has finished
    const unsigned long long tk_t0 = MI_CLK();
    MiWatch wd;  // progress of the launch as seen from here: a target finished or a board was opened (both counters only grow)
    mi_watch_begin(wd, 0u);
    unsigned int wd_periods = 0u;
    while (mi_ld_u32(&Q->targets_done) < (unsigned int)ntg && mi_ld_u32(&Q->pad[0]) == 0u) {
        if (!mi_help<L, NXY, PRE, R4>(Q, boards, res, bacc, lane)) {
            mi_nap_tail();
            wd.n |= 255u;  // (a nap is ~7 us: look at the clock every time)
            if (mi_watch_expired(wd, mi_ld_u32(&Q->targets_done) + mi_ld_u32(&Q->n_boards))) {
                wd.t0 = 0ull;
                if (++wd_periods >= 5u) {  // 10 s in which no target finished and no board was opened
                    if (lane == 0) atomicExch(&Q->pad[0], 2u);
                    break;
                }
            }
        } else {
            wd_periods = 0u;
            wd.t0 = 0ull;
        }
    }
    tk_tail = MI_CLK() - tk_t0;
    if (lane == 0) {  // t_body: first tests of own jobs, t_ctl: board phases, t_sleep: helping before jobs, n_seg: tail; total in pad[1] units of 2^10 ticks
        atomicAdd(&Q->t_body, tk_seq);
        atomicAdd(&Q->t_ctl, tk_board);
        atomicAdd(&Q->t_sleep, tk_help);
        atomicAdd(&Q->n_seg, tk_tail);
        atomicAdd(&Q->t_total, tk_t0 - tk_begin);
#ifdef FW_MI_TICKS
        for (int q = 0; q < 12; ++q) atomicAdd(&Q->tick[q], dh_mi_ticks[threadIdx.x >> 6][q]);
#endif
    }
}

// One wavefront per target: the lanes merge the job's segment records in parallel, then run the sequential part
// (commit, advance, next job) in lock-step -- every lane holds the same copy of the state, lane 0 writes.
//
// Elimination-phase look-ahead (spec_depth > 0, FW_ELIM_FULL windows): the phase tests the members c_0, c_1, ... of TPC one
// after the other, each against the pool without itself, and nearly always keeps them (cfg3: 96 %), so a target with
// 170 members is a chain of 170 rounds of one job each.  With the job of c_0 (pool L_0) the launch therefore also
// carries the jobs of c_1 .. c_d against the pools they will see IF every earlier member is kept:
// L_{j+1} = (L_j + [c_j]) \ {c_{j+1}} (hiton.jl:134-149: a kept member re-enters the pool at its end), each in its own
// accepted-list buffer.  The step kernel commits them in order; the first dropped member ends the chain (the later
// look-ahead jobs saw a pool that still held it: their results are discarded and they run again).
__device__ __forceinline__ void dh_step_dev(DhTgt *__restrict__ tg, int ntg, DhGlobal *__restrict__ g, const DhArrays &A,
                                            const FwSegOut *__restrict__ so, const long long *__restrict__ seg0,
                                            unsigned long long *__restrict__ win, unsigned int *__restrict__ sp,
                                            unsigned long long *__restrict__ win2, const int32_t *__restrict__ act, const DhParams &P,
                                            const int ci /* position in the list of unfinished targets (seg0 is indexed by it) */)
{
    const int lane = threadIdx.x & 63;
    const int d1 = P.spec_depth + 1;
    unsigned long long mywin = 0ull;
    if (ci < (int)g->n_act) {
        const int t = act[(size_t)g->act_sel * ntg + ci];
        DhTgt x = tg[t];
        const long long jseg0 = seg0[ci];
        const int nsp_done = x.nsp;
        const int spmode_done = x.spmode;
        // records of the target's own job, then of each look-ahead job (their window jwin2 can differ from the job's)
        const int nseg_all = (int)(seg0[ci + 1] - jseg0);
        int jnseg = nseg_all / (1 + nsp_done), jnseg2 = jnseg;
        if (nsp_done > 0 && spmode_done == 1) {
            const unsigned long long sl = g->seglen;  // the finished launch's segment length (the next plan has not run yet)
            jnseg = (int)dh_ceil_div(x.jwin, sl, 1.0 / (double)sl);
            jnseg2 = (nseg_all - jnseg) / nsp_done;
        }
        x.nsp = 0;
        x.spmode = 0;
        bool finished = false, kept = false;
        if (x.jactive) {
            if (x.phase == 0) (x.jnext == 0ull ? x.r_first0 : x.r_more0) += 1u;
            else (x.jnext == 0ull ? x.r_first1 : x.r_more1) += 1u;
            const DhMerge M = dh_merge(so, jseg0, jnseg, lane);
            double r_stat = 0.0, r_p = 1.0;
            int r_pow = 0;
            unsigned long long r_nt = 0ull;
            bool done = false;
            if (M.stop) {
                r_stat = M.stat;
                r_p = M.p;
                r_pow = M.pow;
                r_nt = M.nt;
                done = true;
            } else if (M.p != -2.0 && M.p >= x.jbest_p) {
                x.jbest_p = M.p;
                x.jbest_stat = M.stat;
            }
            x.jevaluated += M.ev;
            if (!done) {
                x.jnext += x.jwin;
                const unsigned long long growth = g->launched_ranks < P.small_launch ? P.growth_small : (g->n_live_prev > P.busy_jobs ? P.growth_busy : P.growth);
                x.jwidth *= growth;
                if (x.jnext >= x.jN) {
                    r_stat = x.jbest_stat;
                    r_p = x.jbest_p < 0.0 ? 0.0 : x.jbest_p;
                    r_pow = 1;
                    r_nt = x.jN;
                    done = true;
                }
            }
            if (done) {
                x.jactive = 0;
                finished = true;
                x.c_ref += r_nt;
                x.c_calls += 1ull;
                x.c_eval += x.jevaluated;
                x.c_eval_short += x.na <= FW_HK_A ? x.jevaluated : 0ull;
                x.na_max = x.na > x.na_max ? x.na : x.na_max;
                x.c_alg += dh_alg_bytes(x.na, x.jevaluated, P.max_k, P.disc_bytes_per_col);
                kept = dh_commit(x, A, lane, d1, r_stat, r_p, r_pow, P.alpha);
            }
        }
        if (nsp_done > 0) {
            // look-ahead jobs, committed in candidate order while the assumption they were built on holds: elimination
            // phase -- every earlier member kept (own pool buffers, whole enumerations); interleaving phase -- every
            // earlier candidate dropped (same accepted list, first window only: a candidate that survives its first
            // window becomes the target's job and ends the chain)
            // ... interleaving phase, spmode 1 (the job was in its LAST window, i.e. it had survived its first one and was about
            // to be accepted): first windows of the next candidates against accepted + [candidate]; they count if the candidate
            // was indeed accepted, and then follow the same in-order rule
            const bool ph1 = x.phase == 1, acc_mode = spmode_done == 1;
            const int n = acc_mode ? x.na : x.na - (kept ? 1 : 0), cur0 = x.cur;
            const unsigned long long wsp = acc_mode ? x.jwin2 : x.jwin;  // look-ahead jobs ride with the first window of the target's job
            if (acc_mode) x.jN = x.jN2;  // the enumeration the look-ahead jobs belong to (accepted list one entry longer)
            bool valid = finished && ((ph1 || acc_mode) ? kept : !kept);
            for (int j = 1; j <= nsp_done; ++j) {
                const DhMerge M = dh_merge(so, jseg0 + (long long)jnseg + (long long)(j - 1) * jnseg2, jnseg2, lane);
                if (!valid) {  // built on an assumption that failed: executed for nothing
                    x.c_eval += M.ev;
                    continue;
                }
                if (ph1) {
                    x.cur = (cur0 + j) % d1;
                    x.na = n;
                }
                if (!M.stop && wsp < x.jN) {  // interleaving: survived the first window -> continues as the target's job
                    const unsigned long long growth = g->launched_ranks < P.small_launch ? P.growth_small : (g->n_live_prev > P.busy_jobs ? P.growth_busy : P.growth);
                    x.jactive = 1;
                    x.jnext = wsp;
                    x.jwidth = wsp * growth;
                    x.jbest_p = M.p != -2.0 ? M.p : -1.0;
                    x.jbest_stat = M.p != -2.0 ? M.stat : 0.0;
                    x.jevaluated = M.ev;
                    valid = false;
                    continue;
                }
                const double r_stat = M.stop ? M.stat : (M.p != -2.0 ? M.stat : 0.0);
                const double r_p = M.stop ? M.p : (M.p < 0.0 ? 0.0 : M.p);
                x.c_ref += M.stop ? M.nt : x.jN;
                x.c_calls += 1ull;
                x.c_eval += M.ev;
                x.c_eval_short += n <= FW_HK_A ? M.ev : 0ull;
                x.c_alg += dh_alg_bytes(n, M.ev, P.max_k, P.disc_bytes_per_col);
                const bool k = dh_commit(x, A, lane, d1, r_stat, r_p, M.stop ? M.pow : 1, P.alpha);
                valid = ph1 ? k : !k;
            }
        }
        if (!x.jactive && x.phase != 2 && dh_advance(x, A, lane, d1)) {
            const unsigned long long N = dh_enum_size(x.na, P.max_k, P.max_tests);  // (32-bit binomials where every term fits)
            x.jN = N;
            x.jnext = 0ull;
            // Elimination phase: the candidate passed every test against (almost) this pool a moment ago, so nearly all
            // of these jobs run to the end -- evaluate the whole enumeration in one window instead of two rounds
            // (FW_ELIM_FULL=0 disables; cfg3: -20 % rounds for +2 % evaluated tests)
            x.jwidth = (x.phase == 1 && P.elim_full) ? N : (x.na >= 64 ? P.w0_big : P.w0_small);
            x.jbest_p = -1.0;
            x.jbest_stat = 0.0;
            x.jevaluated = 0ull;
            x.jactive = 1;
            if (lane == 0 && (unsigned int)x.na > g->max_a) atomicMax(&g->max_a, (unsigned int)x.na);  // rare: only on a new maximum
            {   // phase 0: the list can still take every whitelisted neighbour it has not met; phase 1: pools never exceed TPC
                const int abi = x.phase == 0 ? x.na + x.wl_n - x.wl_used : (x.na > x.ntpc ? x.na : x.ntpc);
                if (lane == 0 && (unsigned int)abi > g->max_ab) atomicMax(&g->max_ab, (unsigned int)abi);
            }
            if (x.phase == 0 && P.spec0_depth > 0 && g->launched_ranks < P.spec0_below && g->n_live_prev < P.spec0_jobs) {
                const int32_t *cands = A.cand0 + x.cand_off;
                int q = 0;
                // (light launches -- the first feed-forward rounds, a rank of a multi-GPU job: the tests are nearly free, the dependent
                // rounds are the cost -> more candidates per round)
                const int depth = g->launched_ranks < P.spec0_light_below ? P.spec0_depth_light : P.spec0_depth;
                while (q < depth && x.pos + 1 + q < x.nc) {
                    if (x.wl_n > 0 && dh_in_wl(x, A, cands[x.pos + 1 + q])) break;  // whitelisted: joins without a test
                    ++q;
                }
                x.nsp = q;
            }
            if (x.phase == 1 && P.elim_full && P.spec_depth > 0 && g->launched_ranks < P.spec_below) {
                // pools of the next members (see above), all built in ONE pass over the current pool:
                // L_j = (L_0 without c_1 .. c_j, in L_0's order) + [c_0, ..., c_{j-1}]
                const int32_t *cands = A.tpc_key + x.co;
                const int n = x.na;
                const int32_t c0 = cands[x.pos];
                int32_t cn[DH_MAX_SPEC];
                int Q = 0;
#pragma unroll
                for (int j = 0; j < DH_MAX_SPEC; ++j) {
                    cn[j] = -1;
                    if (j == Q && j < P.spec_depth && x.pos + 1 + j < x.nc) {
                        const int32_t c = cands[x.pos + 1 + j];
                        if (!(x.wl_n > 0 && dh_in_wl(x, A, c))) {  // whitelisted: kept without a test (hiton.jl:20-30)
                            cn[j] = c;
                            Q = j + 1;
                        }
                    }
                }
                const int32_t *src = A.acc + DH_ACC_OFF(x, x.cur, d1);
                int w[DH_MAX_SPEC];
#pragma unroll
                for (int j = 0; j < DH_MAX_SPEC; ++j) w[j] = 0;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                for (int base = 0; base < n; base += 64) {
                    const int i = base + lane;
                    const int32_t v = i < n ? src[i] : -1;
                    bool alive = i < n;
#pragma unroll
                    for (int j = 0; j < DH_MAX_SPEC; ++j) {
                        if (j < Q) {  // wavefront-uniform
                            alive = alive && v != cn[j];
                            const unsigned long long m = __ballot(alive);
                            int32_t *dst = A.acc + DH_ACC_OFF(x, (x.cur + 1 + j) % d1, d1);
                            if (alive) dst[w[j] + __popcll(m & ((1ull << lane) - 1ull))] = v;
                            w[j] += __popcll(m);
                        }
                    }
                }
                int q = 0;
#pragma unroll
                for (int j = 0; j < DH_MAX_SPEC; ++j) {
                    // a pool that lost anything but exactly c_1 .. c_{j+1} (duplicated entries: whitelists) ends the chain
                    if (j == q && j < Q && w[j] == n - (j + 1)) {
                        int32_t *dst = A.acc + DH_ACC_OFF(x, (x.cur + 1 + j) % d1, d1);
                        if (lane <= j) {
                            int32_t tail = c0;
#pragma unroll
                            for (int u = 0; u < DH_MAX_SPEC; ++u)
                                if (lane == u + 1) tail = cn[u];
                            dst[w[j] + lane] = tail;
                        }
                        q = j + 1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                x.nsp = q;
            }
        }
        if (x.jactive) {
            const unsigned long long left = x.jN - x.jnext;
            mywin = x.jwidth < left ? x.jwidth : left;
        }
        x.jwin = mywin;
        if (x.nsp > 0) x.jwin2 = mywin, x.jN2 = x.jN;  // spmode 0: the look-ahead jobs share the job's window
        if (x.jactive && x.phase == 0 && x.jnext > 0ull && mywin == x.jN - x.jnext && x.nsp == 0 && P.spec1_depth > 0 &&
            g->launched_ranks < P.spec0_below && g->n_live_prev < P.spec0_jobs) {
            // The candidate survived its first window(s) and this launch finishes its enumeration: it is about to be accepted
            // (cfg3: a heavy target accepts half of its candidates, and each of them cost a first window, a last window and only
            // then the next candidate -- the longest chain of the pass).  The first windows of the next candidates ride along
            // against accepted + [candidate]: the entry is written behind the list now (dh_commit writes the same value there)
            const int32_t *cands = A.cand0 + x.cand_off;
            int q = 0;
            while (q < P.spec1_depth && x.pos + 1 + q < x.nc) {
                if (x.wl_n > 0 && dh_in_wl(x, A, cands[x.pos + 1 + q])) break;  // whitelisted: joins without a test
                ++q;
            }
            if (q > 0) {
                if (lane == 0) A.acc[DH_ACC_OFF(x, x.cur, d1) + x.na] = cands[x.pos];
                const unsigned long long N2 = dh_enum_size(x.na + 1, P.max_k, P.max_tests);
                const unsigned long long w0 = x.na + 1 >= 64 ? P.w0_big : P.w0_small;
                x.nsp = q;
                x.spmode = 1;
                x.jN2 = N2;
                x.jwin2 = w0 < N2 ? w0 : N2;
                if (lane == 0 && (unsigned int)(x.na + 1) > g->max_a) atomicMax(&g->max_a, (unsigned int)(x.na + 1));
            }
        }
        if (lane == 0) {
            tg[t] = x;
            win[t] = mywin;  // 0 = no job in the coming launch
            sp[t] = (unsigned int)x.nsp;
            win2[t] = x.nsp > 0 ? x.jwin2 : 0ull;
        }
    }
}

__global__ __launch_bounds__(256) void dh_step_kernel(DhTgt *__restrict__ tg, int ntg, DhGlobal *__restrict__ g, DhArrays A,
                                                      const FwSegOut *__restrict__ so, const long long *__restrict__ seg0,
                                                      unsigned long long *__restrict__ win, unsigned int *__restrict__ sp,
                                                      unsigned long long *__restrict__ win2, const int32_t *__restrict__ act, DhParams P)
{
    dh_step_dev(tg, ntg, g, A, so, seg0, win, sp, win2, act, P, (int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
}

// one workgroup: totals of the coming launch, its segment length, per-target segment counts and their exclusive scan.
// The segment length has no upper cap here, so the launch never holds more than seg_target + (live jobs) segments:
// the fixed grid (seg_target + targets) always covers it.
#define DH_PER 16  // targets per planning thread held in registers (more targets: extra passes over global memory)
struct DhPlanArgs {
    unsigned int seg_target, seg_q, seg_min, log_cap;
    ulonglong2 *log;
    unsigned long long seg_a, seg_b;
};
// NT = threads of the one workgroup that plans (1024: dh_plan_kernel)
// (the last workgroup of dh_step_kernel)
template <int NT>
__device__ __forceinline__ void dh_plan_dev(int ntg, DhGlobal *__restrict__ g, const unsigned long long *__restrict__ win,
                                            const unsigned int *__restrict__ sp, const unsigned long long *__restrict__ win2,
                                            const int32_t *__restrict__ act_all, long long *__restrict__ seg0, const DhPlanArgs PA)
{
    constexpr int NW = NT / 64;
    const unsigned int seg_target = PA.seg_target, seg_q = PA.seg_q, seg_min = PA.seg_min, log_cap = PA.log_cap;
    ulonglong2 *__restrict__ log = PA.log;
    const unsigned long long seg_a = PA.seg_a, seg_b = PA.seg_b;
    __shared__ unsigned long long s_tot[16];
    __shared__ unsigned int s_live[16], s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = (int)g->n_act;  // unfinished targets; everything below is indexed by the position in that list
    const int32_t *act = act_all + (size_t)g->act_sel * ntg;
    const int per = (na + NT - 1) / NT;
    const int b = tid * per, e = (b + per) < na ? (b + per) : na;
    unsigned long long tot = 0ull;
    unsigned int live = 0u;
    unsigned long long wr[DH_PER];   // this thread's windows (registers when per <= DH_PER)
    unsigned long long w2r[DH_PER];  // window of each look-ahead job (win2: equal to the job's own except in spmode 1)
    unsigned int mr[DH_PER];         // look-ahead jobs
#pragma unroll
    for (int q = 0; q < DH_PER; ++q) {
        const int tq = (b + q < e) ? act[b + q] : 0;
        wr[q] = (b + q < e) ? win[tq] : 0ull;
        mr[q] = (b + q < e) ? sp[tq] : 0u;
        w2r[q] = (b + q < e && mr[q] > 0u) ? win2[tq] : 0ull;
    }
#pragma unroll
    for (int q = 0; q < DH_PER; ++q) {
        tot += wr[q] + w2r[q] * mr[q];
        live += wr[q] != 0ull ? 1u + mr[q] : 0u;
    }
    for (int t = b + DH_PER; t < e; ++t) {
        const unsigned long long w = win[act[t]];
        const unsigned int m = sp[act[t]];
        tot += w + (m ? win2[act[t]] * m : 0ull);
        live += w != 0ull ? 1u + m : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tot += __shfl_xor(tot, o);
        live += __shfl_xor(live, o);
    }
    if (lane == 0) {
        s_tot[wave] = tot;
        s_live[wave] = live;
    }
    __syncthreads();
    unsigned long long total = 0ull;
    unsigned int n_live = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        total += s_tot[w];
        n_live += s_live[w];
    }
    // small launches: fewer, longer segments (one / two residency waves of workgroups instead of three) -- every
    // workgroup pays its set-up (accepted list, thresholds, LDS table) once, and a launch of a few M ranks is latency-bound
    const unsigned int tgt = total < seg_a ? (seg_target + 2u) / 3u : (total < seg_b ? (2u * seg_target + 2u) / 3u : seg_target);
    unsigned long long seglen = (total / tgt + seg_q - 1ull) / seg_q * seg_q;
    seglen = seglen < seg_min ? seg_min : seglen;
    const double inv = 1.0 / (double)seglen;
    unsigned int local = 0u;
    unsigned int nr[DH_PER];
#pragma unroll
    for (int q = 0; q < DH_PER; ++q) {
        nr[q] = dh_ceil_div(wr[q], seglen, inv) + dh_ceil_div(w2r[q], seglen, inv) * mr[q];
        local += nr[q];
    }
    for (int t = b + DH_PER; t < e; ++t)
        local += dh_ceil_div(win[act[t]], seglen, inv) + (sp[act[t]] ? dh_ceil_div(win2[act[t]], seglen, inv) * sp[act[t]] : 0u);
    unsigned int incl = local;  // inclusive scan inside the wavefront, then over the 16 wavefront totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    unsigned int wbase = 0u, ns = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wave) wbase += s_wsum[w];
        ns += s_wsum[w];
    }
    unsigned int run = wbase + incl - local;
#pragma unroll
    for (int q = 0; q < DH_PER; ++q)
        if (b + q < e) {
            seg0[b + q] = (long long)run;
            run += nr[q];
        }
    for (int t = b + DH_PER; t < e; ++t) {
        seg0[t] = (long long)run;
        run += dh_ceil_div(win[act[t]], seglen, inv) + (sp[act[t]] ? dh_ceil_div(win2[act[t]], seglen, inv) * sp[act[t]] : 0u);
    }
    if (tid == 0) {
        seg0[na] = (long long)ns;
        g->ns = ns;
        g->any_big = 0u;
        g->seglen = (unsigned int)seglen;
        g->launched_ranks = total;
        g->n_live_prev = n_live;
        if (n_live == 0u) g->done = 1u;
        const unsigned int r = g->rounds;
        g->ns_ring[r & 63u] = ns;
        g->rounds = r + 1u;
        if (log && r < log_cap) log[r] = make_ulonglong2(total, ((unsigned long long)n_live << 32) | ns);  // FW_DH_LOG
    }
}

__global__ __launch_bounds__(1024) void dh_plan_kernel(int ntg, DhGlobal *__restrict__ g, const unsigned long long *__restrict__ win,
                                                       const unsigned int *__restrict__ sp, const unsigned long long *__restrict__ win2,
                                                       const int32_t *__restrict__ act_all, long long *__restrict__ seg0, DhPlanArgs PA)
{
    dh_plan_dev<1024>(ntg, g, win, sp, win2, act_all, seg0, PA);
}
// The same plan on ONE wavefront per SIMD (r06).  The 1 024-thread form is 4 wavefronts x 127 registers per SIMD: it starts only on a CU that is EMPTY, and while the
// other chain's long-list segment kernel (3 x 168 registers per SIMD, thousands of workgroups pending) keeps every CU refilled that happens when that kernel has
// nothing left to dispatch -- cfg5: 3.1 ms per call, 7 588 calls (profiles/r05_cfg5_kernel_stats.csv; stream priority does not help: r06_cfg5_plan_kernel.txt).
// 256 threads fit beside two resident segment wavefronts as soon as one workgroup leaves.  cfg5 59.1 -> 57.8 s; and at cfg3, whose size-3 kernel fills the register
// file with 4 x 128, the two chains stop holding each other's rounds up: plan 27 -> 8 us per call in the two-chain pass, headline 159.6 -> 152.2 ms
// (profiles/r06_cfg5_plan_kernel.txt, r06_cfg3_plan_kernel.txt).  The default (FW_DH_PLAN_SMALL=0: the 1 024-thread form).
__global__ __launch_bounds__(256) void dh_plan_small_kernel(int ntg, DhGlobal *__restrict__ g, const unsigned long long *__restrict__ win,
                                                            const unsigned int *__restrict__ sp, const unsigned long long *__restrict__ win2,
                                                            const int32_t *__restrict__ act_all, long long *__restrict__ seg0, DhPlanArgs PA)
{
    dh_plan_dev<256>(ntg, g, win, sp, win2, act_all, seg0, PA);
}

// fz rounds (r06): the local correlation matrix of every target that has one -- M[a][b] = cor[id(a)][id(b)], id(0) = T, id(1 ..) = T's
// level-0 neighbours in ascending id order (every variable a job of T can name: candidates, TPC / PC members and whitelisted
// neighbours are all level-0 neighbours of T).  The size-3 test gathers one matrix entry cor[z3][z2] per test, 64 scattered 4-byte
// words per wavefront step out of a 400 MB matrix: L2 hit 88 %, i.e. every step waited for a fabric round trip and the pass read
// 105 GB through the fabric (profiles/r05_cfg3_pmc_summary.json) -- out of a target's own (deg + 1)^2 floats (cfg3's heaviest: 250 KB)
// the same gathers stay in L2.  One workgroup per target, rows a, columns b over the threads.
__global__ __launch_bounds__(256) void dh_tmat_build_kernel(const DhTgt *__restrict__ tg, int ntg, const int32_t *__restrict__ nb_idx,
                                                            const float *__restrict__ cor, int p, float *__restrict__ tmat)
{
    __shared__ int32_t s_id[4096];
    const DhTgt &x = tg[blockIdx.x];
    if (x.tm_off < 0) return;  // (workgroup-uniform)
    const int m = x.nb_n + 1;
    const int32_t *ids = nb_idx + x.nb_off;
    for (int b = threadIdx.x; b < m; b += 256) s_id[b] = b == 0 ? x.T : ids[b - 1];
    __syncthreads();
    float *M = tmat + x.tm_off;
    // rows blockIdx.y, blockIdx.y + gridDim.y, ...: four rows in flight per thread (independent gathers)
    for (int a0 = (int)blockIdx.y * 4; a0 < m; a0 += (int)gridDim.y * 4)
        for (int b = threadIdx.x; b < m; b += 256) {
            const size_t col = (size_t)s_id[b];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a0 + q < m ? cor[(size_t)s_id[a0 + q] * (size_t)p + col] : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (a0 + q < m) M[(size_t)(a0 + q) * m + b] = v[q];
        }
}

// segment record s of the coming launch (one thread)
__device__ __forceinline__ void dh_fill_one(const unsigned int s, const DhTgt *__restrict__ tg, int ntg, DhGlobal *__restrict__ g,
                                            const long long *__restrict__ seg0, const DhArrays &A, FwSeg *__restrict__ segs, int d1,
                                            const int32_t *__restrict__ act)
{
    int lo = 0, hi = (int)g->n_act;  // first list position with seg0 > s; the job owning slot s is the one before it
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (seg0[mid] <= (long long)s)
            lo = mid + 1;
        else
            hi = mid;
    }
    const DhTgt &x = tg[act[(size_t)g->act_sel * ntg + lo - 1]];
    const unsigned long long seglen = g->seglen;
    unsigned long long k = (unsigned long long)((long long)s - seg0[lo - 1]);
    const int32_t *cands = x.phase == 0 ? A.cand0 + x.cand_off : A.tpc_key + x.co;
    unsigned int slot = 0;  // 0 = the target's job, j = its j-th look-ahead job (own window jwin2; elimination: own pool buffer)
    if (x.nsp > 0) {
        const unsigned int all = (unsigned int)(seg0[lo] - seg0[lo - 1]);
        const unsigned int n0 = x.spmode == 1 ? dh_ceil_div(x.jwin, seglen, 1.0 / (double)seglen) : all / (1u + (unsigned int)x.nsp);
        if ((unsigned int)k >= n0) {
            const unsigned int per = (all - n0) / (unsigned int)x.nsp;
            slot = 1u + ((unsigned int)k - n0) / per;
            k -= (unsigned long long)n0 + (unsigned long long)(slot - 1u) * per;
        }
    }
    const bool acc_mode = slot > 0u && x.spmode == 1;  // list = accepted + [current candidate], first window of its own enumeration
    FwSeg sg;
    sg.X = x.T;
    sg.Y = cands[x.pos + (int)slot];
    sg.acc_off = DH_ACC_OFF(x, x.phase == 1 ? (x.cur + (int)slot) % d1 : x.cur, d1);
    sg.acc_len = x.na + (acc_mode ? 1 : 0);
    if (sg.acc_len > FW_TAB_A) g->any_big = 1u;  // same value from every writer
    sg.pad = act[(size_t)g->act_sel * ntg + lo - 1];  // fz_nz: the job's record slot = the target's index in the run (dh_nz_recs_kernel)
    const unsigned long long lo_r = acc_mode ? 0ull : x.jnext;
    sg.start = lo_r + k * seglen;
    const unsigned long long hi_r = lo_r + (slot > 0u ? x.jwin2 : x.jwin);
    sg.end = sg.start + seglen < hi_r ? sg.start + seglen : hi_r;
    const bool has_tm = A.tmat != nullptr && x.tm_off >= 0;
    sg.tm = has_tm ? (uint64_t)(A.tmat + x.tm_off) : 0ull;
    sg.tm_ids = has_tm ? (uint64_t)(A.nb_idx + x.nb_off) : 0ull;
    sg.tm_m = has_tm ? x.nb_n + 1 : 0;
    sg.pad1 = 0;
    segs[s] = sg;
}
__global__ __launch_bounds__(256) void dh_fill_kernel(const DhTgt *__restrict__ tg, int ntg, DhGlobal *__restrict__ g,
                                                      const long long *__restrict__ seg0, const DhArrays A,
                                                      FwSeg *__restrict__ segs, int d1, const int32_t *__restrict__ act)
{
    const unsigned int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= g->ns) return;
    dh_fill_one(s, tg, ntg, g, seg0, A, segs, d1, act);
}

// (r04 fused step + plan (+ fill) into the last workgroup to finish, r05 into a cooperative grid with one device-memory barrier and a
// redundant plan in every workgroup: both bit-identical, neither faster -- the barrier and the second plan cost what the saved
// launches cost, ~5 us of idle time per device-wide dependency either way.  profiles/r04_fused_round.json, profiles/r05_coop_round.json.)

// One workgroup, once per batch of rounds (between dh_step_kernel and dh_plan_kernel): drops the finished targets from
// the list the three kernels above walk.  Most targets finish early (cfg3: 10 000 targets, a few hundred alive for
// most of the rounds; cfg4: 50 020 / a few thousand), so step, plan and fill shrink with the work that is left.
// Order-preserving; writes the other half of the ping-pong list and flips act_sel.
__global__ __launch_bounds__(1024) void dh_compact_kernel(const DhTgt *__restrict__ tg, int ntg, DhGlobal *__restrict__ g,
                                                          int32_t *__restrict__ act_all)
{
    __shared__ unsigned int s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = (int)g->n_act;
    const unsigned int sel = g->act_sel;
    const int32_t *src = act_all + (size_t)sel * ntg;
    int32_t *dst = act_all + (size_t)(sel ^ 1u) * ntg;
    const int per = (na + 1023) / 1024;
    const int b = tid * per, e = (b + per) < na ? (b + per) : na;
    unsigned int local = 0u;
    for (int i = b; i < e; ++i) {
        const DhTgt &x = tg[src[i]];
        local += (x.phase != 2 || x.jactive) ? 1u : 0u;
    }
    unsigned int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    unsigned int wbase = 0u, total = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wave) wbase += s_wsum[w];
        total += s_wsum[w];
    }
    unsigned int pos = wbase + incl - local;
    for (int i = b; i < e; ++i) {
        const int t = src[i];
        const DhTgt &x = tg[t];
        if (x.phase != 2 || x.jactive) dst[pos++] = t;
    }
    if (tid == 0) {
        g->n_act = total;
        g->act_sel = sel ^ 1u;
    }
}

// fz_nz on the device rounds (r05): the per-job records of the sub-matrix kernel (fznz_submat_kernel: correlations over the rows where
// T and the candidate are both non-zero, statfuns.jl:138-155), one slot per target.  A job that enters its FIRST window gets a fresh
// record -- variables, list, the target's own arena slice ((candidates + 2)^2 floats: a list never outgrows the candidate list) -- and
// its matrix is computed in this round; a job in a later window keeps record and matrix (bit 0 of pad: nothing to compute), as does
// a target without a job.  (The host pool recomputes the matrix for every window of a job.)
__global__ __launch_bounds__(256) void dh_nz_recs_kernel(const DhTgt *__restrict__ tg, int ntg, const DhGlobal *__restrict__ g,
                                                         const int32_t *__restrict__ act, const DhArrays A, FwNzJob *__restrict__ recs,
                                                         const long long *__restrict__ arena_off)
{
    const int ci = (int)(blockIdx.x * 256u + threadIdx.x);
    if (ci >= (int)g->n_act) return;
    const int t = act[(size_t)g->act_sel * ntg + ci];
    const DhTgt &x = tg[t];
    FwNzJob *r = recs + t;
    if (x.jactive && x.jwin > 0ull && x.jnext == 0ull) {
        r->X = x.T;
        r->Y = (x.phase == 0 ? A.cand0 + x.cand_off : A.tpc_key + x.co)[x.pos];
        r->acc_off = DH_ACC_OFF(x, x.cur, 1);
        r->acc_len = x.na;
        r->m = x.na + 2;
        r->cor_off = arena_off[t];
        r->pad = 0;
    } else {
        r->pad = 1;
    }
}

// ---- the whole feed-forward schedule of the discrete kinds on the device (fwi_devhiton_mi_schedule) ------------------------------
// per-target state of EVERY target of the schedule, built on the device from the level-0 CSR: a target's arrays (candidates in
// hiton.jl:211-217 order, TPC / PC, accepted list, whitelist) all sit at its level-0 offset nb_off[T] with its degree as capacity
__global__ __launch_bounds__(256) void dh_mi_init_kernel(DhTgt *__restrict__ tg, int nt, const int32_t *__restrict__ sched,
                                                         const long long *__restrict__ nb_off, const int32_t *__restrict__ levels)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= nt) return;
    DhTgt x;
    __builtin_memset(&x, 0, sizeof(x));
    const int T = sched[i];
    const long long o = nb_off[T];
    const int deg = (int)(nb_off[T + 1] - o);
    x.T = T;
    x.nc = x.cap = deg;
    x.phase = (deg == 0 || levels[T] < 2) ? 2 : 0;  // hiton.jl:182-184 (a constant variable), :336-338 (no candidate)
    x.co = x.cand_off = x.wl_off = x.nb_off = o;
    x.nb_n = deg;
    x.wl_unsorted = 1;
    tg[i] = x;
}

// after a round: the running graph of interleaved.jl:136-140, kept only where a later round will read it -- target T with u in PC(T)
// is a whitelisted neighbour of u (hiton.jl:20-30) when u's own round comes later; T's own list is never read again.  One entry per
// (T, u): no duplicates.  The level-0 lists are symmetric and PC(T) is a subset of T's candidates, so u's list holds at most deg(u).
__global__ __launch_bounds__(256) void dh_wl_append_kernel(const DhTgt *__restrict__ tg, int ntg, const int32_t *__restrict__ pc_key,
                                                           const int32_t *__restrict__ round_of, int round, int32_t *__restrict__ wl,
                                                           const long long *__restrict__ nb_off, unsigned int *__restrict__ wl_cnt)
{
    const int t = (int)(blockIdx.x * 4u + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (t >= ntg) return;
    const int T = tg[t].T, npc = tg[t].npc;
    const long long co = tg[t].co;
    for (int i = lane; i < npc; i += 64) {
        const int32_t u = pc_key[co + i];
        if (round_of[u] > round) wl[nb_off[u] + (long long)atomicAdd(&wl_cnt[u], 1u)] = T;
    }
}

// results of the whole schedule, packed for ONE small download: every target's PC entries (target, neighbour, statistic, p) as a block
// of consecutive records in insertion order -- the blocks land where an atomic ticket puts them: the host's CSR over targets
// (fw_hiton.cpp: a stable counting sort by target) does not depend on the order of the blocks -- and the per-target counters summed
// (integers and multiples of 1/8: every order gives the same sums).  tot[0] entries, [1] tests in reference order, [2] jobs,
// [3] executed tests, [4] unfinished targets (must stay 0); alg: algorithmic bytes
__global__ __launch_bounds__(256) void dh_mi_pack_kernel(const DhTgt *__restrict__ tg, int nt, const int32_t *__restrict__ pc_key,
                                                         const double *__restrict__ pc_stat, const double *__restrict__ pc_p,
                                                         int32_t *__restrict__ o_t, int32_t *__restrict__ o_u, double *__restrict__ o_s,
                                                         double *__restrict__ o_p, unsigned long long *__restrict__ tot, double *__restrict__ alg)
{
    const int t = (int)(blockIdx.x * 4u + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (t >= nt) return;
    const DhTgt &x = tg[t];
    const int npc = x.npc;
    unsigned long long base = 0ull;
    if (lane == 0) {
        if (npc > 0) base = atomicAdd(&tot[0], (unsigned long long)npc);
        if (x.c_ref) atomicAdd(&tot[1], x.c_ref);
        if (x.c_calls) atomicAdd(&tot[2], x.c_calls);
        if (x.c_eval) atomicAdd(&tot[3], x.c_eval);
        if (x.phase != 2) atomicAdd(&tot[4], 1ull);
        if (x.c_alg != 0.0) atomicAdd(alg, x.c_alg);
    }
    base = mi_rfl64(base);
    for (int i = lane; i < npc; i += 64) {
        o_t[base + i] = x.T;
        o_u[base + i] = pc_key[x.co + i];
        o_s[base + i] = pc_stat[x.co + i];
        o_p[base + i] = pc_p[x.co + i];
    }
}

}  // namespace

// window / look-ahead / board policy of a run (the sweeps that chose the defaults are quoted where each value is set)
static DhParams dh_make_params(fw_ctx *c, int ntg, int spec_depth, int spec0_depth)
{
    DhParams P{};
    P.alpha = c->P.alpha;
    P.max_k = c->P.max_k;
    P.max_tests = c->P.max_tests;
    {
        const char *e = fw_knob("FW_SMALL_LAUNCH");
        P.small_launch = e ? (unsigned long long)atoll(e) : (1ull << 22);
        const char *w = fw_knob("FW_W0_BIG");
        P.w0_big = w ? (unsigned long long)atoll(w) : 16384ull;
    }
    {
        const char *e = fw_knob("FW_ELIM_FULL");
        P.elim_full = e ? atoi(e) : 1;
    }
    {
        auto envu = [](const char *n, unsigned long long d) { const char *e = getenv(n); return e && atoll(e) > 0 ? (unsigned long long)atoll(e) : d; };
        P.growth_small = envu("FW_DH_GROWTH_SMALL", 256ull);
        P.growth = envu("FW_DH_GROWTH", 4ull);  // cfg3 sweeps: r01 4 -> 322 ms, 8 -> 321.5, 16 -> 331, 32 -> 350; r02 (three chains) 2 -> 219.6, 4 -> 218.3, 8 -> 219.7, 16 -> 241 (the host pool uses 16: its rounds cost 3x more)
        P.growth_busy = envu("FW_DH_GROWTH_BUSY", 4ull);
        P.busy_jobs = (unsigned int)envu("FW_DH_BUSY_JOBS", 2048ull);
        P.spec_depth = spec_depth;
        P.spec_below = envu("FW_DH_SPEC_BELOW", (c->P.max_k <= 3 && ntg >= 256) ? 30000000ull : 12000000ull);  // chains of few targets (a rank of 4 / 8: 98 / 49 per chain) are latency-bound and pay for it: slowest of 8 ranks 83.9 -> 91.0 ms with 30 M;  // max_k 4-5: whole enumerations of the look-ahead pools are too dear in big launches (cfg5, first 40 000 targets: 0.97 -> 1.95 s with 30 M)  // (r03, after the job-count gate of the interleaving look-ahead moved: 12 M -> 199.8 ms, 20 M 193.6, 30 M 193.4, 50 M 194.7, none 193.6; cfg3 without feed-forward 172.1 -> 168.4)
        P.spec0_depth = spec0_depth;
        P.spec0_below = envu("FW_DH_SPEC0_BELOW", 12000000ull);
        { const char *e = fw_knob("FW_DH_SPEC0_LIGHT"); P.spec0_depth_light = spec0_depth > 0 ? std::min(std::max(e ? atoi(e) : spec0_depth, spec0_depth), DH_MAX_SPEC) : 0; }
        P.spec0_light_below = envu("FW_DH_SPEC0_LIGHT_BELOW", 400000ull);
        P.spec0_jobs = (unsigned int)envu("FW_DH_SPEC0_JOBS", 4096ull);  // (r03: 512 kept it off in the light feed-forward rounds of 1 024 targets: cfg3 204.8 -> 198.6 ms, 9 222 -> 7 966 launches)
        {   // look-ahead behind a candidate that is about to be accepted (dh_step_kernel, spmode 1)
            const char *e = fw_knob("FW_DH_SPEC1");
            P.spec1_depth = c->P.kind == FW_FZ ? std::min(std::max(e ? atoi(e) : 2, 0), DH_MAX_SPEC) : 0;
        }
        P.mi_seq = (unsigned int)envu("FW_MI_SEQ", 48ull);  // r04 sweep on the final kernels (cfg4, ms with / without feed-forward, two runs each): 8: 149 / 106, 16 (r02-r03): 111.0 / 72.5, 24: 118 / 71.8, 32: 102.4 / 70.9, 48: 102.4 / 70.5, 64: 102.2 / 70.8, 128: 112.9 / 72.6, never a board: 111.5 / 79.1; cfg2 neutral
        P.mi_win0 = (unsigned int)envu("FW_MI_WIN0", 128ull);
        P.mi_chunk_div = (unsigned int)envu("FW_MI_CHUNK_DIV", 256ull);
        P.mi_chunk_min = (unsigned int)envu("FW_MI_CHUNK_MIN", 8ull);
        P.mi_chunk_max = (unsigned int)envu("FW_MI_CHUNK_MAX", 64ull);
        { const char *e = fw_knob("FW_MI_HELP_JOBS"); P.mi_help_jobs = e ? (unsigned int)atoi(e) : 1u; }
        { const char *e = fw_knob("FW_MI_ELIM_MIN"); P.mi_elim_min = e ? (unsigned int)atoi(e) : 64u; }
        { const char *e = fw_knob("FW_MI_HEAVY"); P.mi_heavy = e ? (unsigned int)atoi(e) : 48u; }
        { const char *e = fw_knob("FW_MI_SEQ_HEAVY"); P.mi_seq_heavy = e ? (unsigned int)atoi(e) : P.mi_seq; }
        { const char *e = fw_knob("FW_MI_SEQ_TAIL"); P.mi_seq_tail = e ? (unsigned int)atoi(e) : 4u; }
        { const char *e = fw_knob("FW_MI_AHEAD"); P.mi_ahead = e ? (unsigned int)atoi(e) : 1u; }
        { const char *e = fw_knob("FW_MI_CHUNK_TAIL"); P.mi_chunk_tail = e ? (unsigned int)std::max(1, atoi(e)) : P.mi_chunk_min; }
        { const char *e = fw_knob("FW_MI_WIN0_TAIL"); P.mi_win0_tail = e ? (unsigned int)std::max(1, atoi(e)) : 1024u; }
    }
    const bool fz = c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ;  // (fz_nz: the same segment kernels on job-local matrices)
    // first window of an interleaving-phase job with fewer than 64 accepted variables.  r04 sweep at cfg3 (two chains, look-ahead 4 / 2,
    // growth 4; ms per pass / launches): 64: 198.7 / 9 790, 128: 194.8, 256 (r01-r03): 189.1 / 8 090, 512: 186.3, 1 024: 183.9, 2 048: 184.7,
    // 4 096: 182.5 / 7 194, 8 192: 184.3, 16 384: 187.6 -- the executed tests move by less than 1 % over that range (a job that stops does
    // so within its first wavefront steps whatever the window), the launches by 25 %.  max_k > 3 keeps 256 (not measured at full cfg5 size).
    // Second sweep (first window 4 096): first window of jobs with 64 accepted variables or more 8 192: 191.2, 16 384 (r01-r03): 183.8,
    // 24 576: 182.4, 32 768: 178.1 - 179.9, 49 152: 182.4, 65 536: 191.4; with 32 768 the small window 2 048: 177.3, 4 096: 178.1, 8 192: 181.7.
    P.w0_small = fz ? (c->P.max_k <= 3 ? 2048ull : 256ull) : 16ull;
    if (fz && c->P.max_k <= 3 && !fw_knob("FW_W0_BIG")) P.w0_big = 32768ull;
    if (fz) {
        const char *e = fw_knob("FW_W0_SMALL");  // first window (ranks) of a job with fewer than 64 accepted variables (r04 sweep: DESIGN section 4)
        if (e && atoll(e) > 0) P.w0_small = (unsigned long long)atoll(e);
    }
    if (!fz) P.w0_big = 16ull;
    P.seg_q = fz ? 256u : 4u;
    P.seg_min = fz ? 256u : 8u;
    P.disc_bytes_per_col = fz ? 0.0 : (double)c->P.n * (c->P.kind == FW_MI ? 1.0 : 2.0) / 8.0;
    return P;
}

static unsigned dh_team_min() { static const unsigned v = [] { const char *e = fw_knob("FW_MI_TEAM_MIN"); return e ? (unsigned)atoi(e) : 96u; }(); return v; }
static unsigned dh_team_max() { static const unsigned v = [] { const char *e = fw_knob("FW_MI_TEAM_MAX"); return e ? (unsigned)atoi(e) : 192u; }(); return v; }  // (64 / 256: cfg2 10.5 ms, cfg4 161.7; 128 / 128: 9.4, 162.7; 96 / 192: 9.3, 159.6)

// launches the persistent discrete kernel over targets tg[0 .. ntg) taken in the order d_order (heaviest first; the first `team` of
// them by a workgroup each); returns the grid.  The queue and the boards must be zero (the caller's memsets on the same stream).
static unsigned dh_mi_launch(fw_ctx *c, hipStream_t st, DhTgt *d_tg, int ntg, const int32_t *d_order, const DhArrays &A, DhParams P, unsigned team,
                             bool trace_host, MiQueue *d_mq, MiBoard *d_boards, FwSegOut *d_mres, int32_t *d_bacc)
{
    MiDev M = fwi_mi_dev(c);
    M.view = M.dense && M.nzmode && c->mi_view;  // HITON-PC under the dense rules tests on row views (hiton.jl:41-50)
    // as many workgroups as stay resident (one per CU: the test routine needs ~260 VGPRs, one wavefront per SIMD; cfg4:
    // 62 ms with one, 71 ms with two requested): the wavefronts fetch targets themselves
    static const unsigned wg_per_cu = [] { const char *e = fw_knob("FW_MI_WG_PER_CU"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : 1u; }();
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->P.device);
    const unsigned grid = std::min((unsigned)((ntg + 3) / 4), wg_per_cu * (unsigned)n_cu);
    static const unsigned team_steps = [] { const char *e = fw_knob("FW_MI_TEAM_STEPS"); return e ? (unsigned)atoi(e) : 2u; }();
    static const unsigned team_tail = [] { const char *e = fw_knob("FW_MI_TEAM_TAIL"); return e ? (unsigned)atoi(e) : 0u; }();
    P.mi_team = team;
    P.mi_team_steps = team_steps;
    P.mi_team_tail = team_tail;
#ifdef FW_MI_TICKS
    P.mi_trace = 1u;
#else
    P.mi_trace = trace_host ? 1u : 0u;
#endif
#define DH_MI_LAUNCH(LL, NN, PP, RR)                                                                                                  \
    hipLaunchKernelGGL((dh_mi_target_kernel<LL, NN, PP, RR>), dim3(grid), dim3(256), 0, st, d_tg, ntg, d_order, A, M, P, \
                       d_mq, d_boards, d_mres, d_bacc)
    const bool pre = c->P.n <= MI_PRE_N && c->P.max_k <= MI_PRE_K;
    // four subsets per wavefront step (mi_test_core4): n <= 5120, max_k <= 3, 2 x 2 cells per stratum.  FW_MI_ROW4=0: one per step
    static const bool row4_env = [] { const char *e = fw_knob("FW_MI_ROW4"); return !(e && atoi(e) == 0); }();
    // ... used up to 2048 samples (four words per lane): cfg2 (n = 500) 15.2 -> 12.5 ms.  At cfg4's n = 5000 a step of four
    // tests takes as long as four one-test steps (34 us: ten words per lane, 390 registers with the spills parked in AGPRs),
    // and since most jobs stop at their first test the three speculative ones are pure cost: measured 56.2 vs 56.8 ms on one
    // GPU, 35.3 vs 39.2 ms for one rank of eight -- FW_MI_ROW4=2 forces it on up to MI4_N for such experiments
    static const bool row4_force = [] { const char *e = fw_knob("FW_MI_ROW4"); return e && atoi(e) == 2; }();
    const bool r4 = row4_env && pre && c->P.n <= (row4_force ? MI4_N : 2048) && (c->L == 2 || c->mi_nxy == 2);
    const bool wide = c->P.n > 65535;  // cell counts beyond 16 bits: one count per register, 32-bit tables (mi_test_core<.., WIDE>)
    if (c->L == 2) {
        if (wide) DH_MI_LAUNCH(2, 2, 2, false); else if (r4) DH_MI_LAUNCH(2, 2, 1, true); else if (pre) DH_MI_LAUNCH(2, 2, 1, false); else DH_MI_LAUNCH(2, 2, 0, false);
    } else if (c->mi_nxy == 2) {
        if (wide) DH_MI_LAUNCH(3, 2, 2, false); else if (r4) DH_MI_LAUNCH(3, 2, 1, true); else if (pre) DH_MI_LAUNCH(3, 2, 1, false); else DH_MI_LAUNCH(3, 2, 0, false);
    } else {
        if (wide) DH_MI_LAUNCH(3, 3, 2, false); else if (pre) DH_MI_LAUNCH(3, 3, 1, false); else DH_MI_LAUNCH(3, 3, 0, false);
    }
#undef DH_MI_LAUNCH
    return grid;
}

static void dh_mi_trace(const MiQueue &hq, unsigned grid)
{
    const double nw = 4.0 * (double)grid, ms = 1e-5;
    if (hq.tick[1])
        fprintf(stderr, "[fw] test routine: %llu calls, %llu tests (mean set size %.2f); per test %.2f us before the core (unranking, list), %.2f us in the core, %.2f us in the accounting (Q)\n",
                hq.tick[0], hq.tick[1], (double)hq.tick[5] / (double)hq.tick[1], 1e-2 * hq.tick[2] / (double)hq.tick[1],
                1e-2 * hq.tick[3] / (double)hq.tick[1], 1e-2 * hq.tick[4] / (double)hq.tick[1]);
    if (hq.tick[1])
        fprintf(stderr, "[fw] state machine, ms per wavefront: dh_advance %.3f, job set-up %.3f, commit + counters %.3f, publish %.3f, idle wait for own board %.3f, merge %.3f; "
                        "test routine: before the core %.3f, core %.3f, accounting %.3f\n",
                ms * hq.tick[6] / nw, ms * hq.tick[7] / nw, ms * hq.tick[8] / nw, ms * hq.tick[9] / nw, ms * hq.tick[10] / nw, ms * hq.tick[11] / nw,
                ms * hq.tick[2] / nw, ms * hq.tick[3] / nw, ms * hq.tick[4] / nw);
    if (hq.tm_steps)
        fprintf(stderr, "[fw] team rounds: %llu wavefront-rounds, %llu tests in them; per wavefront-round %.2f us in the test routine, %.2f us at the barrier\n",
                hq.tm_steps, hq.tm_tests, 1e-2 * hq.tm_run / (double)hq.tm_steps, 1e-2 * hq.tm_wait / (double)hq.tm_steps);
    fprintf(stderr, "[fw] boards %u records %u; per wavefront (%u wavefronts): until out of targets %.2f ms = own first tests %.2f + board phases %.2f + helping before jobs %.2f + state machine %.2f; tail %.2f ms\n",
            hq.n_boards, hq.res_top, 4u * grid, ms * hq.t_total / nw, ms * hq.t_body / nw, ms * hq.t_ctl / nw, ms * hq.t_sleep / nw,
            ms * ((double)hq.t_total - (double)hq.t_body - (double)hq.t_ctl - (double)hq.t_sleep) / nw, ms * hq.n_seg / nw);
}

// One round of targets on the device.  in: T ids, interleaving candidates and (sorted) whitelists per target;
// out: PC (keys, statistics, p-values) per target in insertion order.
int fwi_devhiton_run(fw_ctx *c, const std::vector<FwDhTarget> &in, std::vector<FwDhResult> &out, FwDhFlat &flat, int chain)
{
    const int ntg = (int)in.size();
    out.assign((size_t)ntg, FwDhResult{});
    if (ntg == 0) return FW_OK;
    static const bool trace_host = fw_knob("FW_TRACE_HOST") != nullptr;  // host-side phase times on stderr
    auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double th0 = wall();
    // chain > 0: a second (third, ...) instance running concurrently from its own host thread on its own stream / arena
    if (chain > 0 && !c->dh_stream[chain]) FW_HIP(c, hipStreamCreateWithFlags(&c->dh_stream[chain], hipStreamNonBlocking));
    hipStream_t st = chain == 0 ? c->pb[0].stream : c->dh_stream[chain];
    const int p = c->P.p;
    // ---- host-side layout ----
    const bool use_devc = in[0].nc_dev >= 0 && c->d_cand != nullptr;  // candidate lists built on the device (fw_bh.hip)
    std::vector<DhTgt> tg((size_t)ntg);
    std::vector<int32_t> cand0, wl;
    long long co = 0, wo = 0;
    int max_cap = 0, max_wl = 0;  // most candidates / most whitelisted neighbours of one target
    unsigned max_a_seen = 0;      // longest accepted list reported so far (lags by up to two batches)
    unsigned max_ab_seen = 0;     // ... and the largest accepted + whitelisted-to-come
    for (int t = 0; t < ntg; ++t) {
        DhTgt x{};
        x.T = in[t].T;
        x.nc = use_devc ? in[t].nc_dev : (int32_t)in[t].cands.size();
        x.cap = x.nc;
        x.cand_off = use_devc ? c->nb_off[in[t].T] : co;
        x.phase = x.nc == 0 ? 2 : 0;
        x.co = co;
        x.wl_off = wo;
        x.wl_n = in[t].wl_n;
        x.nb_off = c->nb_off[x.T];
        x.nb_n = (int32_t)(c->nb_off[x.T + 1] - c->nb_off[x.T]);
        if (!use_devc) cand0.insert(cand0.end(), in[t].cands.begin(), in[t].cands.end());
        if (in[t].wl_n) wl.insert(wl.end(), in[t].wl, in[t].wl + in[t].wl_n);
        co += x.nc;
        wo += in[t].wl_n;
        max_cap = std::max(max_cap, x.nc);
        max_wl = std::max(max_wl, (int)in[t].wl_n);
        tg[t] = x;
    }
    const size_t tot = (size_t)co;
    // fz rounds (r06): local correlation matrices for the targets whose jobs can be long (dh_tmat_build_kernel); FW_FZ_TMAT=0: none
    size_t tm_floats = 0;
    {
        static const int tm_env = [] { const char *e = fw_knob("FW_FZ_TMAT"); return e ? atoi(e) : -1; }();  // smallest degree that gets one (0: off)
        const int tm_min = tm_env >= 0 ? tm_env : (c->P.max_k > 3 ? 1 : 16);                                   // (cfg3 sweep: 16; cfg5: 1 -- 54.6 -> 53.8 s)
        const bool tm_on = c->P.kind == FW_FZ && c->P.max_k <= 5 && c->d_cor != nullptr && tm_min > 0;  // (max_k 6-7: the general-form kernel reads the p x p matrix)
        for (int t = 0; t < ntg; ++t) {
            tg[t].tm_off = -1;
            const size_t m = (size_t)tg[t].nb_n + 1;
            if (!tm_on || tg[t].nb_n < tm_min || m > 4096 || tm_floats + m * m > (size_t)1 << 32) continue;  // (<= 16 GB per chain; ids of a target staged in 16 KB of LDS)
            tg[t].tm_off = (long long)tm_floats;
            tm_floats += m * m;
        }
    }
    // ---- device buffers (one arena) ----
    const bool nb_on_dev = c->d_nb_idx != nullptr;
    const size_t nnz = (size_t)c->nb_off[p];
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    static const unsigned seg_target_env = [] { const char *e = fw_knob("FW_SEG_TARGET"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : 0u; }();
    const unsigned seg_target = seg_target_env ? seg_target_env : ((c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) ? (c->P.max_k > 3 ? 8192u : 3072u) : 4096u);  // cfg3 sweep: 3072; cfg5 (max_k 5, launches of 10^8 ranks and more): 8192 (r06: 53.9 -> 52.4 s)
    // elimination-phase look-ahead (fz, FW_ELIM_FULL windows; see dh_step_kernel): FW_DH_SPEC = members tested ahead per
    // target, FW_DH_SPEC_BELOW = only while the last launch held fewer ranks than this.  cfg3 sweep (ms per pass, one
    // GPU / one rank of 8): off 326.7 / 112.0; depth 4 always 338 / 103; depth 4 below 4M 318.5 / 104.4, below 8M
    // 317.3 / 104.6, below 12M 314.8 / 103.6, below 16M 341 -- a launch that already fills the GPU only pays for the
    // jobs wasted behind every dropped member (3.7 % of the members at cfg3)
    static const int spec_env = [] { const char *e = fw_knob("FW_DH_SPEC"); return e ? atoi(e) : 4; }();
    // discrete kinds: persistent wavefronts + boards (dh_mi_target_kernel); FW_MI_ROUNDS=1 keeps the level-synchronous rounds over
    // the segment kernels (the path of the ABI's fw_test_subsets_batch) for comparison.  Fisher-z always runs as rounds (a
    // persistent-workgroup variant was built in r02, lost 140 vs 58 ms on the heavy rounds, and was removed in r03).
    static const bool mi_rounds = [] { const char *e = fw_knob("FW_MI_ROUNDS"); return e && atoi(e) != 0; }();
    // (more than 65 535 samples: 32-bit cell counts -- only the segment kernels have that form, so such data takes the rounds)
    const bool nzk = c->P.kind == FW_FZ_NZ;  // fz_nz (r05): rounds like fz, with the sub-matrix kernel in front of the segment kernel
    const bool per_target = c->P.kind != FW_FZ && !nzk && !mi_rounds;  // (r04: the persistent kernel has a 32-bit-count form too, PRE = 2)
    const int spec_depth = c->P.kind == FW_FZ ? std::min(std::max(spec_env, 0), DH_MAX_SPEC) : 0;
    const int d1 = spec_depth + 1;
    // interleaving-phase look-ahead (first windows of the next candidates, same accepted list): FW_DH_SPEC0 candidates,
    // only while the last launch held fewer than FW_DH_SPEC0_BELOW ranks and fewer than FW_DH_SPEC0_JOBS jobs -- it
    // pays where the rounds are latency-bound, i.e. on a rank of a multi-GPU job (one rank of 8: 103.6 -> 95.2 ms,
    // one of 2: 208 -> 204.7 ms) and in the tail of a single-GPU pass (315.8 -> 314.2 ms)
    static const int spec0_env = [] { const char *e = fw_knob("FW_DH_SPEC0"); return e ? atoi(e) : 2; }();
    const int spec0_depth = c->P.kind == FW_FZ ? std::min(std::max(spec0_env, 0), DH_MAX_SPEC) : 0;
    // segments per launch by launch size (fz): below FW_SEG_A ranks a third of seg_target, below FW_SEG_B two thirds.
    // cfg3, ms per pass on one GPU / one rank of 2 / of 8: fixed 3 072: 298.8 / 200.3 / 94.9; A, B = 4M, 8M: 295.8 /
    // 194.4 / 82.3; 6M, 10M: 295.8 / 192.8 / 80.3; 8M, 12M: 295.1 / 193.7 / 79.8 (fixed 1 024: 81.1 for the rank of 8
    // but 221 for the rank of 2; 512: 102)
    static const unsigned long long seg_a_env = [] { const char *e = fw_knob("FW_SEG_A"); return e ? (unsigned long long)atoll(e) : 8000000ull; }();
    static const unsigned long long seg_b_env = [] { const char *e = fw_knob("FW_SEG_B"); return e ? (unsigned long long)atoll(e) : 12000000ull; }();
    const unsigned long long seg_a = (c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) ? seg_a_env : 0ull, seg_b = (c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) ? seg_b_env : 0ull;
    const unsigned max_ns = seg_target + (unsigned)ntg * (unsigned)(1 + DH_MAX_SPEC) + 256u;  // capacity of the segment list (a job + its look-ahead jobs each round up)
    // striding workgroups of the segment kernel.  FW_SEG_GRID caps them (experiment: with fewer workgroups than resident
    // slots the one-workgroup step / plan kernels of the OTHER chain find a free CU at once instead of queueing behind
    // this launch's pending workgroups -- cfg5 profile: dh_plan_kernel 6.3 ms per call, all of it waiting)
    static const unsigned seg_grid_env = [] { const char *e = fw_knob("FW_SEG_GRID"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : 0u; }();
    // r06 (cfg3, one box): 3 584 workgroups 152.6 ms / ff = 0 135.2; 2 560: 151.6 / 133.3; 2 304: 150.6 / 132.2; **2 048: 149.3 / 131.8**; 1 792: 151.7 / 135.6; 1 536: 157.6; 1 024: 164 -- two
    // resident sets of the size-3 table kernel (256 CUs x 4 workgroups), the rest of the list by striding (profiles/r06_cfg3_segment_grid.txt).  max_k > 3 and fz_nz: not measured, as before.
    const unsigned grid_dflt = (c->P.kind == FW_FZ && c->P.max_k <= 3) ? std::min(seg_target + 512u, 2048u) : seg_target + 512u;
    const unsigned grid_seg = seg_grid_env ? std::min(seg_target + 512u, seg_grid_env) : grid_dflt;
    // FW_DH_LOG=<file>: one line per planned launch (ranks, live jobs, segments) -- profiling aid, see profiles/README.md
    static const char *log_path = fw_knob("FW_DH_LOG");
    constexpr unsigned LOG_CAP = 1u << 16;
    size_t need = (log_path ? pad(sizeof(ulonglong2) * LOG_CAP) : 0) + pad(sizeof(int32_t) * 2 * (size_t)ntg) + pad(sizeof(unsigned int) * ((size_t)ntg + 1)) + pad(sizeof(DhTgt) * ntg) + pad(sizeof(DhGlobal)) + 3 * pad(sizeof(long long) * ((size_t)ntg + 1));
    need += pad(4 * tot + 4) * 3 + pad(4 * 2 * tot * (size_t)d1 + 4) + pad(8 * tot + 8) * 4 + pad(4 * wl.size() + 4);
    need += pad(sizeof(FwSeg) * max_ns) + pad(sizeof(FwSegOut) * max_ns) + pad(sizeof(float) * tm_floats + 4);
    if (!nb_on_dev) need += pad(8 * ((size_t)p + 1)) + pad(4 * nnz + 4) + 2 * pad(8 * nnz + 8);
    const size_t rec_cap = MI_REC_CAP, bacc_cap = MI_BACC_CAP;
    if (per_target) need += pad(sizeof(MiQueue)) + pad(sizeof(MiBoard) * MI_BOARD_CAP) + pad(sizeof(FwSegOut) * rec_cap) + pad(sizeof(int32_t) * bacc_cap);
    // fz_nz: a record and an arena slice of (longest list + 2)^2 floats per target.  Without whitelists an accepted list never outgrows
    // the candidate list; with whitelists (feed-forward) a whitelisted member of the elimination pool is pushed a second time
    // (hiton.jl:24-26), so a list can reach twice the candidates -- the capacity of the accepted buffers (DH_ACC_OFF: 2 x cap)
    std::vector<long long> nz_aoff;
    size_t nz_arena = 0;
    if (nzk) {
        nz_aoff.resize((size_t)ntg);
        for (int t = 0; t < ntg; ++t) {
            nz_aoff[t] = (long long)nz_arena;
            const size_t mt = (size_t)(wl.empty() ? 1 : 2) * (size_t)tg[t].nc + 2;
            nz_arena += mt * mt;
        }
        need += pad(sizeof(FwNzJob) * (size_t)ntg) + pad(sizeof(long long) * (size_t)ntg) + pad(sizeof(float) * nz_arena + 4);
    }
    int rc;
    if ((rc = fw_dev_reserve(c, c->d_dh[chain], need))) return rc;
    if ((rc = fw_pin_reserve(c, c->h_dh[chain], 4096))) return rc;
    DhGlobal *hg = (DhGlobal *)c->h_dh[chain].ptr;  // two pinned copies of the device record (one per batch in flight)
    memset(hg, 0, 2 * sizeof(DhGlobal));
    char *B = (char *)c->d_dh[chain].ptr;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        char *q = B + off;
        off += pad(bytes);
        return q;
    };
    DhTgt *d_tg = (DhTgt *)carve(sizeof(DhTgt) * ntg);
    DhGlobal *d_g = (DhGlobal *)carve(sizeof(DhGlobal));
    long long *d_seg0 = (long long *)carve(sizeof(long long) * ((size_t)ntg + 1));
    unsigned long long *d_win = (unsigned long long *)carve(sizeof(unsigned long long) * ((size_t)ntg + 1));
    unsigned int *d_sp = (unsigned int *)carve(sizeof(unsigned int) * ((size_t)ntg + 1));
    unsigned long long *d_win2 = (unsigned long long *)carve(sizeof(unsigned long long) * ((size_t)ntg + 1));
    int32_t *d_act = (int32_t *)carve(sizeof(int32_t) * 2 * (size_t)ntg);  // ping-pong list of the unfinished targets
    DhArrays A{};
    int32_t *d_cand0 = (int32_t *)carve(4 * tot + 4);
    A.cand0 = use_devc ? c->d_cand : d_cand0;
    A.tpc_key = (int32_t *)carve(4 * tot + 4);
    A.pc_key = (int32_t *)carve(4 * tot + 4);
    A.acc = (int32_t *)carve(4 * 2 * tot * (size_t)d1 + 4);
    A.tpc_stat = (double *)carve(8 * tot + 8);
    A.tpc_p = (double *)carve(8 * tot + 8);
    A.pc_stat = (double *)carve(8 * tot + 8);
    A.pc_p = (double *)carve(8 * tot + 8);
    int32_t *d_wl = (int32_t *)carve(4 * wl.size() + 4);
    A.wl = d_wl;
    FwSeg *d_segs = (FwSeg *)carve(sizeof(FwSeg) * max_ns);
    FwSegOut *d_so = (FwSegOut *)carve(sizeof(FwSegOut) * max_ns);
    float *d_tmat = tm_floats ? (float *)carve(sizeof(float) * tm_floats + 4) : nullptr;
    ulonglong2 *d_log = log_path ? (ulonglong2 *)carve(sizeof(ulonglong2) * LOG_CAP) : nullptr;
    MiQueue *d_mq = per_target ? (MiQueue *)carve(sizeof(MiQueue)) : nullptr;
    MiBoard *d_boards = per_target ? (MiBoard *)carve(sizeof(MiBoard) * MI_BOARD_CAP) : nullptr;
    FwSegOut *d_mres = per_target ? (FwSegOut *)carve(sizeof(FwSegOut) * rec_cap) : nullptr;
    int32_t *d_bacc = per_target ? (int32_t *)carve(sizeof(int32_t) * bacc_cap) : nullptr;
    FwNzJob *d_nzrecs = nzk ? (FwNzJob *)carve(sizeof(FwNzJob) * (size_t)ntg) : nullptr;
    long long *d_nzaoff = nzk ? (long long *)carve(sizeof(long long) * (size_t)ntg) : nullptr;
    float *d_nzarena = nzk ? (float *)carve(sizeof(float) * nz_arena + 4) : nullptr;
    if (nzk) {
        FW_HIP(c, hipMemcpyAsync(d_nzaoff, nz_aoff.data(), sizeof(long long) * (size_t)ntg, hipMemcpyHostToDevice, st));
        FW_HIP(c, hipMemsetAsync(d_nzrecs, 0xff, sizeof(FwNzJob) * (size_t)ntg, st));  // (pad bit 0 set: nothing to compute until a job writes its record)
    }
    FW_HIP(c, hipMemcpyAsync(d_tg, tg.data(), sizeof(DhTgt) * ntg, hipMemcpyHostToDevice, st));
    hg[0].n_act = (unsigned int)ntg;  // every target starts on the list (the pinned page is the staging copy: stream-ordered)
    FW_HIP(c, hipMemcpyAsync(d_g, hg, sizeof(DhGlobal), hipMemcpyHostToDevice, st));
    {
        std::vector<int32_t> iota((size_t)ntg);
        for (int t = 0; t < ntg; ++t) iota[t] = t;
        FW_HIP(c, hipMemcpy(d_act, iota.data(), sizeof(int32_t) * (size_t)ntg, hipMemcpyHostToDevice));
    }
    FW_HIP(c, hipMemsetAsync(d_seg0, 0, sizeof(long long) * ((size_t)ntg + 1), st));
    if (tot && !use_devc) FW_HIP(c, hipMemcpyAsync(d_cand0, cand0.data(), 4 * tot, hipMemcpyHostToDevice, st));
    if (!wl.empty()) FW_HIP(c, hipMemcpyAsync(d_wl, wl.data(), 4 * wl.size(), hipMemcpyHostToDevice, st));
    if (nb_on_dev) {
        A.nb_off = c->d_nb_off;
        A.nb_idx = c->d_nb_idx;
        A.nb_stat = c->d_nb_stat;
        A.nb_p = c->d_nb_p;
    } else {
        long long *o = (long long *)carve(8 * ((size_t)p + 1));
        int32_t *ix = (int32_t *)carve(4 * nnz + 4);
        double *s1 = (double *)carve(8 * nnz + 8), *s2 = (double *)carve(8 * nnz + 8);
        FW_HIP(c, hipMemcpyAsync(o, c->nb_off.data(), 8 * ((size_t)p + 1), hipMemcpyHostToDevice, st));
        if (nnz) {
            FW_HIP(c, hipMemcpyAsync(ix, c->nb_idx.data(), 4 * nnz, hipMemcpyHostToDevice, st));
            FW_HIP(c, hipMemcpyAsync(s1, c->nb_stat.data(), 8 * nnz, hipMemcpyHostToDevice, st));
            FW_HIP(c, hipMemcpyAsync(s2, c->nb_p.data(), 8 * nnz, hipMemcpyHostToDevice, st));
        }
        A.nb_off = o;
        A.nb_idx = ix;
        A.nb_stat = s1;
        A.nb_p = s2;
    }
    A.tmat = d_tmat;
    if (d_tmat) {
        hipLaunchKernelGGL(dh_tmat_build_kernel, dim3((unsigned)ntg, 8u), dim3(256), 0, st, (const DhTgt *)d_tg, ntg, A.nb_idx, (const float *)c->d_cor, p, d_tmat);
        FW_HIP(c, hipGetLastError());
        if (trace_host) fprintf(stderr, "[fw] chain %d: local correlation matrices: %.1f MB\n", chain, 4e-6 * (double)tm_floats);
    }
    const bool fz = c->P.kind == FW_FZ || nzk;  // the rounds over the Fisher-z segment kernels (fz_nz: on job-local matrices)
    DhParams P = dh_make_params(c, ntg, spec_depth, spec0_depth);
    // The in-lane kernel (accepted lists beyond FW_TAB_A) is only launched when such a list can exist in the coming batch:
    // without whitelists an accepted list grows by at most one entry per round, so max_a (longest list so far, read
    // back once per batch) + BATCH bounds it; with whitelists a round can append several entries -> static bound.
    const bool any_wl = !wl.empty();
    const bool any_big_static = (any_wl ? 2 * max_cap : max_cap) > FW_TAB_A;
    const unsigned g_fill = (max_ns + 255) / 256;
    unsigned n_act_bound = (unsigned)ntg;  // unfinished targets as of the last record read (only ever shrinks)
    const unsigned *d_ns = &d_g->ns;
    // ---- rounds ----
    // Kernel timing (fw_counters.t_dev_subsets_s): HIP events around one segment launch in FW_DH_TIME_EVERY (default 4),
    // the sampled slot rotating from batch to batch so that every position of the 16-round batch is covered; the
    // sampled average is scaled to all non-empty launches.  Every event pair costs ~12 us of idle GPU around the launch
    // (rocprofv3 trace: 5.9 us before + 5.7 us after, back-to-back otherwise): cfg3, ms per pass / average launch us at
    // k = 1: 275.3 / 224.2, k = 2: 272.2 / 224.5, k = 4: 270.4 / 224.5, k = 8: 270.2 / 224.8 -- same average, 2 % less time.
    // Two batches are kept in flight: the host enqueues batch b + 1 before it waits for the end of batch b, so the GPU
    // never runs dry while the host looks at the round record (a stream synchronisation per batch left ~200 us of
    // idle GPU per 16 rounds).  Rounds after the last one are no-ops (no live segment, nothing to merge).
    constexpr int BATCH = 16;
    // Rounds per batch.  The host sees "every target has finished" one batch late, so a pass of few, light targets (a rank of an
    // 8-rank job holds 128 targets per feed-forward round at cfg3, 15 dependent rounds) ran 33 rounds, half of them empty: short
    // batches for short lists (r03: 1.05 -> 0.75 ms per light round; the host enqueues 4 rounds in ~60 us, a round takes 25-100 us)
    static const int nb_env = [] { const char *e = fw_knob("FW_DH_BATCH"); return e && atoi(e) > 0 ? std::min(atoi(e), 16) : 0; }();
    const int nb = nb_env ? nb_env : (ntg <= 1024 ? 4 : BATCH);
    static const int time_every = [] { const char *e = fw_knob("FW_DH_TIME_EVERY"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
    // events live in a holder whose destructor synchronises the stream and destroys them on EVERY exit path (error returns
    // and the watchdog of the persistent kernel included: r02 leaked 66 events per failed call and left the stream running)
    struct EvHolder {
        hipStream_t st;
        hipEvent_t ev[2][2 * BATCH], ev_end[2];
        bool ok = true;
        explicit EvHolder(hipStream_t s) : st(s)
        {
            for (int q = 0; q < 2; ++q) {
                for (hipEvent_t &e : ev[q]) e = nullptr;
                ev_end[q] = nullptr;
            }
            for (int q = 0; q < 2 && ok; ++q) {
                for (hipEvent_t &e : ev[q]) ok = ok && hipEventCreate(&e) == hipSuccess;
                ok = ok && hipEventCreateWithFlags(&ev_end[q], hipEventDisableTiming) == hipSuccess;
            }
        }
        ~EvHolder()
        {
            (void)hipStreamSynchronize(st);
            for (int q = 0; q < 2; ++q) {
                for (hipEvent_t &e : ev[q])
                    if (e) (void)hipEventDestroy(e);
                if (ev_end[q]) (void)hipEventDestroy(ev_end[q]);
            }
        }
    } evh(st);
    if (!evh.ok) return fw_fail(c, FW_ERR_DEVICE, "device HITON: hipEventCreate failed");
    auto &ev = evh.ev;
    auto &ev_end = evh.ev_end;
    DhPlanArgs PA{};
    PA.seg_target = seg_target;
    PA.seg_q = P.seg_q;
    PA.seg_min = P.seg_min;
    PA.log_cap = LOG_CAP;
    PA.log = d_log;
    PA.seg_a = seg_a;
    PA.seg_b = seg_b;
    // FW_DH_HP=1: the small kernels of a round on the chain's high-priority stream (fw_ctx::dh_hp_stream) -- an experiment kept behind its knob
    static const int hp_env = [] { const char *e = fw_knob("FW_DH_HP"); return e ? atoi(e) : -1; }();
    const bool use_hp = !per_target && hp_env > 0;  // (measured at cfg5: no effect -- what held the plan kernel back was its size, not its queue; default off)
    static const int ps_env = [] { const char *e = fw_knob("FW_DH_PLAN_SMALL"); return e ? atoi(e) : -1; }();
    const bool plan_small = ps_env >= 0 ? ps_env != 0 : true;
    hipStream_t hs = st;
    if (use_hp) {
        if (!c->dh_hp_stream[chain]) {
            int lo = 0, hi = 0;
            FW_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));  // (hi: numerically lowest = highest priority)
            FW_HIP(c, hipStreamCreateWithPriority(&c->dh_hp_stream[chain], hipStreamNonBlocking, hi));
            for (int e = 0; e < 2; ++e) FW_HIP(c, hipEventCreateWithFlags(&c->dh_hp_ev[chain][e], hipEventDisableTiming));
        }
        hs = c->dh_hp_stream[chain];
    }
    auto planfill = [&](bool compact) {
        if (use_hp) {  // behind the segment kernel of this round ...
            (void)hipEventRecord(c->dh_hp_ev[chain][0], st);
            (void)hipStreamWaitEvent(hs, c->dh_hp_ev[chain][0], 0);
        }
        hipLaunchKernelGGL(dh_step_kernel, dim3((n_act_bound + 3u) / 4u), dim3(256), 0, hs, d_tg, ntg, d_g, A,
                           (const FwSegOut *)d_so, (const long long *)d_seg0, d_win, d_sp, d_win2, (const int32_t *)d_act, P);
        if (compact)  // between step and plan: seg0 of the coming launch is built on the new list
            hipLaunchKernelGGL(dh_compact_kernel, dim3(1), dim3(1024), 0, hs, (const DhTgt *)d_tg, ntg, d_g, d_act);
        if (plan_small)
            hipLaunchKernelGGL(dh_plan_small_kernel, dim3(1), dim3(256), 0, hs, ntg, d_g, (const unsigned long long *)d_win,
                               (const unsigned int *)d_sp, (const unsigned long long *)d_win2, (const int32_t *)d_act, d_seg0, PA);
        else
            hipLaunchKernelGGL(dh_plan_kernel, dim3(1), dim3(1024), 0, hs, ntg, d_g, (const unsigned long long *)d_win,
                               (const unsigned int *)d_sp, (const unsigned long long *)d_win2, (const int32_t *)d_act, d_seg0, PA);
        hipLaunchKernelGGL(dh_fill_kernel, dim3(g_fill), dim3(256), 0, hs, (const DhTgt *)d_tg, ntg, d_g,
                           (const long long *)d_seg0, A, d_segs, d1, (const int32_t *)d_act);
        if (use_hp) {  // ... and in front of the next one
            (void)hipEventRecord(c->dh_hp_ev[chain][1], hs);
            (void)hipStreamWaitEvent(st, c->dh_hp_ev[chain][1], 0);
        }
    };
    const double th1 = wall();
    int rc2 = FW_OK;
    double timed_s = 0.0;
    long timed_n = 0, launches_n = 0;
    std::vector<float> log_ms;  // FW_DH_LOG: segment-kernel time of launch i (planned by plan #i)
    if (per_target) {
        std::vector<int32_t> order((size_t)ntg);
        for (int t = 0; t < ntg; ++t) order[t] = t;
        std::stable_sort(order.begin(), order.end(), [&](int32_t u, int32_t v) { return tg[u].nc > tg[v].nc; });  // heaviest first
        FW_HIP(c, hipMemcpyAsync(d_act, order.data(), sizeof(int32_t) * (size_t)ntg, hipMemcpyHostToDevice, st));
        {
        unsigned team = 0u;
        {   // targets with at least FW_MI_TEAM_MIN candidates (at most FW_MI_TEAM_MAX of them): a workgroup each
            const unsigned team_min = dh_team_min(), team_max = dh_team_max();
            while (team_min > 0u && team < team_max && (int)team < ntg && (unsigned)tg[order[team]].nc >= team_min) ++team;
            if (trace_host) {
                int c32 = 0, c64 = 0, c128 = 0, c192 = 0;
                for (int t = 0; t < ntg; ++t) c32 += tg[t].nc >= 32, c64 += tg[t].nc >= 64, c128 += tg[t].nc >= 128, c192 += tg[t].nc >= 192;
                fprintf(stderr, "[fw] chain %d: %u targets run by a workgroup each (>= %u candidates); targets with >= 32 / 64 / 128 / 192 candidates: %d / %d / %d / %d; "
                                "candidates of the 1st / 64th / 256th heaviest: %d / %d / %d\n", chain, team, team_min, c32, c64, c128, c192,
                        tg[order[0]].nc, ntg > 63 ? tg[order[63]].nc : -1, ntg > 255 ? tg[order[255]].nc : -1);
            }
        }
        FW_HIP(c, hipMemsetAsync(d_mq, 0, sizeof(MiQueue), st));
        FW_HIP(c, hipMemsetAsync(d_boards, 0, sizeof(MiBoard) * MI_BOARD_CAP, st));  // ready flags, claimed / finished counts
        FW_HIP(c, hipEventRecord(ev[0][0], st));
        const unsigned grid = dh_mi_launch(c, st, d_tg, ntg, (const int32_t *)d_act, A, P, team, trace_host, d_mq, d_boards, d_mres, d_bacc);
        FW_HIP(c, hipGetLastError());
        FW_HIP(c, hipEventRecord(ev[0][1], st));
        FW_HIP(c, hipStreamSynchronize(st));
        MiQueue hq{};
        FW_HIP(c, hipMemcpy(&hq, d_mq, sizeof(hq), hipMemcpyDeviceToHost));
        if (hq.pad[0]) return fw_fail(c, FW_ERR_DEVICE, "discrete HITON kernel: watchdog %u (boards %u, targets done %u of %d)", hq.pad[0], hq.n_boards, hq.targets_done, ntg);
        if (trace_host) dh_mi_trace(hq, grid);
        float ms = 0.0f;
        FW_HIP(c, hipEventElapsedTime(&ms, ev[0][0], ev[0][1]));
        timed_s = 1e-3 * (double)ms;
        timed_n = launches_n = 1;
        }
    } else {
    planfill(true);  // nothing to merge yet: creates the first jobs and the first launch (plan #0)
    max_a_seen = 0;
    max_ab_seen = (unsigned)max_wl;  // before any record: all whitelisted neighbours are still to come
    bool big_skipped[2] = {false, false};  // per batch slot: some round of it ran without the in-lane kernel
    auto enqueue_batch = [&](unsigned b) -> int {
        const int q = (int)(b & 1u);
        for (int r = 0; r < nb; ++r) {
            const bool timed = ((r + (int)(b % (unsigned)time_every)) % time_every) == 0;  // the sampled slot rotates from batch to batch
            // lists grow by at most one entry per round: 3 batches cover the lag of the record plus this batch
            // ... and a whitelisted neighbour is appended at most once per target: max_ab (accepted + whitelisted neighbours
            // still to come, maximum over the targets, kept by dh_step_kernel) bounds what whitelists can add.  (r02
            // profile: with whitelists the in-lane kernel was launched every round "in case" and left on the device flag --
            // 16 us of every round's critical path, 26 ms per chain and pass at cfg3.)
            const bool any_big = any_big_static && max_ab_seen + 3u * (unsigned)BATCH + 1u > (unsigned)FW_TAB_A;
            if (r == 0) big_skipped[q] = false;
            if (!any_big) big_skipped[q] = true;
            if (timed) (void)hipEventRecord(ev[q][2 * r], st);
            // discrete segment kernel: one workgroup per record, no stride loop -> the grid follows the bound on the list
            const unsigned grid_mi = std::min(max_ns, seg_target + n_act_bound * (unsigned)(1 + std::max(spec_depth, spec0_depth)) + 256u);
            int rc;
            if (nzk) {
                // this round's fresh jobs: records (one thread per unfinished target), their matrices, then the enumeration
                hipLaunchKernelGGL(dh_nz_recs_kernel, dim3((n_act_bound + 255u) / 256u), dim3(256), 0, st, (const DhTgt *)d_tg, ntg, (const DhGlobal *)d_g,
                                   (const int32_t *)d_act, A, d_nzrecs, (const long long *)d_nzaoff);
                rc = fwi_fznz_submatrices_dev(c, ntg, d_nzrecs, A.acc, d_nzarena, (any_wl ? 2 * max_cap : max_cap) + 2, true, st);
                if (!rc) rc = fwi_fznz_segments_dev(c, grid_seg, d_segs, A.acc, d_so, d_ns, any_big, &d_g->any_big, d_nzrecs, d_nzarena, st);
            } else {
                rc = fz ? fwi_fz_segments_dev(c, grid_seg, d_segs, A.acc, d_so, d_ns, any_big, &d_g->any_big, st)
                        : fwi_mi_segments_dev(c, grid_mi, d_segs, A.acc, d_so, d_ns, st);
            }
            if (rc) return rc;
            if (timed) (void)hipEventRecord(ev[q][2 * r + 1], st);
            planfill((b * (unsigned)nb + (unsigned)r) % 16u == 0u);  // the list of unfinished targets is compacted every 16 rounds
        }
        FW_HIP(c, hipGetLastError());
        FW_HIP(c, hipMemcpyAsync(hg + q, d_g, sizeof(DhGlobal), hipMemcpyDeviceToHost, st));
        FW_HIP(c, hipEventRecord(ev_end[q], st));
        return FW_OK;
    };
    // returns 1 when the record of batch b says that every target has finished
    auto retire_batch = [&](unsigned b, int *done) -> int {
        const int q = (int)(b & 1u);
        FW_HIP(c, hipEventSynchronize(ev_end[q]));
        const DhGlobal &rec = hg[q];
        for (int r = 0; r < nb; ++r) {
            if (rec.ns_ring[(b * (unsigned)nb + (unsigned)r) & 63u] == 0) continue;  // empty launch after the last round
            ++launches_n;
            if (((r + (int)(b % (unsigned)time_every)) % time_every) != 0) continue;
            float ms = 0.0f;
            FW_HIP(c, hipEventElapsedTime(&ms, ev[q][2 * r], ev[q][2 * r + 1]));
            timed_s += 1e-3 * (double)ms;
            ++timed_n;
            if (log_path) {
                const size_t idx = (size_t)b * (size_t)nb + (size_t)r;
                if (log_ms.size() <= idx) log_ms.resize(idx + 1, 0.0f);
                log_ms[idx] = ms;
            }
        }
        if (fz && c->P.max_k <= 3 && big_skipped[q] && rec.max_a > (unsigned)FW_TAB_A)  // the bound above failed: fail loudly
            return fw_fail(c, FW_ERR_DEVICE, "device rounds: an accepted list of %u entries met a batch without the in-lane kernel", rec.max_a);
        max_a_seen = rec.max_a > max_a_seen ? rec.max_a : max_a_seen;
        max_ab_seen = rec.max_ab > max_ab_seen ? rec.max_ab : max_ab_seen;
        n_act_bound = rec.n_act < n_act_bound ? rec.n_act : n_act_bound;
        *done = rec.done != 0u;
        return FW_OK;
    };
    {
        unsigned b = 0;
        int done = 0;
        rc2 = enqueue_batch(0);
        while (!rc2) {
            if ((rc2 = enqueue_batch(b + 1))) break;   // keep the GPU fed ...
            if ((rc2 = retire_batch(b, &done))) break;  // ... while the host reads the previous batch's record
            ++b;
            if (done) {
                int d2 = 0;
                rc2 = retire_batch(b, &d2);  // the batch still in flight holds only no-op rounds
                break;
            }
            if (b > 250000u) {  // every round finishes at least one window: this is a logic error, not a workload
                rc2 = fw_fail(c, FW_ERR_DEVICE, "device HITON: no convergence after %u rounds", b * (unsigned)nb);
                break;
            }
        }
        if (rc2) (void)hipStreamSynchronize(st);
    }
    }  // rounds
    static std::mutex cnt_mu;  // concurrent chains share the context's counters
    {
        std::lock_guard<std::mutex> lk(cnt_mu);
        if (timed_n > 0) c->cnt.t_dev_subsets_s += timed_s * (double)launches_n / (double)timed_n;
        c->cnt.subsets_launches += launches_n;
        c->cnt.kernel_launches += per_target ? launches_n : (nzk ? 7 : 4) * launches_n;
    }
    if (rc2) return rc2;
    const double th2 = wall();
    // ---- results ----
    FW_HIP(c, hipMemcpy(tg.data(), d_tg, sizeof(DhTgt) * ntg, hipMemcpyDeviceToHost));
    if (d_log) {
        std::vector<ulonglong2> lg(LOG_CAP);
        DhGlobal fin{};
        FW_HIP(c, hipMemcpy(&fin, d_g, sizeof(DhGlobal), hipMemcpyDeviceToHost));
        FW_HIP(c, hipMemcpy(lg.data(), d_log, sizeof(ulonglong2) * LOG_CAP, hipMemcpyDeviceToHost));
        long long n_tpc = 0, n_pc = 0;  // interleaving survivors / elimination survivors
        for (const DhTgt &x : tg) {
            n_tpc += x.ntpc;
            n_pc += x.npc;
        }
        if (FILE *f = fopen(log_path, "a")) {
            fprintf(f, "# targets %d rounds %u tpc %lld pc %lld\n", ntg, fin.rounds, n_tpc, n_pc);
            for (unsigned r = 0; r < fin.rounds && r < LOG_CAP; ++r)
                fprintf(f, "%u %llu %llu %llu %.1f\n", r, lg[r].x, lg[r].y >> 32, lg[r].y & 0xffffffffull,
                        r < log_ms.size() ? 1e3 * (double)log_ms[r] : 0.0);
            fclose(f);
        }
    }
    std::vector<int32_t> &pk = flat.key;
    std::vector<double> &ps = flat.stat, &pp = flat.pval;
    pk.resize(tot);
    ps.resize(tot);
    pp.resize(tot);
    if (tot) {
        FW_HIP(c, hipMemcpy(pk.data(), A.pc_key, 4 * tot, hipMemcpyDeviceToHost));
        FW_HIP(c, hipMemcpy(ps.data(), A.pc_stat, 8 * tot, hipMemcpyDeviceToHost));
        FW_HIP(c, hipMemcpy(pp.data(), A.pc_p, 8 * tot, hipMemcpyDeviceToHost));
    }
    for (int t = 0; t < ntg; ++t) {
        const DhTgt &x = tg[t];
        if (x.phase != 2) return fw_fail(c, FW_ERR_DEVICE, "device HITON: target %d did not finish (phase %d)", x.T, x.phase);
        out[t].off = x.co;
        out[t].n = x.npc;
    }
    if (trace_host) {  // chain statistics: the longest per-target sequences bound the pass from below
        unsigned long long mx_ref = 0, mx_calls = 0, tot_ref = 0, tot_calls = 0;
        int t_ref = -1, t_calls = -1;
        for (const DhTgt &x : tg) {
            tot_ref += x.c_ref;
            tot_calls += x.c_calls;
            if (x.c_ref > mx_ref) mx_ref = x.c_ref, t_ref = x.T;
            if (x.c_calls > mx_calls) mx_calls = x.c_calls, t_calls = x.T;
        }
        fprintf(stderr, "[fw] chain %d: tests %llu jobs %llu; most tests in one target %llu (T=%d), most jobs in one target %llu (T=%d)\n",
                chain, tot_ref, tot_calls, mx_ref, t_ref, mx_calls, t_calls);
        unsigned long long ev_all = 0, ev_short = 0;
        int na_max = 0;
        for (const DhTgt &x : tg) {
            ev_all += x.c_eval;
            ev_short += x.c_eval_short;
            na_max = std::max(na_max, x.na_max);
        }
        fprintf(stderr, "[fw] chain %d: executed tests %llu, of them in jobs with at most %d accepted variables %llu; longest accepted list %d\n",
                chain, ev_all, (int)FW_HK_A, ev_short, na_max);
        if (per_target) {  // the targets that finished last: when they were taken, how long they ran, what they ran
            unsigned int t0 = ~0u;
            for (const DhTgt &x : tg) t0 = std::min(t0, x.r_first0);
            std::vector<const DhTgt *> by_end;
            for (const DhTgt &x : tg) by_end.push_back(&x);
            std::sort(by_end.begin(), by_end.end(), [&](const DhTgt *a, const DhTgt *b) { return a->r_more0 - t0 > b->r_more0 - t0; });
            for (size_t i = 0; i < by_end.size() && i < 6; ++i) {
                const DhTgt *w = by_end[i];
                fprintf(stderr, "[fw]   finished at %.2f ms (taken at %.2f): T=%d, %d candidates, %d in PC, %llu jobs, %llu tests (%llu executed); "
                                "sequential prefixes %.2f ms, board phases %.2f ms (%llu jobs went to a board; %llu team rounds)\n",
                        1e-5 * (w->r_more0 - t0), 1e-5 * (w->r_first0 - t0), w->T, w->cap, w->npc, w->c_calls, w->c_ref, w->c_eval,
                        1e-5 * w->r_first1, 1e-5 * w->r_more1, w->c_eval_short & 0xffffffffull, w->c_eval_short >> 32);
            }
        } else {  // the target that was busy for the most rounds: where its rounds went
            const DhTgt *w = &tg[0];
            for (const DhTgt &x : tg)
                if (x.r_first0 + x.r_more0 + x.r_first1 + x.r_more1 > w->r_first0 + w->r_more0 + w->r_first1 + w->r_more1) w = &x;
            fprintf(stderr, "[fw] chain %d: longest-busy target T=%d: %d candidates, %d in TPC, %d in PC, %llu jobs; rounds: interleaving first windows %u, later windows %u; elimination first %u, later %u\n",
                    chain, w->T, w->cap, w->ntpc, w->npc, w->c_calls, w->r_first0, w->r_more0, w->r_first1, w->r_more1);
        }
    }
    if (trace_host)
        fprintf(stderr, "[fw] chain %d: longest accepted list %u, accepted + whitelisted to come %u (most whitelisted neighbours of one target %d)\n",
                chain, max_a_seen, max_ab_seen, max_wl);
    if (trace_host)
        fprintf(stderr, "[fw] device rounds chain %d: %d targets, set-up %.2f ms, rounds %.2f ms, results %.2f ms\n", chain, ntg,
                1e3 * (th1 - th0), 1e3 * (th2 - th1), 1e3 * (wall() - th2));
    std::lock_guard<std::mutex> lk(cnt_mu);
    for (const DhTgt &x : tg) {
        c->cnt.cond_tests_ref += (int64_t)x.c_ref;
        c->cnt.subsets_calls += (int64_t)x.c_calls;
        c->cnt.cond_tests_evaluated += (int64_t)x.c_eval;
        c->cnt.alg_bytes_subsets += x.c_alg;
    }
    return FW_OK;
}

// The WHOLE feed-forward schedule of the discrete kinds on the device (r05; one GPU, no exchange between the rounds).  r02-r04 ran one
// persistent launch per round with the host in between: download the round's PC lists, update the running graph, build the next round's
// targets and whitelists, upload -- cfg4: ~1.4 ms per round besides the kernel (ten rounds) plus 5 ms before the first one, for nine
// launches of 1-3 ms each.  Here the state of every target of the schedule is built on the device from the level-0 CSR
// (dh_mi_init_kernel), the whitelists grow on the device between two launches (dh_wl_append_kernel: interleaved.jl:136-140 kept where a
// later round reads it), and the launches of all rounds are enqueued back to back; the host reads the results once.  Semantics per
// round are those of fwi_devhiton_run (same kernel, same per-round order and team size).  sched[0 .. nt): the targets in schedule order
// (learning.jl:97-98); rounds of R targets.  Appends (target, neighbour, statistic, p): a target's entries together, in PC insertion order.
int fwi_devhiton_mi_schedule(fw_ctx *c, const int32_t *sched, int nt, int R, bool feed_forward, std::vector<int32_t> &all_t,
                             std::vector<int32_t> &all_u, std::vector<double> &all_s, std::vector<double> &all_p)
{
    if (nt == 0) return FW_OK;
    if (!c->d_cand || !c->d_nb_idx) return fw_fail(c, FW_ERR_STATE, "device schedule: the level-0 lists are not on the device");
    static const bool trace_host = fw_knob("FW_TRACE_HOST") != nullptr;
    auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double th0 = wall();
    hipStream_t st = c->pb[0].stream;
    const int p = c->P.p;
    const size_t nnz = (size_t)c->nb_off[p];
    if (R <= 0 || R > nt) R = nt;
    const int nrounds = (nt + R - 1) / R;
    // ---- host: per-round order (heaviest first, stable), team sizes, round of every variable ----
    std::vector<int32_t> order((size_t)nt), round_of((size_t)p, 0x7fffffff);
    std::vector<unsigned> team((size_t)nrounds, 0u);
    std::vector<int32_t> deg((size_t)nt);
    for (int i = 0; i < nt; ++i) {
        deg[i] = (int32_t)(c->nb_off[sched[i] + 1] - c->nb_off[sched[i]]);
        round_of[sched[i]] = i / R;
    }
    for (int r = 0; r < nrounds; ++r) {
        const int r0 = r * R, r1 = std::min(nt, r0 + R);
        for (int i = r0; i < r1; ++i) order[i] = i - r0;
        std::stable_sort(order.begin() + r0, order.begin() + r1, [&](int32_t u, int32_t v) { return deg[r0 + u] > deg[r0 + v]; });
        unsigned tm = 0u;
        const unsigned team_min = dh_team_min(), team_max = dh_team_max();
        while (team_min > 0u && tm < team_max && (int)tm < r1 - r0 && (unsigned)deg[r0 + order[r0 + tm]] >= team_min) ++tm;
        team[r] = tm;
    }
    // ---- device arena ----
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t need = pad(sizeof(DhTgt) * (size_t)nt) + 2 * pad(4 * (size_t)nt) + pad(4 * (size_t)p) * 2 + pad(4 * nnz + 4) * 3 + pad(4 * 2 * nnz + 4) +
                  pad(8 * nnz + 8) * 4 + pad(sizeof(MiQueue) * (size_t)nrounds) + pad(sizeof(MiBoard) * MI_BOARD_CAP) +
                  pad(sizeof(FwSegOut) * (size_t)MI_REC_CAP) + pad(sizeof(int32_t) * (size_t)MI_BACC_CAP) + 2 * pad(4 * nnz + 4) + 2 * pad(8 * nnz + 8) + pad(64);
    int rc;
    if ((rc = fw_dev_reserve(c, c->d_dh[0], need))) return rc;
    char *B = (char *)c->d_dh[0].ptr;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        char *q = B + off;
        off += pad(bytes);
        return q;
    };
    DhTgt *d_tg = (DhTgt *)carve(sizeof(DhTgt) * (size_t)nt);
    int32_t *d_sched = (int32_t *)carve(4 * (size_t)nt);
    int32_t *d_order = (int32_t *)carve(4 * (size_t)nt);
    int32_t *d_round_of = (int32_t *)carve(4 * (size_t)p);
    unsigned int *d_wl_cnt = (unsigned int *)carve(4 * (size_t)p);
    DhArrays A{};
    A.cand0 = c->d_cand;
    A.tpc_key = (int32_t *)carve(4 * nnz + 4);
    A.pc_key = (int32_t *)carve(4 * nnz + 4);
    int32_t *d_wl = (int32_t *)carve(4 * nnz + 4);
    A.wl = d_wl;
    A.wl_cnt = feed_forward ? d_wl_cnt : nullptr;
    A.acc = (int32_t *)carve(4 * 2 * nnz + 4);
    A.tpc_stat = (double *)carve(8 * nnz + 8);
    A.tpc_p = (double *)carve(8 * nnz + 8);
    A.pc_stat = (double *)carve(8 * nnz + 8);
    A.pc_p = (double *)carve(8 * nnz + 8);
    A.nb_off = c->d_nb_off;
    A.nb_idx = c->d_nb_idx;
    A.nb_stat = c->d_nb_stat;
    A.nb_p = c->d_nb_p;
    MiQueue *d_mq = (MiQueue *)carve(sizeof(MiQueue) * (size_t)nrounds);
    MiBoard *d_boards = (MiBoard *)carve(sizeof(MiBoard) * MI_BOARD_CAP);
    FwSegOut *d_mres = (FwSegOut *)carve(sizeof(FwSegOut) * (size_t)MI_REC_CAP);
    int32_t *d_bacc = (int32_t *)carve(sizeof(int32_t) * (size_t)MI_BACC_CAP);
    int32_t *d_ot = (int32_t *)carve(4 * nnz + 4), *d_ou = (int32_t *)carve(4 * nnz + 4);
    double *d_os = (double *)carve(8 * nnz + 8), *d_op = (double *)carve(8 * nnz + 8);
    unsigned long long *d_tot = (unsigned long long *)carve(64);  // [0..4] integers, [5] the Float64 sum of algorithmic bytes
    FW_HIP(c, hipMemsetAsync(d_tot, 0, 64, st));
    FW_HIP(c, hipMemcpyAsync(d_sched, sched, 4 * (size_t)nt, hipMemcpyHostToDevice, st));
    FW_HIP(c, hipMemcpyAsync(d_order, order.data(), 4 * (size_t)nt, hipMemcpyHostToDevice, st));
    FW_HIP(c, hipMemcpyAsync(d_round_of, round_of.data(), 4 * (size_t)p, hipMemcpyHostToDevice, st));
    FW_HIP(c, hipMemsetAsync(d_wl_cnt, 0, 4 * (size_t)p, st));
    FW_HIP(c, hipMemsetAsync(d_mq, 0, sizeof(MiQueue) * (size_t)nrounds, st));
    hipLaunchKernelGGL(dh_mi_init_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, d_tg, nt, (const int32_t *)d_sched, c->d_nb_off,
                       (const int32_t *)c->d_levels);
    const DhParams P = dh_make_params(c, R, 0, 0);
    struct Ev2 {
        hipStream_t st;
        hipEvent_t e[2] = {nullptr, nullptr};
        bool ok;
        explicit Ev2(hipStream_t s) : st(s) { ok = hipEventCreate(&e[0]) == hipSuccess && hipEventCreate(&e[1]) == hipSuccess; }
        ~Ev2()
        {
            (void)hipStreamSynchronize(st);
            for (hipEvent_t &q : e)
                if (q) (void)hipEventDestroy(q);
        }
    } ev(st);
    if (!ev.ok) return fw_fail(c, FW_ERR_DEVICE, "device schedule: hipEventCreate failed");
    const double th1 = wall();
    FW_HIP(c, hipEventRecord(ev.e[0], st));
    std::vector<unsigned> grids((size_t)nrounds, 0u);
    for (int r = 0; r < nrounds; ++r) {
        const int r0 = r * R, ntg = std::min(nt, r0 + R) - r0;
        FW_HIP(c, hipMemsetAsync(d_boards, 0, sizeof(MiBoard) * MI_BOARD_CAP, st));  // ready flags, claimed / finished counts
        grids[r] = dh_mi_launch(c, st, d_tg + r0, ntg, (const int32_t *)(d_order + r0), A, P, team[r], trace_host, d_mq + r, d_boards, d_mres, d_bacc);
        if (feed_forward && r + 1 < nrounds)
            hipLaunchKernelGGL(dh_wl_append_kernel, dim3((unsigned)((ntg + 3) / 4)), dim3(256), 0, st, (const DhTgt *)(d_tg + r0), ntg,
                               (const int32_t *)A.pc_key, (const int32_t *)d_round_of, r, d_wl, c->d_nb_off, d_wl_cnt);
    }
    FW_HIP(c, hipGetLastError());
    FW_HIP(c, hipEventRecord(ev.e[1], st));
    // ---- results: packed on the device, one small download ----
    hipLaunchKernelGGL(dh_mi_pack_kernel, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, st, (const DhTgt *)d_tg, nt, (const int32_t *)A.pc_key,
                       (const double *)A.pc_stat, (const double *)A.pc_p, d_ot, d_ou, d_os, d_op, d_tot, (double *)(d_tot + 5));
    FW_HIP(c, hipGetLastError());
    std::vector<MiQueue> hq((size_t)nrounds);
    unsigned long long htot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    FW_HIP(c, hipMemcpyAsync(hq.data(), d_mq, sizeof(MiQueue) * (size_t)nrounds, hipMemcpyDeviceToHost, st));
    FW_HIP(c, hipMemcpyAsync(htot, d_tot, 64, hipMemcpyDeviceToHost, st));
    FW_HIP(c, hipStreamSynchronize(st));
    const double th2 = wall();
    for (int r = 0; r < nrounds; ++r) {
        if (hq[r].pad[0])
            return fw_fail(c, FW_ERR_DEVICE, "discrete HITON kernel, round %d: watchdog %u (boards %u, targets done %u)", r, hq[r].pad[0], hq[r].n_boards, hq[r].targets_done);
        if (trace_host) dh_mi_trace(hq[r], grids[r]);
    }
    float ms = 0.0f;
    FW_HIP(c, hipEventElapsedTime(&ms, ev.e[0], ev.e[1]));
    if (htot[4]) return fw_fail(c, FW_ERR_DEVICE, "device schedule: %llu targets did not finish", htot[4]);
    const size_t nres = (size_t)htot[0], at0 = all_t.size();
    if (nres > nnz) return fw_fail(c, FW_ERR_DEVICE, "device schedule: %zu result entries for %zu level-0 entries", nres, nnz);
    all_t.resize(at0 + nres);
    all_u.resize(at0 + nres);
    all_s.resize(at0 + nres);
    all_p.resize(at0 + nres);
    if (nres) {
        FW_HIP(c, hipMemcpyAsync(all_t.data() + at0, d_ot, 4 * nres, hipMemcpyDeviceToHost, st));
        FW_HIP(c, hipMemcpyAsync(all_u.data() + at0, d_ou, 4 * nres, hipMemcpyDeviceToHost, st));
        FW_HIP(c, hipMemcpyAsync(all_s.data() + at0, d_os, 8 * nres, hipMemcpyDeviceToHost, st));
        FW_HIP(c, hipMemcpyAsync(all_p.data() + at0, d_op, 8 * nres, hipMemcpyDeviceToHost, st));
        FW_HIP(c, hipStreamSynchronize(st));
    }
    c->cnt.cond_tests_ref += (int64_t)htot[1];
    c->cnt.subsets_calls += (int64_t)htot[2];
    c->cnt.cond_tests_evaluated += (int64_t)htot[3];
    {
        double alg;
        memcpy(&alg, &htot[5], sizeof(double));
        c->cnt.alg_bytes_subsets += alg;
    }
    c->cnt.t_dev_subsets_s += 1e-3 * (double)ms;
    c->cnt.subsets_launches += nrounds;
    c->cnt.kernel_launches += 2 * nrounds + 1;
    if (trace_host)
        fprintf(stderr, "[fw] device schedule: %d targets in %d rounds, set-up %.2f ms, launches + download %.2f ms (kernels %.2f ms), results %.2f ms\n", nt, nrounds,
                1e3 * (th1 - th0), 1e3 * (th2 - th1), (double)ms, 1e3 * (wall() - th2));
    return FW_OK;
}
