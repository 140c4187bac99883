// Host driver: the caller side of the hot path, restated for hosts without Julia.
//   LGL                      learning.jl:203-279 (target order :97-98)
//   si_HITON_PC              hiton.jl:283-400 (interleaving/elimination via hiton_backend :109-149,
//                            check_candidate! :80-107, update_sig_result! :53-78, update_PC_dict! :249-256)
//   feed-forward whitelist   interleaved.jl:112-183 (as level-synchronous rounds)
//   make_weights / make_symmetric_graph   misc.jl:137-159, 201-272
// Every target is a small state machine; one step of the loop collects the pending (T, candidate, accepted)
// job of every active target and runs them as ONE fw_test_subsets batch on the device.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>
#include <functional>
#include <thread>
#include <unordered_map>

#include "fw_internal.h"

namespace {

struct ODict {  // OrderedDict{Int,Tuple{Float64,Float64}}: insertion order, re-assignment keeps the slot
    std::vector<int32_t> key;
    std::vector<double> stat, pval;
    int find(int32_t k) const
    {
        for (size_t i = 0; i < key.size(); ++i)
            if (key[i] == k) return (int)i;
        return -1;
    }
    // Every key is assigned exactly once per phase (the candidate lists hold distinct variables: a neighbour list in
    // the interleaving phase, keys(TPC) in the elimination phase), so assignment is an append -- no lookup.
    void set(int32_t k, double s, double p)
    {
        key.push_back(k);
        stat.push_back(s);
        pval.push_back(p);
    }
};

struct Target {
    int32_t T = 0;
    int phase = 0;  // 0 = interleaving, 1 = elimination, 2 = finished
    size_t pos = 0;
    std::vector<int32_t> cands;   // current phase's candidate list
    int32_t nc_dev = -1;          // device rounds: number of candidates in the device-built list (no host copy)
    std::vector<int32_t> acc;     // accepted (conditioning pool)
    ODict TPC, PC;
    const int32_t *wl = nullptr;  // sorted whitelist (snapshot of the running graph)
    int wl_n = 0;
    // speculative posting (interleaving phase): candidates [pos, posted_end) have a job in the pool or a buffered result
    size_t posted_end = 0;
    std::vector<std::pair<int32_t, FwJobOut>> ready;  // finished but not yet committed (candidate index, result)
    bool in_wl(int32_t v) const { return wl_n > 0 && std::binary_search(wl, wl + wl_n, v); }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// Advance a target until it needs a device test (returns true, job = (T, cands[pos], acc)) or finishes.
static bool advance(const fw_ctx *c, Target &t)
{
    const int64_t o = c->nb_off[t.T];
    const int deg = (int)(c->nb_off[t.T + 1] - o);
    auto univar = [&](int32_t v, double &s, double &p) {
        const int32_t *b = c->nb_idx.data() + o;
        const int32_t *it = std::lower_bound(b, b + deg, v);
        s = c->nb_stat[o + (it - b)];
        p = c->nb_p[o + (it - b)];
    };
    for (;;) {
        if (t.phase == 2) return false;
        ODict &dict = t.phase == 0 ? t.TPC : t.PC;
        while (t.pos < t.cands.size()) {
            const int32_t cand = t.cands[t.pos];
            if (t.in_wl(cand)) {  // hiton.jl:20-30
                t.acc.push_back(cand);
                dict.set(cand, NAN, NAN);
                ++t.pos;
                continue;
            }
            if (t.phase == 1)  // hiton.jl:134-136
                t.acc.erase(std::remove(t.acc.begin(), t.acc.end(), cand), t.acc.end());
            if (t.acc.empty()) {  // tests.jl:285 sentinel + hiton.jl:57-59
                double s, p;
                if (t.phase == 0) {
                    univar(cand, s, p);
                } else {
                    const int i = t.TPC.find(cand);
                    s = t.TPC.stat[i];
                    p = t.TPC.pval[i];
                }
                t.acc.push_back(cand);
                dict.set(cand, s, p);
                ++t.pos;
                continue;
            }
            return true;
        }
        if (t.phase == 0) {  // hiton.jl:242: elimination over keys(TPC) in insertion order
            t.phase = 1;
            t.cands = t.TPC.key;
            t.acc = t.cands;
            t.pos = 0;
            t.posted_end = 0;  // every interleaving candidate has been committed at this point
            t.ready.clear();
        } else {  // hiton.jl:249-256 update_PC_dict!
            for (size_t i = 0; i < t.PC.key.size(); ++i) {
                const int ti = t.TPC.find(t.PC.key[i]);
                if (ti >= 0 && (t.TPC.pval[ti] > t.PC.pval[i] || std::isnan(t.PC.pval[i]))) {
                    t.PC.stat[i] = t.TPC.stat[ti];
                    t.PC.pval[i] = t.TPC.pval[ti];
                }
            }
            t.phase = 2;
        }
    }
}

static double maxweight(double w1, double w2)
{  // misc.jl:201-218
    if (std::isnan(w1)) return w2;
    if (std::isnan(w2)) return w1;
    const double s1 = (w1 > 0) - (w1 < 0), s2 = (w2 > 0) - (w2 < 0);
    if (s1 * s2 < 0) return w1;  // "arbitrarily choosing one": the lower-index endpoint's direction
    return std::max(std::fabs(w1), std::fabs(w2)) * s1;
}

// The per-round exchange of fw_learn_network through a fw_dev_exchange (fw_learn_network_dev): the round's directed entries are
// packed into 24-byte records here (no numpy on the way), copied into the caller's device send buffer, all-gathered by the caller's
// collective (RCCL on torch tensors in bench.py) and unpacked from the gathered buffer.  Implements fw_allgather_fn.
namespace {
struct DevXRec {
    int32_t t, u;
    double s, p;
};
static_assert(sizeof(DevXRec) == 24, "round exchange record");
struct DevXAdapter {
    fw_ctx *c;
    const fw_dev_exchange *x;
    int world;
    std::vector<DevXRec> stage;
    std::vector<int32_t> t, u;
    std::vector<double> s, p;
};
int devx_allgather(void *user, int64_t n, const int32_t *tgt, const int32_t *nbr, const double *stat, const double *pval, int64_t *n_total,
                   const int32_t **tgt_all, const int32_t **nbr_all, const double **stat_all, const double **pval_all)
{
    DevXAdapter *A = (DevXAdapter *)user;
    std::vector<int64_t> counts((size_t)A->world, 0), aux((size_t)A->world, 0);
    void *d_send = nullptr, *d_recv = nullptr;
    int64_t cap = 0;
    if (A->x->prepare(A->x->user, n, 0, (int32_t)sizeof(DevXRec), &d_send, &d_recv, counts.data(), aux.data(), &cap)) return 1;
    if (n > cap || !d_recv || (n > 0 && !d_send)) return 2;
    A->stage.resize((size_t)std::max<int64_t>(n, 1));
    for (int64_t i = 0; i < n; ++i) A->stage[(size_t)i] = DevXRec{tgt[i], nbr[i], stat[i], pval[i]};
    if (n > 0 && hipMemcpy(d_send, A->stage.data(), (size_t)n * sizeof(DevXRec), hipMemcpyHostToDevice) != hipSuccess) return 3;
    if (A->x->exchange(A->x->user)) return 4;
    int64_t total = 0;
    for (int r = 0; r < A->world; ++r) {
        if (counts[(size_t)r] < 0 || counts[(size_t)r] > cap) return 5;
        total += counts[(size_t)r];
    }
    A->stage.resize((size_t)std::max<int64_t>(total, 1));
    int64_t off = 0;
    for (int r = 0; r < A->world; ++r) {
        const int64_t k = counts[(size_t)r];
        if (k > 0 && hipMemcpy(A->stage.data() + off, (const char *)d_recv + (size_t)r * (size_t)cap * sizeof(DevXRec), (size_t)k * sizeof(DevXRec),
                               hipMemcpyDeviceToHost) != hipSuccess)
            return 6;
        off += k;
    }
    A->t.resize((size_t)std::max<int64_t>(total, 1));
    A->u.resize(A->t.size());
    A->s.resize(A->t.size());
    A->p.resize(A->t.size());
    for (int64_t i = 0; i < total; ++i) {
        const DevXRec &q = A->stage[(size_t)i];
        A->t[(size_t)i] = q.t;
        A->u[(size_t)i] = q.u;
        A->s[(size_t)i] = q.s;
        A->p[(size_t)i] = q.p;
    }
    *n_total = total;
    *tgt_all = A->t.data();
    *nbr_all = A->u.data();
    *stat_all = A->s.data();
    *pval_all = A->p.data();
    return 0;
}
}  // namespace

// fwi_devhiton_mi_schedule serves: FW_MI / FW_MI_NZ on bit planes, one rank, no exchange callback, level-0 lists and candidate order on
// the device, rounds the persistent kernel is worth launching for (the reference's single_il rounds of one target stay on the host pool).
// FW_MI_SCHED=0 keeps the per-round loop (A/B runs, tests/test_gpu_mi.py compares the two).
// Host threads of the graph passes at the end of fw_learn_network, kept for the life of the context: starting fifteen threads per pass
// cost more than the passes' work at cfg3 (2.4 of 2.8 ms for 48 040 edges; r05).  run(fn): fn(w, lo[w], hi[w]) for every block w, block 0
// on the caller.
#include <condition_variable>
#include <mutex>
struct FwHostWorkers {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int, int, int)> *fn = nullptr;
    const int *blk = nullptr;
    unsigned long long gen = 0;
    int pending = 0;
    bool quit = false, failed = false;
    explicit FwHostWorkers(int n)
    {
        for (int w = 1; w < n; ++w)
            th.emplace_back([this, w] {
                unsigned long long seen = 0;
                for (;;) {
                    const std::function<void(int, int, int)> *f;
                    const int *b;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return quit || gen != seen; });
                        if (quit) return;
                        seen = gen;
                        f = fn;
                        b = blk;
                    }
                    bool ok = true;
                    try {
                        (*f)(w, b[w], b[w + 1]);
                    } catch (...) {  // (std::bad_alloc of a block-local vector: an exception that leaves a thread is std::terminate)
                        ok = false;
                    }
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (!ok) failed = true;
                        if (--pending == 0) cv_done.notify_one();
                    }
                }
            });
    }
    // false: a block threw (out of memory).  The workers hold pointers to the caller's function object and block list, so the call
    // never leaves -- normally or by an exception of block 0 -- before every worker has finished its block (r05 unwound past them).
    bool run(const std::function<void(int, int, int)> &f, const int *b)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = &f;
            blk = b;
            pending = (int)th.size();
            failed = false;
            ++gen;
        }
        cv_go.notify_all();
        bool ok0 = true;
        try {
            f(0, b[0], b[1]);
        } catch (...) {
            ok0 = false;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        return ok0 && !failed;
    }
    ~FwHostWorkers()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (std::thread &t : th) t.join();
    }
};
void fwi_host_workers_free(fw_ctx *c)
{
    delete c->host_workers;
    c->host_workers = nullptr;
}

static bool mi_schedule_on_device(const fw_ctx *c, const fw_learn_opts &opt, bool has_exchange, int nt)
{
    if (!(c->P.kind == FW_MI || c->P.kind == FW_MI_NZ) || c->mi_generic || has_exchange || opt.world_size > 1 || c->P.max_k > FW_MAX_K_FAST) return false;
    if (!c->d_cand || !c->d_nb_idx || !c->d_nb_off) return false;
    const char *hh = fw_knob("FW_HOST_HITON"), *mr = fw_knob("FW_MI_ROUNDS"), *sc = fw_knob("FW_MI_SCHED"), *mt = fw_knob("FW_DEV_MIN_TARGETS");
    if ((hh && atoi(hh) == 1) || (mr && atoi(mr) != 0) || (sc && atoi(sc) == 0)) return false;
    const int R = (opt.round_size <= 0 || opt.round_size > nt) ? nt : opt.round_size;
    const int min_targets = mt ? atoi(mt) : 256;
    // (R = 1 is the reference's single_il master, whose first round holds TWO targets -- interleaved.jl:62,76-86: the per-round loop
    // below knows that rule, the device schedule cuts rounds of exactly R; r05 fuzz, 3 of 4 500 networks with the threshold forced to 1)
    return R >= min_targets && R >= 2;
}

extern "C" int fw_learn_network(fw_ctx *c, const fw_learn_opts *opts_in, fw_allgather_fn allgather, void *user, int64_t *n_edges_out);

extern "C" int fw_learn_network_dev(fw_ctx *c, const fw_learn_opts *opts_in, const fw_dev_exchange *x, int64_t *n_edges_out)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    const int world = opts_in ? std::max(opts_in->world_size, 1) : 1;
    if (world > 1 && (!x || !x->prepare || !x->exchange)) return fw_fail(c, FW_ERR_ARG, "fw_learn_network_dev: world_size > 1 needs both exchange callbacks");
    if (!x) return fw_learn_network(c, opts_in, nullptr, nullptr, n_edges_out);
    (void)hipSetDevice(c->P.device);
    DevXAdapter A{c, x, world, {}, {}, {}, {}, {}};
    return fw_learn_network(c, opts_in, devx_allgather, &A, n_edges_out);
}

extern "C" int fw_learn_network(fw_ctx *c, const fw_learn_opts *opts_in, fw_allgather_fn allgather, void *user,
                                int64_t *n_edges_out)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    (void)hipSetDevice(c->P.device);
    fw_learn_opts opt{};
    opt.feed_forward = 1;
    opt.round_size = 1;
    opt.world_size = 1;
    if (opts_in) opt = *opts_in;
    if (opt.world_size < 1) opt.world_size = 1;
    if (opt.rank < 0 || opt.rank >= opt.world_size) return fw_fail(c, FW_ERR_ARG, "fw_learn_network: rank %d outside world of %d", opt.rank, opt.world_size);
    if (opt.world_size > 1 && !allgather) return fw_fail(c, FW_ERR_ARG, "fw_learn_network: world_size > 1 needs an allgather callback");
    if (!c->have_level0) {
        int rc = fw_level0(c, nullptr);
        if (rc) return rc;
    }
    // dense rules + mi_nz: HITON-PC hands test_subsets a row view of the data (prepare_nzdata, hiton.jl:41-50,85,193); the
    // kernels apply it as one more AND plane.  Restored on every exit path by the guard.
    struct ViewGuard {
        fw_ctx *c;
        int old;
        ~ViewGuard() { c->mi_view = old; }
    } view_guard{c, c->mi_view};
    c->mi_view = 1;
    const int p = c->P.p;
    const bool discrete = c->P.kind == FW_MI || c->P.kind == FW_MI_NZ;
    const double t0 = now_s();

    // learning.jl:97-98: ascending univariate degree, stable
    std::vector<int32_t> order(p);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return (c->nb_off[a + 1] - c->nb_off[a]) < (c->nb_off[b + 1] - c->nb_off[b]);
    });
    int nt = p;
    if (opt.max_targets > 0 && opt.max_targets < p) nt = opt.max_targets;

    // per-target directed results (all ranks hold all of them after each round's exchange)
    // directed results of every target in arrival order (a target's entries arrive together, in PC insertion order); the
    // CSR over targets is built once at the end (vectors of vectors cost 120 000 small allocations per cfg4 pass)
    std::vector<int32_t> all_t, all_u;
    std::vector<double> all_s, all_p;
    std::vector<std::vector<int32_t>> adj(p);  // running graph, sorted
    std::vector<uint8_t> adj_dirty((size_t)p, 0);
    std::vector<int32_t> dirty;

    if (c->P.max_k == 0) {  // learning.jl:171-172
        if (int rc = fwi_nb_host_ensure(c)) return rc;
        for (int v = 0; v < p; ++v) {
            const int64_t o = c->nb_off[v];
            const int deg = (int)(c->nb_off[v + 1] - o);
            for (int q = 0; q < deg; ++q) {
                all_t.push_back(v);
                all_u.push_back(c->nb_idx[o + q]);
                all_s.push_back(c->nb_stat[o + q]);
                all_p.push_back(c->nb_p[o + q]);
            }
        }
    } else if (mi_schedule_on_device(c, opt, allgather != nullptr, nt)) {
        // discrete kinds, one GPU, rounds of a few hundred targets or more: the whole schedule stays on the device (whitelists built
        // between the launches, one download at the end) -- same kernel, order and team sizes per round as the loop below
        const int R = (opt.round_size <= 0) ? nt : opt.round_size;
        if (int rc = fwi_devhiton_mi_schedule(c, order.data(), nt, R, opt.feed_forward != 0, all_t, all_u, all_s, all_p)) return rc;
    } else {
        const int R = (opt.round_size <= 0) ? nt : opt.round_size;
        for (int r0 = 0, r1 = 0; r0 < nt; r0 = r1) {
            // R = 1 is the reference's single_il master: job_q_buff_size = 1, so the first TWO targets of the schedule are
            // enqueued up front with an empty whitelist (interleaved.jl:62,76-86); from the third target on a job sees
            // neighbors(graph, T).  The first round therefore holds two targets.
            r1 = std::min(nt, r0 + ((R == 1 && r0 == 0) ? 2 : R));
            // this rank's targets of the round.  The targets of a round are independent of each other (whitelists only change
            // between rounds), so any deal gives the same network; what matters is the balance.  r02 dealt them round-robin in
            // schedule order, and at cfg3 / 8 ranks the heaviest rank carried 1.86e9 of the round's tests against a mean of
            // 1.49e9.  Now: longest-processing-time-first on an estimate of a target's work -- the number of conditioning
            // subsets its candidate list can span, C(deg, <= max_k) ~ deg^max_k (+ a constant for the chain of jobs every
            // target pays) -- heaviest first, each to the least loaded rank, ties to the lower rank.  Every rank computes the
            // same deal from the replicated level-0 lists.
            std::vector<int32_t> owner((size_t)(r1 - r0), 0);
            if (opt.world_size > 1) {
                // the schedule is sorted by ascending degree, and the estimate is monotone in the degree: heaviest first = the
                // round's targets in REVERSE schedule order (no sort; r03's first version sorted with pow() in the comparator:
                // 15 ms per cfg4 round on every rank)
                const int kk = std::min(std::max(c->P.max_k, 1), 3);
                std::vector<double> load((size_t)opt.world_size, 0.0);
                for (int32_t j = r1 - r0 - 1; j >= 0; --j) {
                    const double d = (double)(c->nb_off[order[r0 + j] + 1] - c->nb_off[order[r0 + j]]);
                    const double est = (kk == 1 ? d : kk == 2 ? d * d : d * d * d) + 64.0;
                    int best = 0;
                    for (int w = 1; w < opt.world_size; ++w)
                        if (load[w] < load[best]) best = w;
                    owner[j] = best;
                    load[best] += est;
                }
            }
            size_t n_my = 0;
            for (int i = r0; i < r1; ++i) n_my += (owner[i - r0] == opt.rank);
            // Device-resident rounds (fw_devhiton.hip; every kind but fz_nz): no host round trip per window.  FW_HOST_HITON=1
            // keeps the host pool below for every kind (it is also what rounds of fewer than 64 targets use: the
            // reference's single_il schedule posts one target per round and would pay the device set-up each time).
            // Discrete kinds run as one persistent launch (dh_mi_target_kernel): worth it from a few hundred targets on.
            const char *hh = fw_knob("FW_HOST_HITON");
            const bool host_only = hh && atoi(hh) == 1;
            const bool no_power = c->P.kind == FW_FZ && c->P.n < c->n_obs_min_eff;  // no device work at all
            const char *mt = fw_knob("FW_DEV_MIN_TARGETS");  // test knob
            const size_t min_targets = mt ? (size_t)atol(mt) : ((c->P.kind == FW_FZ || c->P.kind == FW_FZ_NZ) ? 64 : 256);  // cfg2 (1000 targets): 19 ms on the device, 28 ms through the host pool
            const bool stream = c->P.kind == FW_FZ && !c->P.recursive_pcor;  // streamed-column tests: host pool over fw_fzs.hip
            // (discrete data with more than three levels -- the generic form of fw_mi_core.h -- runs through the host job pool as well)
            // fz_nz (r05): device rounds too when its tests run on job-local Float32 matrices (recursive_pcor) and the longest possible
            // list fits the sub-matrix kernel's LDS; FW_NZ_DEV=0 keeps the host pool (A/B, tests)
            bool nz_dev = false;
            if (c->P.kind == FW_FZ_NZ && c->P.recursive_pcor && !(fw_knob("FW_NZ_DEV") && atoi(fw_knob("FW_NZ_DEV")) == 0)) {
                int64_t dmax = 0;
                for (int i = r0; i < r1; ++i) dmax = std::max<int64_t>(dmax, c->nb_off[order[i] + 1] - c->nb_off[order[i]]);
                // (feed-forward: a whitelisted member of the elimination pool is pushed a second time, hiton.jl:24-26 -- a list can reach
                // twice the candidates)
                nz_dev = fwi_fznz_dev_limits(c, (int)(opt.feed_forward ? 2 * dmax : dmax) + 2) == FW_OK && c->P.n >= c->n_obs_min_eff;
            }
            // (conditioning sets of 6 and 7 variables: general-form kernels, host job pool)
            const bool use_dev = !host_only && (c->P.kind != FW_FZ_NZ || nz_dev) && !stream && !no_power && !c->mi_generic && n_my >= min_targets &&
                                 c->P.max_k <= FW_MAX_K_FAST;
            const bool dev_cands = use_dev && c->d_cand != nullptr;  // candidate order already built on the device (fw_bh.hip)
            if (!dev_cands)
                if (int rc = fwi_nb_host_ensure(c)) return rc;
            std::vector<Target> tg;
            tg.reserve(n_my);
            for (int i = r0; i < r1; ++i) {
                if (owner[i - r0] != opt.rank) continue;
                Target t;
                t.T = order[i];
                if (discrete && c->levels[t.T] < 2) {  // hiton.jl:182-184
                    t.phase = 2;
                    tg.push_back(std::move(t));
                    continue;
                }
                // hiton.jl:211-217: candidates with adj p < alpha, stable sort by p
                const int64_t o = c->nb_off[t.T];
                const int deg = (int)(c->nb_off[t.T + 1] - o);
                if (dev_cands) {
                    t.nc_dev = deg;  // every stored neighbour has adj p < alpha; the sorted list lives in c->d_cand
                    if (deg == 0) t.phase = 2;
                } else {
                    std::vector<int32_t> idx;
                    for (int q = 0; q < deg; ++q)
                        if (c->nb_p[o + q] < c->P.alpha) idx.push_back(q);
                    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return c->nb_p[o + a] < c->nb_p[o + b]; });
                    for (int32_t q : idx) t.cands.push_back(c->nb_idx[o + q]);
                    if (t.cands.empty()) t.phase = 2;  // hiton.jl:336-338
                }
                if (opt.feed_forward && !adj[t.T].empty()) {
                    t.wl = adj[t.T].data();  // adj is only modified between rounds
                    t.wl_n = (int)adj[t.T].size();
                }
                tg.push_back(std::move(t));
            }
            bool ran_dev = false;
            std::vector<int32_t> dev_lt, dev_ln;  // device rounds: the round's directed results (target, neighbour, stat, p)
            std::vector<double> dev_ls, dev_lp;
            {
                if (use_dev) {
                    std::vector<FwDhTarget> din(tg.size());
                    for (size_t i = 0; i < tg.size(); ++i) {
                        din[i].T = tg[i].T;
                        if (tg[i].phase != 2) {
                            din[i].cands = tg[i].cands;
                            din[i].nc_dev = tg[i].nc_dev;
                        } else if (dev_cands) {
                            din[i].nc_dev = 0;
                        }
                        din[i].wl = tg[i].wl;
                        din[i].wl_n = tg[i].wl_n;
                    }
                    std::vector<FwDhResult> dres;
                    std::vector<int> chain_of;
                    std::vector<size_t> chain_idx;
                    std::vector<std::vector<FwDhResult>> pres;
                    std::vector<FwDhFlat> pflat;
                    const double tdev0 = now_s();
                    if (fw_knob("FW_TRACE_HOST")) fprintf(stderr, "[fw] round set-up on the host: %.2f ms\n", 1e3 * (tdev0 - t0));
                    // FW_DH_CHAINS = K (default 2, FlashWeave-S): the round's targets are dealt to K independent chains of device
                    // rounds that run concurrently (own host thread, stream and arena each): while one chain is between
                    // two launches (step / plan / fill, the thinning tail of its segment kernel) the other keeps the CUs
                    // busy.  cfg3, ms per pass with 1 / 2 / 3 / 4 chains: 261.8 / 229.4 / 224.2 / 274.3 on one GPU,
                    // 70.5 / 62.4 / 63.7 for one rank of eight; the per-launch duration of the segment kernel grows with
                    // the overlap (224 -> 167 us for launches half the size), which is what HIP events and rocprofv3 see
                    const int dh_chains = [] { const char *e = fw_knob("FW_DH_CHAINS"); return std::min(std::max(e ? atoi(e) : 2, 1), FW_DH_MAX_CHAINS); }();  /* read per round: bench.py times a one-chain pass for the per-kernel figures */  // r02: cfg3 227 / 218 / 268 ms with 2 / 3 / 4 in a bare process, but 226 / 298 under torch.distributed.run and 325 with GPU_MAX_HW_QUEUES=8: the third stream's hardware queue is not ours to choose -> 2
                    static const size_t dh_chain_min = [] { const char *e = fw_knob("FW_DH_CHAIN_MIN"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)48; }();  // r03: 256 -> 48 (one rank of eight holds 98 targets in cfg3's last round: 75 -> 69 ms with two chains)
                    static const int dh_chains_disc = [] { const char *e = fw_knob("FW_DH_CHAINS_DISC"); return std::min(std::max(e ? atoi(e) : 2, 1), FW_DH_MAX_CHAINS); }();  // cfg4: 248.7 / 232.9 / 227.2 / 253.1 ms with 1 / 2 / 3 / 4
                    // discrete kinds run as ONE persistent launch that fills the GPU by itself (dh_mi_target_kernel); concurrent
                    // chains only apply to their level-synchronous form (FW_MI_ROUNDS=1)
                    static const bool mi_rounds = [] { const char *e = fw_knob("FW_MI_ROUNDS"); return e && atoi(e) != 0; }();
                    const int want = c->P.kind == FW_FZ ? dh_chains : (mi_rounds ? dh_chains_disc : 1);
                    const int K = din.size() >= (size_t)want * dh_chain_min ? want : 1;
                    int rc = FW_OK;
                    pres.resize((size_t)K);
                    pflat.resize((size_t)K);
                    if (K == 1) {
                        chain_of.assign(din.size(), 0);
                        chain_idx.resize(din.size());
                        std::iota(chain_idx.begin(), chain_idx.end(), (size_t)0);
                        rc = fwi_devhiton_run(c, din, pres[0], pflat[0]);
                    } else {
                        std::vector<std::vector<FwDhTarget>> part((size_t)K);
                        // which chain target i goes to (and its index there).  Default: dealt in schedule order.  Few targets (the
                        // latency-bound regime: a rank of a multi-GPU job, the last feed-forward round): the heaviest FW_DH_HEAVY_FRAC of
                        // them get chain 0 to themselves -- its launches stay small, so the rounds of the longest chains are short
                        static const int heavy_pct = [] { const char *e = fw_knob("FW_DH_HEAVY_PCT"); return e ? atoi(e) : 0; }();
                        static const size_t heavy_below = [] { const char *e = fw_knob("FW_DH_HEAVY_BELOW"); return e ? (size_t)atol(e) : (size_t)512; }();
                        chain_of.assign(din.size(), 0);
                        chain_idx.assign(din.size(), 0);
                        {
                            std::vector<size_t> cnt((size_t)K, 0);
                            const bool split = K >= 2 && heavy_pct > 0 && din.size() <= heavy_below;
                            const size_t n_heavy = split ? std::max<size_t>(1, din.size() * (size_t)heavy_pct / 100) : 0;
                            for (size_t i = 0; i < din.size(); ++i) {
                                int q;
                                if (split)  // schedule order = ascending degree: the last n_heavy targets are the heaviest
                                    q = i >= din.size() - n_heavy ? 0 : 1 + (int)(i % (size_t)(K - 1));
                                else
                                    q = (int)(i % (size_t)K);
                                chain_of[i] = q;
                                chain_idx[i] = cnt[(size_t)q]++;
                            }
                        }
                        for (size_t i = 0; i < din.size(); ++i) part[(size_t)chain_of[i]].push_back(std::move(din[i]));
                        std::vector<int> rcs((size_t)K, FW_OK);
                        std::vector<std::thread> th;
                        for (int q = 1; q < K; ++q)
                            th.emplace_back([&, q] {
                                (void)hipSetDevice(c->P.device);
                                rcs[q] = fwi_devhiton_run(c, part[q], pres[q], pflat[q], q);
                            });
                        rcs[0] = fwi_devhiton_run(c, part[0], pres[0], pflat[0], 0);
                        for (std::thread &t : th) t.join();
                        for (int q = 0; q < K; ++q)
                            if (rcs[q]) rc = rcs[q];
                    }
                    if (rc) return rc;
                    if (fw_knob("FW_TRACE_HOST")) fprintf(stderr, "[fw] device rounds (all chains): %.2f ms\n", 1e3 * (now_s() - tdev0));
                    // this round's directed results straight from the chains' flat arrays (target i went to chain i % K)
                    {
                        size_t nres = 0;
                        for (size_t i = 0; i < tg.size(); ++i) nres += (size_t)pres[(size_t)chain_of[i]][chain_idx[i]].n;
                        dev_lt.reserve(nres);
                        dev_ln.reserve(nres);
                        dev_ls.reserve(nres);
                        dev_lp.reserve(nres);
                    }
                    for (size_t i = 0; i < tg.size(); ++i) {
                        const FwDhResult &r = pres[(size_t)chain_of[i]][chain_idx[i]];
                        const FwDhFlat &f = pflat[(size_t)chain_of[i]];
                        for (int32_t j = 0; j < r.n; ++j) {
                            dev_lt.push_back(tg[i].T);
                            dev_ln.push_back(f.key[(size_t)r.off + j]);
                            dev_ls.push_back(f.stat[(size_t)r.off + j]);
                            dev_lp.push_back(f.pval[(size_t)r.off + j]);
                        }
                        tg[i].phase = 2;
                    }
                    ran_dev = true;
                }
            }
            if (!ran_dev) {
            // Asynchronous job pool with speculative candidates.  A rejected candidate leaves the accepted set unchanged
            // (hiton.jl:67-70), so during the interleaving phase the next FW_SPEC_DEPTH candidates of a target are posted
            // together against the current accepted set; results are committed strictly in candidate order, and the
            // first acceptance bumps the target's epoch, which cancels / voids everything posted after it.  The sequence
            // of committed (T, candidate, accepted) jobs is therefore exactly the reference's; only the number of
            // latency-bound rounds shrinks.  Every pool round = one window of every in-flight job = ONE kernel launch.
            static const int FW_SPEC_DEPTH = [] { const char *e = fw_knob("FW_SPEC_DEPTH"); return e ? atoi(e) : 8; }();
            static const long FW_SPEC_TARGETS = [] { const char *e = fw_knob("FW_SPEC_TARGETS"); return e ? atol(e) : 512l; }();

            long n_unfinished = (long)tg.size();
            FwPool pool;
            std::vector<int32_t> epoch(tg.size(), 0);
            pool.owner_epoch = &epoch;
            std::vector<FwPoolJob> fin;
            std::vector<int> touched(tg.size());
            std::iota(touched.begin(), touched.end(), 0);
            std::vector<uint8_t> is_touched(tg.size(), 0);
            for (;;) {
                const double ta0 = now_s();
                for (int ti : touched) {
                    Target &t = tg[ti];
                    is_touched[ti] = 0;
                    // commit finished results in candidate order
                    while (advance(c, t)) {
                        int ri = -1;
                        for (size_t q = 0; q < t.ready.size(); ++q)
                            if ((size_t)t.ready[q].first == t.pos) {
                                ri = (int)q;
                                break;
                            }
                        if (ri < 0) break;
                        const FwJobOut o = t.ready[ri].second;
                        t.ready.erase(t.ready.begin() + ri);
                        c->cnt.cond_tests_ref += o.num_tests;
                        c->cnt.subsets_calls += 1;
                        const int32_t cand = t.cands[t.pos];
                        ++t.pos;
                        if (o.pval < c->P.alpha && o.suff_power) {  // issig, tests.jl:1-3; hiton.jl:61-63
                            t.acc.push_back(cand);
                            (t.phase == 0 ? t.TPC : t.PC).set(cand, o.stat, o.pval);
                            ++epoch[ti];  // accepted set changed: later speculative jobs / results are void
                            t.ready.clear();
                            t.posted_end = t.pos;
                        }
                    }
                    if (t.phase == 2) {
                        --n_unfinished;
                        continue;
                    }
                    // post: the current candidate, plus speculative ones while interleaving.  Speculation is only used
                    // once few targets are left (the latency-bound tail); with thousands of active targets the launches
                    // are full anyway and the extra host bookkeeping would cost more than the saved rounds.
                    if (t.posted_end < t.pos) t.posted_end = t.pos;
                    const size_t depth = n_unfinished <= FW_SPEC_TARGETS ? (size_t)FW_SPEC_DEPTH : 1;
                    const size_t limit = t.phase == 0 ? std::min(t.cands.size(), t.pos + depth) : t.pos + 1;
                    while (t.posted_end < limit) {
                        const size_t ci = t.posted_end;
                        if (ci > t.pos && t.in_wl(t.cands[ci])) break;  // a whitelisted candidate will change the accepted set
                        fwi_pool_add(c, pool, t.T, t.cands[ci], t.acc.data(), (int)t.acc.size(), ti);
                        pool.live.back().aux = (int32_t)ci;
                        pool.live.back().epoch = epoch[ti];
                        ++t.posted_end;
                    }
                }
                touched.clear();
                // a speculative job (not the head candidate of its target) only ever runs its first window: most
                // rejections happen within the first few tests, and a voided long job would be pure waste
                for (FwPoolJob &j : pool.live) j.hold = (size_t)j.aux != tg[(size_t)j.tag].pos && j.next > 0;
                c->cnt.t_host_advance_s += now_s() - ta0;
                if (pool.live.empty()) break;
                fin.clear();
                int rc = fwi_pool_round(c, pool, fin);
                if (rc) return rc;
                const double ta1 = now_s();
                for (FwPoolJob &j : fin) {
                    const int ti = (int)j.tag;
                    c->cnt.cond_tests_evaluated += j.out.evaluated;
                    c->cnt.alg_bytes_subsets += fwi_alg_bytes(c, (int)j.acc.size(), j.out.evaluated);
                    if (j.epoch != epoch[ti]) continue;  // posted before an acceptance: void
                    tg[ti].ready.emplace_back(j.aux, j.out);
                    if (!is_touched[ti]) {
                        is_touched[ti] = 1;
                        touched.push_back(ti);
                    }
                }
                c->cnt.t_host_advance_s += now_s() - ta1;
            }
            c->cnt.cond_tests_evaluated += pool.dropped_evaluated;
            c->cnt.alg_bytes_subsets += pool.dropped_alg_bytes;
            }  // host pool
            // exchange this round's directed results (target, neighbour, stat, p)
            std::vector<int32_t> lt = std::move(dev_lt), ln = std::move(dev_ln);
            std::vector<double> ls = std::move(dev_ls), lp = std::move(dev_lp);
            for (Target &t : tg)
                for (size_t i = 0; i < t.PC.key.size(); ++i) {
                    lt.push_back(t.T);
                    ln.push_back(t.PC.key[i]);
                    ls.push_back(t.PC.stat[i]);
                    lp.push_back(t.PC.pval[i]);
                }
            int64_t ntot = (int64_t)lt.size();
            const int32_t *at = lt.data(), *an = ln.data();
            const double *as = ls.data(), *ap = lp.data();
            if (allgather) {  // also with world_size = 1 (the callback then returns what it was given): one code path
                int rc = allgather(user, (int64_t)lt.size(), lt.data(), ln.data(), ls.data(), lp.data(), &ntot, &at, &an, &as, &ap);
                if (rc) return fw_fail(c, FW_ERR_ARG, "fw_learn_network: allgather callback failed (%d)", rc);
            }
            all_t.insert(all_t.end(), at, at + ntot);
            all_u.insert(all_u.end(), an, an + ntot);
            all_s.insert(all_s.end(), as, as + ntot);
            all_p.insert(all_p.end(), ap, ap + ntot);
            // interleaved.jl:136-140 add_edge! (idempotent): both directions appended, the touched lists sorted and
            // de-duplicated once per round (sorted inserts one entry at a time were 15 ms of a 170 ms cfg4 pass)
            // (only the whitelists of later rounds read the running graph: nothing to maintain after the last round or
            // without feed-forward -- cfg4, one round: 380 000 appends and 50 000 sorts for nothing)
            const bool need_adj = opt.feed_forward && r1 < nt;
            for (int64_t i = 0; need_adj && i < ntot; ++i) {
                const int32_t T = at[i], u = an[i];
                adj[T].push_back(u);
                adj[u].push_back(T);
                if (!adj_dirty[T]) adj_dirty[T] = 1, dirty.push_back(T);
                if (!adj_dirty[u]) adj_dirty[u] = 1, dirty.push_back(u);
            }
            for (int32_t v : dirty) {
                std::vector<int32_t> &l = adj[v];
                std::sort(l.begin(), l.end());
                l.erase(std::unique(l.begin(), l.end()), l.end());
                adj_dirty[v] = 0;
            }
            dirty.clear();
        }
    }
    c->cnt.t_cond_s += now_s() - t0;
    if (fw_knob("FW_TRACE_HOST")) fprintf(stderr, "[fw] conditional stage: %.2f ms\n", 1e3 * (now_s() - t0));

    const double tp0 = now_s();
    if (discrete)
        if (int rc = fwi_nb_host_ensure(c)) return rc;
    const double tp1 = now_s();
    if (fw_knob("FW_TRACE_HOST")) fprintf(stderr, "[fw] neighbour lists to the host: %.2f ms\n", 1e3 * (tp1 - tp0));
    // CSR over targets (stable: arrival order inside a target = PC insertion order)
    const size_t ne = all_t.size();
    c->pc_off.assign((size_t)p + 1, 0);
    for (size_t i = 0; i < ne; ++i) c->pc_off[(size_t)all_t[i] + 1]++;
    for (int T = 0; T < p; ++T) c->pc_off[T + 1] += c->pc_off[T];
    c->pc_idx.resize(ne);
    c->pc_w.resize(ne);
    c->pc_p.resize(ne);
    // The passes below walk the directed CSR with data-dependent look-ups: contiguous blocks of variables on a few host threads (kept
    // in the context), every block into its own vectors, concatenated in block order -- the same edge list as the sequential loop.
    const int n_thr = ne < 20000 ? 1 : (int)std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency()));
    if (n_thr > 1 && (!c->host_workers || (int)c->host_workers->th.size() + 1 != n_thr)) {
        fwi_host_workers_free(c);
        c->host_workers = new FwHostWorkers(n_thr);
    }
    std::vector<int> blk((size_t)n_thr + 1, p);
    blk[0] = 0;
    for (int w = 1; w < n_thr; ++w) {  // block w starts where the entries before it reach w / n_thr of the total
        const int64_t want = (int64_t)ne * w / n_thr;
        blk[w] = (int)(std::lower_bound(c->pc_off.begin(), c->pc_off.end(), want) - c->pc_off.begin());
        if (blk[w] > p) blk[w] = p;
    }
    bool blocks_ok = true;  // a block ran out of memory: reported as FW_ERR_NOMEM behind the passes (no exception crosses the C ABI)
    auto run_blocks = [&](const std::function<void(int, int, int)> &fn) {
        if (!blocks_ok) return;
        if (n_thr == 1) {
            try {
                fn(0, 0, p);
            } catch (...) {
                blocks_ok = false;
            }
            return;
        }
        blocks_ok = c->host_workers->run(fn, blk.data());
    };
    // the scatter into the CSR: every block reads the whole arrival list and places the entries of its own targets, in arrival order
    // (one thread: 1.2 ms of random writes for cfg4's 157 000 entries)
    run_blocks([&](int, int lo, int hi) {
        if (lo >= hi) return;
        std::vector<int64_t> fill(c->pc_off.begin() + lo, c->pc_off.begin() + hi);
        for (size_t i = 0; i < ne; ++i) {
            const int32_t T = all_t[i];
            if (T < lo || T >= hi) continue;
            const int64_t d = fill[(size_t)(T - lo)]++;
            c->pc_idx[d] = all_u[i];
            c->pc_w[d] = all_s[i];
            c->pc_p[d] = all_p[i];
        }
    });
    const double tq1 = now_s();
    // misc.jl:137-159 make_weights ("cond_stat"): discrete tests take the sign of the univariate statistic
    if (discrete)
        run_blocks([&](int, int lo, int hi) {
            for (int T = lo; T < hi; ++T) {
                const int64_t o = c->nb_off[T];
                const int deg = (int)(c->nb_off[T + 1] - o);
                const int32_t *b = c->nb_idx.data() + o;
                for (int64_t i = c->pc_off[T]; i < c->pc_off[T + 1]; ++i) {
                    const int32_t *it = std::lower_bound(b, b + deg, c->pc_idx[i]);
                    const double us = (it != b + deg && *it == c->pc_idx[i]) ? c->nb_stat[o + (it - b)] : NAN;
                    const double sg = std::isnan(us) ? NAN : (double)((us > 0) - (us < 0));
                    c->pc_w[i] = sg * std::fabs(c->pc_w[i]);
                }
            }
        });
    const double tq2 = now_s();
    // misc.jl:230-272 make_symmetric_graph (OR rule, maxweight merge, NaN edges dropped)
    c->e_src.clear();
    c->e_dst.clear();
    c->e_w.clear();
    // incoming lists (b -> a for every a) with their weights, ascending in b: the transpose of the CSR by counting sort.  The pass below
    // then reads two contiguous ranges per variable; looking the reverse direction up in b's own list instead cost two or three
    // cache lines from another core per entry (cfg4: 3.1-5.0 ms on 16 / 8 threads for 380 000 entries; r05).
    std::vector<int64_t> in_off((size_t)p + 1, 0);
    std::vector<int32_t> in_idx(ne);
    std::vector<double> in_w(ne);
    {
        for (size_t i = 0; i < ne; ++i) in_off[(size_t)c->pc_idx[i] + 1]++;
        for (int T = 0; T < p; ++T) in_off[T + 1] += in_off[T];
        std::vector<int64_t> fill(in_off.begin(), in_off.end() - 1);
        for (int T = 0; T < p; ++T)  // sources visited in ascending order -> every incoming list comes out sorted
            for (int64_t i = c->pc_off[T]; i < c->pc_off[T + 1]; ++i) {
                const int64_t d = fill[c->pc_idx[i]]++;
                in_idx[(size_t)d] = T;
                in_w[(size_t)d] = c->pc_w[i];
            }
    }
    const double tq3 = now_s();
    std::vector<std::vector<int32_t>> bs((size_t)n_thr), bd((size_t)n_thr);
    std::vector<std::vector<double>> bw((size_t)n_thr);
    std::vector<double> w_t0((size_t)n_thr, 0.0), w_t1((size_t)n_thr, 0.0);
    run_blocks([&](int w, int lo, int hi) {
        w_t0[(size_t)w] = now_s();
        std::vector<int32_t> es, ed;  // block-local, handed over at the end: the headers of bs[w], bs[w + 1] share cache lines and every
        std::vector<double> ew;       // push_back writes one (150 ns per entry on 8 threads; r05)
        const size_t room = (size_t)(c->pc_off[hi] - c->pc_off[lo]);
        es.reserve(room);
        ed.reserve(room);
        ew.reserve(room);
        std::vector<int32_t> out_of((size_t)p, -1);  // out_of[b] == a: the direction a -> b exists
        for (int a = lo; a < hi; ++a) {
            const int32_t *ib = in_idx.data() + in_off[a], *ie = in_idx.data() + in_off[a + 1];
            for (int64_t i = c->pc_off[a]; i < c->pc_off[a + 1]; ++i) {  // direction a -> b exists
                const int32_t b = c->pc_idx[i];
                out_of[(size_t)b] = a;
                if (b <= a) continue;
                const int32_t *it = std::lower_bound(ib, ie, b);
                const double ww = maxweight(c->pc_w[i], (it != ie && *it == b) ? in_w[(size_t)(in_off[a] + (it - ib))] : NAN);
                if (std::isnan(ww)) continue;
                es.push_back(a);
                ed.push_back(b);
                ew.push_back(ww);
            }
            for (int64_t q = in_off[a]; q < in_off[a + 1]; ++q) {  // only b -> a exists
                const int32_t b = in_idx[(size_t)q];
                if (b <= a || out_of[(size_t)b] == a) continue;
                const double ww = maxweight(in_w[(size_t)q], NAN);
                if (std::isnan(ww)) continue;
                es.push_back(a);
                ed.push_back(b);
                ew.push_back(ww);
            }
        }
        bs[(size_t)w] = std::move(es);
        bd[(size_t)w] = std::move(ed);
        bw[(size_t)w] = std::move(ew);
        w_t1[(size_t)w] = now_s();
    });
    const double tq4 = now_s();
    if (!blocks_ok) return fw_fail(c, FW_ERR_NOMEM, "weights / symmetric graph: a host block ran out of memory");
    for (int w = 0; w < n_thr; ++w) {
        c->e_src.insert(c->e_src.end(), bs[(size_t)w].begin(), bs[(size_t)w].end());
        c->e_dst.insert(c->e_dst.end(), bd[(size_t)w].begin(), bd[(size_t)w].end());
        c->e_w.insert(c->e_w.end(), bw[(size_t)w].begin(), bw[(size_t)w].end());
    }
    c->have_network = true;
    if (fw_knob("FW_TRACE_HOST"))
        fprintf(stderr, "[fw] weights + symmetric graph on the host: %.2f ms (directed CSR %.2f, signs %.2f, transpose %.2f, edges %.2f; %d threads)\n",
                1e3 * (now_s() - tp1), 1e3 * (tq1 - tp1), 1e3 * (tq2 - tq1), 1e3 * (tq3 - tq2), 1e3 * (now_s() - tq3), n_thr);
    if (fw_knob("FW_TRACE_HOST")) {
        double d0 = 0, d1 = 0, lmax = 0;
        for (int w = 0; w < n_thr; ++w) {
            d0 = std::max(d0, w_t0[(size_t)w] - tq3);
            d1 = std::max(d1, w_t1[(size_t)w] - w_t0[(size_t)w]);
            lmax = std::max(lmax, tq4 - w_t1[(size_t)w]);
        }
        fprintf(stderr, "[fw]   edges pass: latest start %.3f ms after the call, longest block %.3f ms, return %.3f ms, concatenation %.3f ms (ne %zu)\n",
                1e3 * d0, 1e3 * d1, 1e3 * (tq4 - tq3), 1e3 * (now_s() - tq4), ne);
    }
#ifdef FW_FZ_FASTDBG
    fwi_fz_fastdbg_print();
#endif
    if (n_edges_out) *n_edges_out = (int64_t)c->e_src.size();
    return FW_OK;
}

extern "C" int fw_network_get(const fw_ctx *c, int32_t *src, int32_t *dst, double *weight)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    if (!c->have_network) return fw_fail(c, FW_ERR_STATE, "fw_network_get: fw_learn_network has not run");
    const size_t k = c->e_src.size();
    if (k) {
        if (src) memcpy(src, c->e_src.data(), sizeof(int32_t) * k);
        if (dst) memcpy(dst, c->e_dst.data(), sizeof(int32_t) * k);
        if (weight) memcpy(weight, c->e_w.data(), sizeof(double) * k);
    }
    return FW_OK;
}

extern "C" int fw_network_get_directed(const fw_ctx *c, int64_t *off, int32_t *idx, double *weight, double *pval)
{
    if (!c) return fw_fail(nullptr, FW_ERR_ARG, "NULL context");
    if (!c->have_network) return fw_fail(c, FW_ERR_STATE, "fw_network_get_directed: fw_learn_network has not run");
    if (off) memcpy(off, c->pc_off.data(), sizeof(int64_t) * c->pc_off.size());
    const size_t k = c->pc_idx.size();
    if (k) {
        if (idx) memcpy(idx, c->pc_idx.data(), sizeof(int32_t) * k);
        if (weight) memcpy(weight, c->pc_w.data(), sizeof(double) * k);
        if (pval) memcpy(pval, c->pc_p.data(), sizeof(double) * k);
    }
    return FW_OK;
}
