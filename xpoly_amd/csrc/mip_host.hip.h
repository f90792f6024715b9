// Host controller of xpoly's branch-and-bound MIP<Mat,T> (src/com/lpsol.h:2087-2702),
// of Lineq::has_solution (src/com/linsys.cpp:830-906) and of the DepPoly::is_empty front
// end (src/eng/poly.cpp:530-573).
//
// The tree walk is the reference's depth-first recursion -- its results depend on DFS order
// through the shared fork_count row and the incumbent (lpsol.h:2474-2497) -- written as an
// explicit stack machine per problem so that MANY problems advance in lock step: in every
// round each unfinished problem contributes the LP relaxation of its current node, nodes of
// equal shape are solved by ONE launch of the LDS-resident batch kernel, and the host feeds
// the answers back into the stack machines. A node is a from-scratch SIX solve with
// max_iter = 10000 (lpsol.h:2441), exactly as in the reference.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>
#include "six_host.hip.h"
#include "lineq_shared.hip.h"
#include "mip_kernels.hip.h"

namespace xpg {

template <class S> struct MipProblem {
    int cols;
    std::vector<S> tgtf, vc, eq, leq;      // flat row-major; vc has cols-1 rows
    int eq_rows, leq_rows;
};

template <class S>
MipProblem<S> make_problem(const S * tgtf, const S * vc, int vc_rows, const S * eqs, int eq_rows, const S * leq,
                           int leq_rows, int cols)
{
    MipProblem<S> Q;
    Q.cols = cols; Q.eq_rows = eq_rows; Q.leq_rows = leq_rows;
    Q.tgtf.assign(tgtf, tgtf + cols);
    Q.vc.assign(vc, vc + (size_t)vc_rows * cols);
    if (eq_rows) Q.eq.assign(eqs, eqs + (size_t)eq_rows * cols);
    if (leq_rows) Q.leq.assign(leq, leq + (size_t)leq_rows * cols);
    return Q;
}

inline bool int_cast_ok(F64) { return true; }
inline bool int_cast_ok(R32 a) { return a.den != 0; }

// One problem's MIP::RecusivePart (lpsol.h:2427-2612) as a resumable stack machine.
template <class S> struct MipTask {
    struct Frame {
        MipProblem<S> Q;
        int stage;                          // 0: LP pending, 1: floor child running, 2: ceiling child running
        int col, lo, hi;
        bool kept; S kept_v; std::vector<S> kept_sol;
    };
    bool is_max, is_bin;
    const uint8_t * allow_rational;         // 1 x cols or null (lpsol.h:2369-2393)
    int rhs0;
    // the by-reference state the reference threads through its recursion
    S v; std::vector<S> sol;
    bool have_best; S best_v; std::vector<S> best_sol;
    std::vector<int> forks;
    std::vector<Frame> stack;
    NormalForm<S> F;                        // normal form of the pending node LP
    bool done; int final_status; long nodes;

    void start(const MipProblem<S> & root, bool mx, bool bin, const uint8_t * allow)
    {
        is_max = mx; is_bin = bin; allow_rational = allow; rhs0 = root.cols - 1;
        v = zero<S>(); sol.clear(); have_best = false; best_v = zero<S>(); best_sol.clear();
        forks.assign(root.cols, 0);
        stack.clear(); done = false; final_status = 0; nodes = 0;
        push(root);
    }
    void push(const MipProblem<S> & Q)
    {
        Frame f; f.Q = Q; f.stage = 0; f.col = 0; f.lo = 0; f.hi = 1; f.kept = false; f.kept_v = zero<S>();
        stack.push_back(f);
    }
    // Normalises the pending node; a negative return is the LP's "status" (shape / undefined).
    int prepare()
    {
        const MipProblem<S> & Q = stack.back().Q;
        nodes++;
        return normalize_host(Q.tgtf.data(), Q.vc.data(), Q.cols - 1, Q.eq_rows ? Q.eq.data() : (const S *)0, Q.eq_rows,
                              Q.leq_rows ? Q.leq.data() : (const S *)0, Q.leq_rows, Q.cols, F);
    }

    // MIP::is_satisfying (lpsol.h:2364-2408); `col` is the first offending entry.
    bool satisfied(std::vector<S> & s, int & col) const
    {
        for (size_t j = 0; j < s.size(); j++) {
            if (allow_rational || is_bin) reduce(s[j]);
            if (allow_rational) {
                if (allow_rational[j]) continue;
                if (!is_int(s[j])) { col = (int)j; return false; }
                if (is_bin && ne(s[j], zero<S>()) && ne(s[j], one<S>())) { col = (int)j; return false; }
            } else if (is_bin) {
                if (ne(s[j], zero<S>()) && ne(s[j], one<S>())) { col = (int)j; return false; }
            } else if (!is_int(s[j])) { col = (int)j; return false; }     // {R,Float}Mat::is_imat
        }
        return true;
    }
    void remember()
    {
        if (!have_best || (is_max ? lt(best_v, v) : gt(best_v, v))) { best_sol = sol; best_v = v; have_best = true; }
    }
    static void add_row(std::vector<S> & rows, int & nrows, int cols, int col, S coef, int rhs, S b)
    {
        rows.resize((size_t)(nrows + 1) * cols, zero<S>());
        for (int j = 0; j < cols; j++) rows[(size_t)nrows * cols + j] = zero<S>();
        rows[(size_t)nrows * cols + col] = coef;
        rows[(size_t)nrows * cols + rhs] = b;
        nrows++;
    }
    // The child of frame `pi`: its problem plus one bound row. Built in place at the top of the stack -- ONE copy of
    // the parent's problem (by index: the emplace may move the frames).
    void push_branch(size_t pi, bool ceiling)
    {
        stack.emplace_back();
        const Frame & p = stack[pi];
        Frame & c = stack.back();
        c.stage = 0; c.col = 0; c.lo = 0; c.hi = 1; c.kept = false; c.kept_v = zero<S>();
        c.Q = p.Q;
        MipProblem<S> & B = c.Q;
        if (is_bin) add_row(B.eq, B.eq_rows, B.cols, p.col, one<S>(), rhs0, S::from_int(ceiling ? p.hi : p.lo));   // lpsol.h:2506-2512, :2548-2553
        else if (!ceiling) add_row(B.leq, B.leq_rows, B.cols, p.col, one<S>(), rhs0, S::from_int(p.lo));          // :2514-2520
        else add_row(B.leq, B.leq_rows, B.cols, p.col, minus_one<S>(), rhs0, S::from_int(-p.hi));                  // :2555-2559
    }

    // Feeds the answer of the pending LP (st: SIX status or a negative error; y: raw values of
    // the normalised variables on success) and runs until the next LP is needed or the tree ends.
    void on_lp(int st, const std::vector<S> & y)
    {
        int ret;
        {
            Frame & f = stack.back();
            v = zero<S>();
            if (st == XPG_SIX_SUCC) {
                std::vector<S> s(f.Q.cols);
                finish_host(F, f.Q.tgtf.data(), y, &v, s.data());
                sol = s;
            }
            if (st < 0) ret = st;
            else if (st == XPG_SIX_UNBOUND) ret = XPG_IP_UNBOUND;
            else if (st == XPG_SIX_TIME_OUT) ret = XPG_ERR_REF_UNDEFINED;     // UNREACH() in the reference
            else if (st != XPG_SIX_SUCC) ret = XPG_IP_NO_PRI_FEASIBLE_SOL;
            else {
                int col = 0;
                if (satisfied(sol, col)) ret = XPG_IP_SUCC;
                else if (have_best && (is_max ? le(v, best_v) : ge(v, best_v))) ret = XPG_IP_NO_BETTER_THAN_BEST_SOL;
                else if (forks[col] >= 1) ret = XPG_IP_NO_PRI_FEASIBLE_SOL;                  // lpsol.h:2486-2496
                else if (!is_bin && !int_cast_ok(sol[col])) ret = XPG_ERR_REF_UNDEFINED;
                else {
                    forks[col]++;
                    f.col = col; f.lo = 0; f.hi = 1;
                    if (!is_bin) { f.lo = to_int(sol[col]); f.hi = f.lo + 1; }
                    f.stage = 1;
                    push_branch(stack.size() - 1, false);
                    return;
                }
            }
        }
        for (;;) {                                      // hand `ret` to the callers up the stack
            stack.pop_back();
            if (stack.empty()) { final_status = ret; done = true; return; }
            Frame & p = stack.back();
            if (ret < 0) continue;
            if (p.stage == 1) {                         // floor branch came back, lpsol.h:2527-2543
                if (ret == XPG_IP_SUCC) { p.kept_sol = sol; p.kept_v = v; p.kept = true; remember(); }
                p.stage = 2;
                push_branch(stack.size() - 1, true);
                return;
            }
            if (ret == XPG_IP_SUCC) {                   // ceiling branch came back, lpsol.h:2563-2611
                if (p.kept && (is_max ? gt(p.kept_v, v) : lt(p.kept_v, v))) { v = p.kept_v; sol = p.kept_sol; }
                remember();
            } else if (p.kept) { v = p.kept_v; sol = p.kept_sol; remember(); ret = XPG_IP_SUCC; }
        }
    }
};

// The host half of a lock-step round -- normalising every tree's pending node (equality substitution, dual
// construction: O(rows x cols^2) exact operations each) and feeding the answers back into the stack machines --
// is independent per tree, and at a few nodes per tree it costs more than the node-batch launches (1024 knapsacks
// of 24 variables on one host thread: prepare 7.9 ms + feed-back 8.6 ms against 8 ms for 15 launches). A small pool
// of persistent host threads (XPG_HOST_THREADS; default 1, i.e. off) can take both loops with STATIC shares --
// worker w always gets the same slice of the index range, so a tree's heap blocks are allocated and freed by one
// thread and stay in one core's cache (handing out chunks dynamically made the loops SLOWER than one thread on a
// 256-core host: 12 + 18 ms) -- and the workers spin for a moment before they sleep: a round has three loops a
// fraction of a millisecond apart. Measured with 1 / 4 / 8 / 16 / 32 threads on one box: 37 / 48 / 38 / 43 / 36 k
// MIPs/s, on another 39 k with 1 and 30 k with 4: loops of half a millisecond do not pay for waking threads on a
// 256-core host reliably, so the default is the calling thread alone.
class MipPool {
    std::vector<std::thread> th_;
    std::mutex m_, run_m_;
    std::condition_variable cv_;
    std::function<void(int)> job_;                      // job_(w): the share of worker w, 0 <= w < size()
    std::atomic<unsigned> gen_{0};
    std::atomic<int> pending_{0};
    std::atomic<bool> stop_{false};
    static void relax() { __builtin_ia32_pause(); }
    void worker(int w)
    {
        unsigned seen = 0;
        for (;;) {
            for (int spin = 0; spin < 20000 && gen_.load(std::memory_order_acquire) == seen; spin++) relax();
            if (gen_.load(std::memory_order_acquire) == seen) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_.load() || gen_.load(std::memory_order_acquire) != seen; });
            }
            if (stop_.load()) return;
            seen = gen_.load(std::memory_order_acquire);
            job_(w);
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
public:
    MipPool()
    {
        unsigned nt = 1;                                    // the calling thread alone unless XPG_HOST_THREADS says otherwise (see above)
        if (const char * e = xpg_env("XPG_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) nt = (unsigned)v; }
        for (unsigned w = 1; w < nt; w++) th_.emplace_back([this, w] { worker((int)w); });
    }
    ~MipPool()
    {
        { std::unique_lock<std::mutex> lk(m_); stop_.store(true); gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (auto & t : th_) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    // f(i) for every i of [0, n): worker w takes the w-th of size() contiguous slices; the caller is worker 0
    void run(size_t n, const std::function<void(size_t)> & f)
    {
        // one caller at a time: a second controller (the _multi entry points run one per device) keeps its loop to itself
        std::unique_lock<std::mutex> mine(run_m_, std::try_to_lock);
        if (n < 128 || th_.empty() || !mine.owns_lock()) { for (size_t i = 0; i < n; i++) f(i); return; }
        const int nt = size();
        std::function<void(int)> share = [&, n, nt](int w) {
            const size_t lo = n * (size_t)w / (size_t)nt, hi = n * (size_t)(w + 1) / (size_t)nt;
            for (size_t i = lo; i < hi; i++) f(i);
        };
        {
            std::unique_lock<std::mutex> lk(m_);
            job_ = share;
            pending_.store(nt - 1, std::memory_order_release);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        share(0);
        while (pending_.load(std::memory_order_acquire) != 0) relax();
    }
};
inline MipPool & mip_pool() { static MipPool p; return p; }
template <class F> inline void mip_parallel_for(size_t n, F f) { mip_pool().run(n, std::function<void(size_t)>(f)); }

// Advances every task to completion; node LPs of equal shape share one kernel launch.
template <class S> int run_mip_tasks(xpg_ctx * ctx, int kind, std::vector<MipTask<S> > & tasks)
{
    struct Key { int is_max, rows, cols; bool operator<(const Key & o) const
        { return is_max != o.is_max ? is_max < o.is_max : (rows != o.rows ? rows < o.rows : cols < o.cols); } };
    const std::vector<S> none;
    static const bool dbg = xpg_hook("XPG_MIP_DEBUG") != 0;
    int rounds = 0, launches = 0; double t_prep = 0, t_gpu = 0, t_feed = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
        const double t0 = now();
        std::map<Key, std::vector<int> > groups;
        std::vector<int> large;
        bool any = false;
        mip_parallel_for(tasks.size(), [&](size_t t) {
            MipTask<S> & T = tasks[t];
            while (!T.done) {
                const int rc = T.prepare();
                if (rc == 0) break;
                T.on_lp(rc, none);                      // malformed / reference-undefined node: no GPU work
            }
        });
        for (size_t t = 0; t < tasks.size(); t++) {
            MipTask<S> & T = tasks[t];
            if (T.done) continue;
            any = true;
            if (T.F.fits_lds(T.is_max)) { Key k = { T.is_max ? 1 : 0, T.F.rows, T.F.n + 1 }; groups[k].push_back((int)t); }
            else large.push_back((int)t);
        }
        t_prep += now() - t0;
        if (!any) {
            if (dbg) fprintf(stderr, "xpoly_amd: MIP controller: %d rounds, %d node-batch launches; host prepare %.1f ms, batches %.1f ms, feed-back %.1f ms\n",
                             rounds, launches, t_prep, t_gpu, t_feed);
            return 0;
        }
        rounds++;
        for (typename std::map<Key, std::vector<int> >::iterator g = groups.begin(); g != groups.end(); ++g) {
            const Key & k = g->first;
            const std::vector<int> & ids = g->second;
            const int nb = (int)ids.size();
            // the round's node LPs are packed straight into the context's pinned staging and the answers are read
            // from it: one copy each way per launch (batch_staged, batch_kernels.hip.h)
            BatchStage<S> bs;
            int rc = batch_stage_prepare<S>(ctx, nb, k.rows, k.cols, bs);
            if (rc) return rc;
            mip_parallel_for((size_t)nb, [&](size_t b) {
                const NormalForm<S> & F = tasks[ids[b]].F;
                S * tg = bs.h_tgtf + b * k.cols;
                S * lq = bs.h_leq + b * (size_t)k.rows * k.cols;
                for (int j = 0; j < k.cols; j++) tg[j] = F.obj[j];
                for (size_t e = 0; e < (size_t)F.rows * (size_t)(F.n + 1); e++) lq[e] = F.Np[e];      // (its own cells, or a view of the node's inequalities: six_host.hip.h)
            });
            const double t1 = now();
            rc = batch_stage_run<S>(ctx, bs, k.is_max, 10000u, /*raw_sol=*/1);
            if (rc) return rc;
            launches++;
            const double t2 = now();
            t_gpu += t2 - t1;
            mip_parallel_for((size_t)nb, [&](size_t b) {
                const S * raw = bs.h_sol + b * k.cols;
                std::vector<S> y(raw, raw + (k.cols - 1));
                tasks[ids[b]].on_lp(bs.h_st[b], y);
            });
            t_feed += now() - t2;
        }
        for (size_t q = 0; q < large.size(); q++) {
            MipTask<S> & T = tasks[large[q]];
            std::vector<S> y;
            const int st = solve_large(ctx, kind, T.is_max, T.F, (const S *)0, (const S *)0, 10000u, y);
            T.on_lp(st, y);
        }
    }
}

template <class S> inline bool mip_device_fits(int leq_rows, int cols, bool is_bin, int eq_rows = 0);
template <class S>
int mip_batch_device(xpg_ctx * ctx, int nb, bool is_max, bool is_bin, const S * tgtf, const S * leq, int leq_rows,
                     int cols, int32_t * out_status, S * out_v, S * out_sol, long long * out_nodes,
                     const uint8_t * allow_rational, const S * eqs, int eq_rows);

// MIP::maxm / minm (lpsol.h:2636-2657, :2681-2702).
template <class S>
int mip_solve(xpg_ctx * ctx, int kind, bool is_max, bool is_bin, const S * tgtf, const S * vc, int vc_rows,
              const S * eqs, int eq_rows, const S * leq, int leq_rows, int cols, const uint8_t * allow_rational,
              S * out_v, S * out_sol, long * out_nodes)
{
    if (!ctx || !tgtf || !vc || !out_v || cols < 2 || vc_rows != cols - 1 || eq_rows < 0 || leq_rows < 0 ||
        (eq_rows == 0 && leq_rows == 0) || (eq_rows > 0 && !eqs) || (leq_rows > 0 && !leq))
        return XPG_ERR_SHAPE;
    // x >= 0 (vc = -I) with inequalities and / or equalities at the root, with or without a rational_indicator, is what
    // the device tree walk takes; free or otherwise bounded variables stay with the host controller
    static const bool on_device = [] { const char * e = xpg_env("XPG_MIP_DEVICE"); return !(e && e[0] == '0'); }();
    if (on_device && mip_device_fits<S>(leq_rows, cols, is_bin, eq_rows)) {
        bool plain = true;
        for (int i = 0; i < vc_rows && plain; i++)
            for (int j = 0; j < cols && plain; j++)
                plain = eq(vc[(size_t)i * cols + j], i == j ? minus_one<S>() : zero<S>());
        if (plain) {
            int32_t st = 0; long long nodes = 0;
            std::vector<S> sol((size_t)cols, zero<S>());
            if (out_sol) for (int j = 0; j < cols; j++) sol[(size_t)j] = out_sol[j];
            const int rc = mip_batch_device<S>(ctx, 1, is_max, is_bin, tgtf, leq, leq_rows, cols, &st, out_v, sol.data(), &nodes, allow_rational, eqs, eq_rows);
            if (rc != XPG_ERR_UNSUPPORTED) {
                if (rc) return rc;
                if (st == XPG_IP_SUCC && out_sol) for (int j = 0; j < cols; j++) out_sol[j] = sol[(size_t)j];
                if (out_nodes) *out_nodes = (long)nodes;
                return st;
            }
        }
    }
    std::vector<MipTask<S> > tasks(1);
    tasks[0].start(make_problem(tgtf, vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols), is_max, is_bin, allow_rational);
    int rc = run_mip_tasks(ctx, kind, tasks);
    if (rc) return rc;
    const MipTask<S> & T = tasks[0];
    *out_v = T.v;
    if (T.final_status == XPG_IP_SUCC && out_sol && (int)T.sol.size() == cols)
        for (int j = 0; j < cols; j++) out_sol[j] = T.sol[j];
    if (out_nodes) *out_nodes = T.nodes;
    return T.final_status;
}

// Whether the node LPs of the deepest path fit the device tree walk's LDS budget, maximising and minimising.
// Rows of the largest node LP: the root's inequalities, one bound row per ancestor under integer branching, and -- with
// equalities at the root -- two rows for every equality convertEq2Ineq may leave unsubstituted (the root's, and under
// 0-1 branching one per ancestor).
inline int mip_rmax(int leq_rows, int eq_rows, int n, bool is_bin)
{
    int r = leq_rows + (is_bin ? 0 : n);
    if (eq_rows > 0) r += 2 * (eq_rows + (is_bin ? n : 0));
    return r;
}
template <class S> inline bool mip_device_fits(int leq_rows, int cols, bool is_bin, int eq_rows)
{
    const int n = cols - 1, rmax = mip_rmax(leq_rows, eq_rows, n, is_bin);
    if (rmax <= 0 || eq_rows + n + 2 > MIP_EQ_MAX) return false;
    return small_lds_bytes<S>(rmax, n) <= 64 * 1024 && small_lds_bytes<S>(n, rmax) <= 64 * 1024;
}
// Launch shape of k_mip_tree for nb trees whose node LPs have at most rmax rows and n variables.
struct MipGeom { size_t lds; int threads, grid; };
template <class S> inline MipGeom mip_geom(const xpg_ctx * ctx, int nb, int rmax, int n, bool is_max)
{
    MipGeom g;
    const int R = is_max ? rmax : n, V = is_max ? n : rmax;
    g.lds = small_lds_bytes<S>(R, V);
    const int cells = R * (V + R + 2), cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    g.threads = cells >= 2048 ? 256 : (cells >= 1024 ? 128 : 64);
    // more trees than the chip holds at that width: one wave per tree, more trees in flight (8192 knapsacks of 24
    // variables: 64 / 128 / 256 threads 623 k / 425 k / 318 k MIPs/s; at 1024, where the deepest tree decides, 163 / 171 / 170 k)
    if (nb >= 8 * cus) g.threads = 64;
    if (const char * t = xpg_hook("XPG_BATCH_THREADS")) { const int v = atoi(t); if (v >= 64 && v <= 256 && v % 64 == 0) g.threads = v; }
    const int per_cu = (int)((160 * 1024) / g.lds) > 0 ? (int)((160 * 1024) / g.lds) : 1;
    g.grid = cus * (per_cu > 8 ? 8 : per_cu) * 4;
    if (g.grid > nb) g.grid = nb;
    return g;
}
// The same batch with the tree walks on the device (mip_kernels.hip.h): one workgroup per problem. Returns
// XPG_ERR_UNSUPPORTED where a node LP of the deepest path would not fit the LDS budget -- the caller then takes the
// host controller below.
template <class S>
int mip_batch_device(xpg_ctx * ctx, int nb, bool is_max, bool is_bin, const S * tgtf, const S * leq, int leq_rows,
                     int cols, int32_t * out_status, S * out_v, S * out_sol, long long * out_nodes,
                     const uint8_t * allow_rational, const S * eqs, int eq_rows)
{
    const int n = cols - 1;
    const int rmax = mip_rmax(leq_rows, eq_rows, n, is_bin);
    const int depth = n + 2;
    const MipGeom g = mip_geom<S>(ctx, nb, rmax, n, is_max);
    if (g.lds > 64 * 1024) return XPG_ERR_UNSUPPORTED;
    const size_t lds = g.lds;
    const int threads = g.threads, grid = g.grid;
    const size_t ws_words = mip_ws_words(rmax, cols, depth);
    const size_t bl = (size_t)nb * leq_rows * cols * 8, bt = (size_t)nb * cols * 8, be = (size_t)nb * eq_rows * cols * 8;
    DevBuf dl, dt, dws, dst, dv, dsol, dn, dal, de;
    if (eq_rows > 0) {
        XPG_TRY(de.alloc(ctx, be));
        XPG_TRY(hipMemcpyAsync(de.p, eqs, be, hipMemcpyHostToDevice, ctx->stream));
    }
    if (allow_rational) {
        XPG_TRY(dal.alloc(ctx, (size_t)cols));
        XPG_TRY(hipMemcpyAsync(dal.p, allow_rational, (size_t)cols, hipMemcpyHostToDevice, ctx->stream));
    }
    // Speculative ceiling children (mip_kernels.hip.h, SP_*): for batches that leave the chip under-filled -- one tree per
    // walking workgroup, the batch lasts as long as its deepest tree -- helper workgroups behind the walkers solve the node
    // LPs the walks will need next. Not with root equalities (the helper builds plain nodes only). XPG_MIP_SPEC=0: off.
    static const int spec_env = [] { const char * e = xpg_hook("XPG_MIP_SPEC"); return e ? atoi(e) : 1; }();
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const bool spec = spec_env != 0 && eq_rows == 0 && grid == nb && nb <= 8 * cus;
    const int nhelp = spec ? (spec_env > 1 ? spec_env : cus) : 0;        // (256 / 512 / 2048 helpers measured alike: 3.72 / 3.77 / 3.86 ms for 1024 knapsacks, 4.41 without)
    DevBuf dq;
    if (spec) { XPG_TRY(dq.alloc(ctx, spq_bytes())); XPG_TRY(hipMemsetAsync(dq.p, 0, spq_bytes(), ctx->stream)); }
    XPG_TRY(dl.alloc(ctx, bl)); XPG_TRY(dt.alloc(ctx, bt)); XPG_TRY(dws.alloc(ctx, (size_t)(grid + nhelp) * ws_words * 8));
    XPG_TRY(dst.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dv.alloc(ctx, (size_t)nb * 8)); XPG_TRY(dsol.alloc(ctx, bt));
    XPG_TRY(dn.alloc(ctx, (size_t)nb * 4));
    if (bl) XPG_TRY(hipMemcpyAsync(dl.p, leq, bl, hipMemcpyHostToDevice, ctx->stream));
    XPG_TRY(hipMemcpyAsync(dt.p, tgtf, bt, hipMemcpyHostToDevice, ctx->stream));
    if (out_sol) XPG_TRY(hipMemcpyAsync(dsol.p, out_sol, bt, hipMemcpyHostToDevice, ctx->stream));
    XPG_TRY(lds_limit((const void *)k_mip_tree<S>, ctx->device, lds));
    hipLaunchKernelGGL((k_mip_tree<S>), dim3(grid + nhelp), dim3(threads), lds, ctx->stream, nb, (const S *)dt.p, (const S *)dl.p,
                       leq_rows, cols, is_max ? 1 : 0, is_bin ? 1 : 0, rmax, depth, (unsigned long long *)dws.p, ws_words,
                       (int32_t *)dst.p, (S *)dv.p, out_sol ? (S *)dsol.p : (S *)0, (int *)dn.p, (const int *)0, (const int *)0,
                       allow_rational ? (const uint8_t *)dal.p : (const uint8_t *)0, eq_rows > 0 ? (const S *)de.p : (const S *)0, eq_rows,
                       grid, spec ? (int *)dq.p : (int *)0);
    XPG_TRY(hipGetLastError());
    if (spec && xpg_hook("XPG_MIP_DEBUG")) {
        int hq[4] = {0, 0, 0, 0};
        XPG_TRY(hipMemcpyAsync(hq, dq.p, sizeof(hq), hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "xpoly_amd: MIP tree walk, %d trees, %d helpers: %d ceiling children requested, %d answers taken parked\n", nb, nhelp, hq[0], hq[3]);
    }
    std::vector<int32_t> nodes((size_t)nb);
    XPG_TRY(hipMemcpyAsync(out_status, dst.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(out_v, dv.p, (size_t)nb * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_sol) XPG_TRY(hipMemcpyAsync(out_sol, dsol.p, bt, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(nodes.data(), dn.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    if (out_nodes) { long long t = 0; for (int b = 0; b < nb; b++) t += nodes[(size_t)b]; *out_nodes = t; }
    return 0;
}

// nb independent MIPs of one shape (x >= 0, inequalities only), advanced together.
template <class S>
int mip_batch(xpg_ctx * ctx, int kind, int nb, bool is_max, bool is_bin, const S * tgtf, const S * leq, int leq_rows,
              int cols, int32_t * out_status, S * out_v, S * out_sol, long long * out_nodes)
{
    if (!ctx || nb < 0 || !tgtf || !leq || leq_rows <= 0 || cols < 2 || !out_status || !out_v) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    // the whole tree walk on the device where the node LPs fit (XPG_MIP_DEVICE=0: the host controller, for A/B runs)
    static const bool on_device = [] { const char * e = xpg_env("XPG_MIP_DEVICE"); return !(e && e[0] == '0'); }();
    if (on_device) {
        const int rc = mip_batch_device<S>(ctx, nb, is_max, is_bin, tgtf, leq, leq_rows, cols, out_status, out_v, out_sol, out_nodes, (const uint8_t *)0, (const S *)0, 0);
        if (rc != XPG_ERR_UNSUPPORTED) return rc;
    }
    const int rhs = cols - 1;
    std::vector<S> vc((size_t)rhs * cols, zero<S>());
    for (int i = 0; i < rhs; i++) vc[(size_t)i * cols + i] = minus_one<S>();
    std::vector<MipTask<S> > tasks(nb);
    for (int b = 0; b < nb; b++)
        tasks[b].start(make_problem<S>(tgtf + (size_t)b * cols, vc.data(), rhs, (const S *)0, 0,
                                       leq + (size_t)b * leq_rows * cols, leq_rows, cols), is_max, is_bin, (const uint8_t *)0);
    int rc = run_mip_tasks<S>(ctx, kind, tasks);
    if (rc) return rc;
    long long nodes = 0;
    for (int b = 0; b < nb; b++) {
        const MipTask<S> & T = tasks[b];
        out_status[b] = T.final_status;
        out_v[b] = T.v;
        nodes += T.nodes;
        if (T.final_status == XPG_IP_SUCC && out_sol && (int)T.sol.size() == cols)
            for (int j = 0; j < cols; j++) out_sol[(size_t)b * cols + j] = T.sol[j];
    }
    if (out_nodes) *out_nodes = nodes;
    return 0;
}

// nb independent MIPs of one shape WITH equalities at the root (x >= 0; the shape PolyTran::FeaSchedule passes,
// src/eng/poly.cpp:5118-5130, batched): leq may be NULL with leq_rows = 0. The device tree walk where the node LPs fit
// (every node runs convertEq2Ineq over the root's and the branches' equalities in its workgroup), the host controller
// -- problems advancing in lock step -- otherwise.
template <class S>
int mip_batch_eq(xpg_ctx * ctx, int kind, int nb, bool is_max, bool is_bin, const S * tgtf, const S * leq, int leq_rows,
                 const S * eqs, int eq_rows, int cols, int32_t * out_status, S * out_v, S * out_sol, long long * out_nodes)
{
    if (!ctx || nb < 0 || !tgtf || eq_rows <= 0 || !eqs || leq_rows < 0 || (leq_rows > 0 && !leq) || cols < 2 || !out_status || !out_v)
        return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    static const bool on_device = [] { const char * e = xpg_env("XPG_MIP_DEVICE"); return !(e && e[0] == '0'); }();
    if (on_device && mip_device_fits<S>(leq_rows, cols, is_bin, eq_rows)) {
        const int rc = mip_batch_device<S>(ctx, nb, is_max, is_bin, tgtf, leq, leq_rows, cols, out_status, out_v, out_sol, out_nodes,
                                           (const uint8_t *)0, eqs, eq_rows);
        if (rc != XPG_ERR_UNSUPPORTED) return rc;
    }
    const int rhs = cols - 1;
    std::vector<S> vc((size_t)rhs * cols, zero<S>());
    for (int i = 0; i < rhs; i++) vc[(size_t)i * cols + i] = minus_one<S>();
    std::vector<MipTask<S> > tasks(nb);
    for (int b = 0; b < nb; b++)
        tasks[b].start(make_problem<S>(tgtf + (size_t)b * cols, vc.data(), rhs, eqs + (size_t)b * eq_rows * cols, eq_rows,
                                       leq_rows > 0 ? leq + (size_t)b * leq_rows * cols : (const S *)0, leq_rows, cols),
                       is_max, is_bin, (const uint8_t *)0);
    int rc = run_mip_tasks<S>(ctx, kind, tasks);
    if (rc) return rc;
    long long nodes = 0;
    for (int b = 0; b < nb; b++) {
        const MipTask<S> & T = tasks[b];
        out_status[b] = T.final_status;
        out_v[b] = T.v;
        nodes += T.nodes;
        if (T.final_status == XPG_IP_SUCC && out_sol && (int)T.sol.size() == cols)
            for (int j = 0; j < cols; j++) out_sol[(size_t)b * cols + j] = T.sol[j];
    }
    if (out_nodes) *out_nodes = nodes;
    return 0;
}

// SIX::reviseTargetFunc on the all-ones objective (lpsol.h:2053-2074, linsys.cpp:851-862).
inline std::vector<R32> feasibility_objective(const R32 * leq, int leq_rows, const R32 * eqs, int eq_rows, int cols, int rhs)
{
    std::vector<R32> tgtf(cols, R32(0, 1));
    for (int j = 0; j < rhs; j++) {
        bool nz = false;
        for (int i = 0; i < leq_rows && !nz; i++) nz = !eq(leq[(size_t)i * cols + j], R32(0, 1));
        for (int i = 0; i < eq_rows && !nz; i++) nz = !eq(eqs[(size_t)i * cols + j], R32(0, 1));
        tgtf[j] = nz ? R32(1, 1) : R32(0, 1);
    }
    return tgtf;
}

// Lineq::has_solution (linsys.cpp:830-906): maxm then minm; success, or an unbounded answer
// when a unique solution is not demanded, means "has a solution".
inline int has_solution(xpg_ctx * ctx, const R32 * leq, int leq_rows, const R32 * eqs, int eq_rows, const R32 * vc,
                        int vc_rows, int cols, int rhs, bool is_int, bool is_unique)
{
    if (!ctx || !vc || cols < 2 || rhs != cols - 1 || vc_rows != rhs) return XPG_ERR_SHAPE;
    if (leq_rows == 0 && eq_rows == 0) return 0;
    if (leq_rows == 0) return XPG_ERR_REF_UNDEFINED;      // the reference sizes tgtf from leq (linsys.cpp:851)
    const std::vector<R32> tgtf = feasibility_objective(leq, leq_rows, eqs, eq_rows, cols, rhs);
    R32 v; std::vector<R32> sol(cols);
    for (int pass = 0; pass < 2; pass++) {
        int st = is_int
            ? mip_solve<R32>(ctx, 1, pass == 0, false, tgtf.data(), vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols,
                             (const uint8_t *)0, &v, sol.data(), (long *)0)
            : six_solve<R32>(ctx, 1, pass == 0, tgtf.data(), vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols,
                             0xFFFFFFFFu, &v, sol.data());
        if (st < 0) return st;
        if (st == 0) return 1;
        if (!is_unique && st == 1) return 1;
    }
    return 0;
}


// DepPoly::is_empty(keepit, vc) (src/eng/poly.cpp:530-573) for nb dependence polyhedra of one shape: the constant
// is column rhs_idx, columns after it are constant symbols. move2var (when there are symbols) -> Lineq::reduce at
// the last column -> inconsistent: empty; no row left: not empty; else Lineq::has_solution(int, unique) with the
// caller's variable constraints vc [rhs_idx][rhs_idx + 1] (NULL: -x_i <= 0, poly.cpp:563-567).
// With symbols that last step is undefined in the reference: has_solution is handed rhs_idx = the number of
// variables while the matrix has grown by the symbols, which SIX::verify (lpsol.h:1526-1552, "No yet support const
// term with multi-columns") only ASSERTs in debug builds -- those systems get XPG_ERR_REF_UNDEFINED, the ones
// reduce decides get their answer.
// symbols_as_vars (opt-in, NOT parity: XPG_DEP_SYMBOLS_AS_VARS): the evident intent of poly.cpp:530-573 for a parametrised
// polyhedron -- after move2var the constant symbols ARE variables (free ones: nothing is known of their sign), so has_solution
// is asked about the widened system, rhs_idx = the last column, vc widened by all-zero rows / columns for the symbols.
inline int dep_is_empty_batch(xpg_ctx * ctx, int nb, const R32 * mats, int rows, int cols, int rhs_idx, const R32 * vc_in,
                              int32_t * out_empty, long * out_nodes, int symbols_as_vars = 0)
{
    if (!ctx || nb < 0 || !mats || rows <= 0 || cols < 2 || !out_empty || rhs_idx < 1 || rhs_idx > cols - 1) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const int last = cols - 1, nsym = last - rhs_idx;
    // No constant symbols and the default x >= 0: the whole test stays on the device -- reduce, the feasibility
    // objectives, the integer maxm walk, the minm walk of what that left open -- and only the verdicts come back.
    static const bool on_dev = [] { const char * e = xpg_env("XPG_MIP_DEVICE"); return !(e && e[0] == '0'); }();
    if (on_dev && nsym == 0 && !vc_in && mip_device_fits<R32>(rows, cols, false) &&
        lineq_lds_bytes(rows, cols) <= 160 * 1024 && rows <= 32767) {
        const int n = cols - 1, rmax = rows + n, depth = n + 2;
        const size_t bm = (size_t)nb * rows * cols * 8, bt = (size_t)nb * cols * 8;
        DevBuf dm, dt, dk, dok, dact, demp, dst, dv, dn, dws;
        XPG_TRY(dm.alloc(ctx, bm)); XPG_TRY(dt.alloc(ctx, bt)); XPG_TRY(dk.alloc(ctx, (size_t)nb * 4));
        XPG_TRY(dok.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dact.alloc(ctx, (size_t)nb * 4)); XPG_TRY(demp.alloc(ctx, (size_t)nb * 4));
        XPG_TRY(dst.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dv.alloc(ctx, (size_t)nb * 8)); XPG_TRY(dn.alloc(ctx, (size_t)nb * 4));
        XPG_TRY(hipMemcpyAsync(dm.p, mats, bm, hipMemcpyHostToDevice, ctx->stream));
        // Lineq::reduce on the device arrays (the C ABI entry: its kernels live in the row-elimination translation unit)
        int rc = xpg_lineq_reduce_batch_rat32_dev(ctx, nb, (xpg_rat32 *)dm.p, rows, cols, last, 1, (int32_t *)dk.p, (int32_t *)dok.p);
        if (rc) return rc;
        hipLaunchKernelGGL(k_dep_prepare, dim3((nb + 3) / 4), dim3(64, 4), 0, ctx->stream, nb, (const R32 *)dm.p, rows, cols,
                           (const int *)dk.p, (const int *)dok.p, (R32 *)dt.p, (int *)dact.p, (int32_t *)demp.p);
        XPG_TRY(hipMemsetAsync(dn.p, 0, (size_t)nb * 4, ctx->stream));
        std::vector<int32_t> nodes_a((size_t)nb, 0), nodes_b((size_t)nb, 0);
        for (int pass = 0; pass < 2; pass++) {
            const bool is_max = pass == 0;
            const MipGeom g = mip_geom<R32>(ctx, nb, rmax, n, is_max);
            const size_t lds = g.lds;
            const int threads = g.threads, grid = g.grid;
            const size_t ws_words = mip_ws_words(rmax, cols, depth);
            const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
            if (pass == 0) XPG_TRY(dws.alloc(ctx, (size_t)(cus * 32 < nb ? cus * 32 : nb) * ws_words * 8));   // the largest grid of either pass
            XPG_TRY(hipMemsetAsync(dn.p, 0, (size_t)nb * 4, ctx->stream));
            XPG_TRY(lds_limit((const void *)k_mip_tree<R32>, ctx->device, lds));
            hipLaunchKernelGGL((k_mip_tree<R32>), dim3(grid), dim3(threads), lds, ctx->stream, nb, (const R32 *)dt.p, (const R32 *)dm.p,
                               rows, cols, is_max ? 1 : 0, 0, rmax, depth, (unsigned long long *)dws.p, ws_words,
                               (int32_t *)dst.p, (R32 *)dv.p, (R32 *)0, (int *)dn.p, (const int *)dk.p, (const int *)dact.p, (const uint8_t *)0,
                               (const R32 *)0, 0, grid, (int *)0);
            XPG_TRY(hipGetLastError());
            XPG_TRY(hipMemcpyAsync(pass == 0 ? nodes_a.data() : nodes_b.data(), dn.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
            hipLaunchKernelGGL(k_dep_update, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, nb, (const int32_t *)dst.p,
                               (int *)dact.p, (int32_t *)demp.p);
        }
        XPG_TRY(hipMemcpyAsync(out_empty, demp.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
        if (out_nodes) { long t = 0; for (int b = 0; b < nb; b++) t += nodes_a[(size_t)b] + nodes_b[(size_t)b]; *out_nodes = t; }
        return 0;
    }
    std::vector<R32> work((size_t)nb * rows * cols);
    if (nsym > 0) {
        for (int b = 0; b < nb; b++)
            move2var_one(mats + (size_t)b * rows * cols, work.data() + (size_t)b * rows * cols, rows, cols, rhs_idx, rhs_idx + 1, last);
    } else {
        work.assign(mats, mats + (size_t)nb * rows * cols);
    }
    std::vector<int32_t> kept(nb), ok(nb);
    int rc = xpg_lineq_reduce_batch_rat32(ctx, nb, (xpg_rat32 *)work.data(), rows, cols, last, 1, kept.data(), ok.data());
    if (rc) return rc;
    const bool widen = nsym > 0 && symbols_as_vars != 0;
    const int nv = widen ? last : rhs_idx;
    std::vector<R32> vc((size_t)nv * (nv + 1), R32(0, 1));
    if (vc_in && !widen) vc.assign(vc_in, vc_in + (size_t)nv * (nv + 1));
    else if (vc_in) {                                // the caller's [rhs_idx][rhs_idx + 1] block; the symbols' rows and columns stay zero (free)
        for (int i = 0; i < rhs_idx; i++) {
            for (int j = 0; j < rhs_idx; j++) vc[(size_t)i * (nv + 1) + j] = vc_in[(size_t)i * (rhs_idx + 1) + j];
            vc[(size_t)i * (nv + 1) + nv] = vc_in[(size_t)i * (rhs_idx + 1) + rhs_idx];
        }
    }
    else for (int i = 0; i < rhs_idx; i++) vc[(size_t)i * (nv + 1) + i] = R32(-1, 1);
    std::vector<int> open;                       // systems still undecided
    for (int b = 0; b < nb; b++) {
        if (!ok[b]) out_empty[b] = 1;            // inconsistent bounds: empty (poly.cpp:550-552)
        else if (kept[b] == 0) out_empty[b] = 0; // only redundant constraints: conservatively non-empty (:553-557)
        else if (nsym > 0 && !widen) out_empty[b] = XPG_ERR_REF_UNDEFINED;
        else { out_empty[b] = 1; open.push_back(b); }
    }
    long nodes = 0;
    for (int pass = 0; pass < 2 && !open.empty(); pass++) {          // maxm, then minm (linsys.cpp:864-876)
        std::vector<MipTask<R32> > tasks(open.size());
        for (size_t t = 0; t < open.size(); t++) {
            const int b = open[t];
            const R32 * leq = work.data() + (size_t)b * rows * cols;
            const std::vector<R32> tgtf = feasibility_objective(leq, kept[b], (const R32 *)0, 0, cols, last);
            tasks[t].start(make_problem<R32>(tgtf.data(), vc.data(), nv, (const R32 *)0, 0, leq, kept[b], cols),
                           pass == 0, false, (const uint8_t *)0);
        }
        rc = run_mip_tasks<R32>(ctx, 1, tasks);
        if (rc) return rc;
        std::vector<int> still;
        for (size_t t = 0; t < open.size(); t++) {
            nodes += tasks[t].nodes;
            const int st = tasks[t].final_status;
            if (st < 0) out_empty[open[t]] = st;             // reference undefined on this system
            else if (st == XPG_IP_SUCC) out_empty[open[t]] = 0;
            else still.push_back(open[t]);
        }
        open.swap(still);
    }
    if (out_nodes) *out_nodes = nodes;
    return 0;
}

} // namespace xpg
