// C++ adapter: a class with the names and signatures of xpoly's Lineq (src/com/linsys.h:61-186) for the
// members that sit on the hot path -- reduce, fme, has_solution, calcBound, move2var, removeIdenRow -- on top of
// the C ABI in ../xpoly_amd.h, plus the small host-side members the engine's own Lineq objects call next to them
// (appendEquation, formatBound, initVarConstraint, is_consistent), so that call sites such as
//
//     Lineq lin(NULL);                                      // src/eng/poly.cpp:537
//     if (!lin.reduce(*coeff, coeff->get_col_size() - 1, true)) return true;
//     return !lin.has_solution(*coeff, eq, lvc, rhs_idx, true, true);   // src/eng/poly.cpp:571, linsys.cpp:830
//
// compile unchanged against `xpoly_amd::Lineq<RMat>` and run on the GPU. Each member is the batch entry point
// with a batch of one (the batch forms in xpoly_amd.h are what a caller with a SCoP's worth of polyhedra
// should use). Header only; relies on the Matrix<Rational> contract only: get_row_size(), get_col_size(),
// get_matrix() (row-major {int32 num; int32 den}, matt.h:152-156, rational.h:66-67), size(), reinit(r, c).
#ifndef XPOLY_AMD_LINEQ_HPP
#define XPOLY_AMD_LINEQ_HPP

#include <cstddef>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>
#include "../xpoly_amd.h"
#include "six.hpp"

namespace xpoly_amd {

template <class RMatT> class Lineq {
    xpg_ctx * m_ctx;
    RMatT * m_coeff;                 // linsys.h:64-70
    int m_rhs_idx;
    xpg_ctx * ctx() { return m_ctx ? m_ctx : detail::shared_context(); }
    static void put(RMatT & m, const std::vector<xpg_rat32> & a, int rows, int cols)
    {
        m.reinit(rows, cols);
        if (rows * cols) std::memcpy((void *)m.get_matrix(), (const void *)a.data(), sizeof(xpg_rat32) * (size_t)rows * cols);
    }
public:
    explicit Lineq(RMatT * m, int rhs_idx = -1, xpg_ctx * c = 0) : m_ctx(c), m_coeff(m), m_rhs_idx(rhs_idx)
    { if (m && rhs_idx == -1) m_rhs_idx = (int)m->get_col_size() - 1; }
    void init(RMatT * m, int rhs_idx = -1) { set_param(m, rhs_idx); }   // linsys.cpp:92-108
    void destroy() { m_coeff = 0; m_rhs_idx = -1; }
    void set_param(RMatT * m, int rhs_idx = -1)             // linsys.cpp:117-131
    { m_coeff = m; m_rhs_idx = (m && rhs_idx == -1) ? (int)m->get_col_size() - 1 : rhs_idx; }

    // ---- host-side members (no arithmetic worth a launch; they only reshape the caller's matrices with the
    //      Matrix type's own operations, exactly the ones the reference uses, so the cells are the reference's)
    // Lineq::appendEquation, linsys.cpp:922-935 (caller PolyTran::scan's bound computation, src/eng/poly.cpp:4804):
    // every equation a = b joins the system as a <= b and -a <= -b.
    void appendEquation(RMatT const & eq)
    {
        if (eq.size() == 0 || eq.get_row_size() == 0) return;
        RMatT both = eq;
        m_coeff->grow_row(both, 0, both.get_row_size() - 1);
        both.mul(-1);
        m_coeff->grow_row(both, 0, both.get_row_size() - 1);
    }
    // Lineq::initVarConstraint, linsys.cpp:803-819 (caller src/eng/poly.cpp:1603): -x_i <= 0 for every variable
    // whose sign entry is >= 0 (all of them without a sign vector), nothing for the others.
    template <class SignVec> void initVarConstraint(SignVec const * sign, RMatT & vc, unsigned rhs_idx)
    {
        typedef typename std::decay<decltype(std::declval<RMatT const &>().get(0u, 0u))>::type R;
        vc.reinit(rhs_idx, rhs_idx + 1);
        vc.set_col(rhs_idx, R(0));
        for (unsigned i = 0; i < rhs_idx; i++)
            if (!sign || sign->get(i) >= 0) vc.set(i, i, -1);
    }
    void initVarConstraint(std::nullptr_t, RMatT & vc, unsigned rhs_idx) { initVarConstraint((all_nonnegative const *)0, vc, rhs_idx); }
private:
    struct all_nonnegative { int get(unsigned) const { return 0; } };
public:
    // Lineq::is_consistent, linsys.cpp:779-800: eliminate the variables one after the other (each step an fme on the
    // device); the system is contradictory as soon as one elimination says so. The caller's matrix is left alone.
    bool is_consistent()
    {
        RMatT * const keep = m_coeff;
        const int keep_rhs = m_rhs_idx;
        RMatT work = *keep;
        set_param(&work, keep_rhs);
        bool consistent = true;
        for (unsigned v = 0; consistent && v < (unsigned)keep_rhs; v++) {
            RMatT next;
            if (!fme(v, next)) consistent = false; else work = next;
        }
        set_param(keep, keep_rhs);
        return consistent;
    }
    // Lineq::formatBound, linsys.cpp:948-1030 (caller src/eng/ldtran.cpp:1534): the rows that mention variable u, its
    // coefficient brought to +-1, the other variables moved to the right-hand side (negated, appended behind the
    // constant / symbol columns), every row then over one common denominator from column 1 on.
    void formatBound(unsigned u, RMatT & bound_of_u)
    {
        typedef typename std::decay<decltype(std::declval<RMatT const &>().get(0u, 0u))>::type R;
        bound_of_u.reinit(0, 0);
        for (unsigned i = 0; i < m_coeff->get_row_size(); i++)
            if (m_coeff->get(i, u) != 0) {
                RMatT row;
                m_coeff->innerRow(row, i, i);
                bound_of_u.grow_row(row);
            }
        const unsigned rows = bound_of_u.get_row_size();
        if (rows == 0) return;                               // nothing constrains u
        const unsigned nvar = (unsigned)m_rhs_idx;
        const unsigned moved_at = bound_of_u.get_col_size();
        if (nvar != 1) bound_of_u.grow_col(nvar - 1);
        for (unsigned i = 0; i < rows; i++) {
            for (unsigned j = moved_at; j < bound_of_u.get_col_size(); j++) bound_of_u.setr(i, j, 0, 1);
            R c = bound_of_u.get(i, u);
            if (c < 0) c = -c;
            if (c != 1) bound_of_u.mulOfRow(i, 1 / c);
            if (nvar == 1) continue;
            unsigned k = moved_at;
            for (unsigned j = 0; j < nvar; j++)
                if (j != u) bound_of_u.set(i, k++, -bound_of_u.get(i, j));
        }
        if (nvar != 1) {
            if (u + 1 != nvar) bound_of_u.del_col(u + 1, nvar - 1);
            if (u > 0) bound_of_u.del_col(0, u - 1);
        }
        for (unsigned i = 0; i < rows; i++) bound_of_u.comden(i, 1);
    }

    // Lineq::reduce, linsys.cpp:359-626: in place; false = the system is inconsistent.
    bool reduce(RMatT & m, unsigned rhs_idx, bool is_intersect)
    {
        const int rows = (int)m.get_row_size(), cols = (int)m.get_col_size();
        if (rows == 0) return true;
        std::vector<xpg_rat32> a((size_t)rows * cols);
        std::memcpy((void *)a.data(), (const void *)m.get_matrix(), sizeof(xpg_rat32) * a.size());
        int32_t kept = 0, ok = 0;
        if (xpg_lineq_reduce_batch_rat32(ctx(), 1, a.data(), rows, cols, (int)rhs_idx, is_intersect ? 1 : 0, &kept, &ok) != 0) return false;
        put(m, a, kept, cols);
        return ok != 0;
    }
    // Lineq::removeIdenRow, linsys.cpp:1209-1268
    void removeIdenRow(RMatT & m)
    {
        const int rows = (int)m.get_row_size(), cols = (int)m.get_col_size();
        if (rows == 0) return;
        std::vector<xpg_rat32> a((size_t)rows * cols);
        std::memcpy((void *)a.data(), (const void *)m.get_matrix(), sizeof(xpg_rat32) * a.size());
        int32_t kept = 0;
        if (xpg_lineq_remove_iden_batch_rat32(ctx(), 1, a.data(), rows, cols, &kept) == 0) put(m, a, kept, cols);
    }
    // Lineq::move2var, linsys.cpp:1177-1200
    void move2var(RMatT & ieq, unsigned rhs_idx, unsigned first_sym, unsigned last_sym, unsigned * first_var_idx, unsigned * last_var_idx)
    {
        xpg_lineq_move2var_batch_rat32(ctx(), 1, (xpg_rat32 *)ieq.get_matrix(), (int)ieq.get_row_size(), (int)ieq.get_col_size(),
                                       (int)rhs_idx, (int)first_sym, (int)last_sym);
        if (first_var_idx) *first_var_idx = rhs_idx;
        if (last_var_idx) *last_var_idx = rhs_idx + (last_sym - first_sym);
    }
    // Lineq::fme on the system given to the constructor / set_param, linsys.cpp:656-774
    bool fme(unsigned u, RMatT & res, bool darkshadow = false)
    {
        // the packed form: only the live rows travel, and they are read straight out of the handle's pinned buffer
        if (m_coeff->size() == 0) { res.clean(); return true; }       // an empty system stays empty (linsys.cpp:661-664)
        const int rows = (int)m_coeff->get_row_size(), cols = (int)m_coeff->get_col_size();
        const xpg_rat32 * view = 0;
        long long off[2] = {0, 0};
        int32_t ok = 0;
        if (xpg_lineq_fme_batch_packed_rat32(ctx(), 1, (const xpg_rat32 *)m_coeff->get_matrix(), rows, cols, m_rhs_idx, (int)u,
                                             darkshadow ? 1 : 0, 0, (xpg_rat32 *)0, 0, &view, off, &ok) != 0) return false;
        const int orows = (int)off[1];
        res.reinit(orows, cols);
        if (orows * cols) std::memcpy((void *)res.get_matrix(), (const void *)view, sizeof(xpg_rat32) * (size_t)orows * cols);
        return ok != 0;
    }
    // Lineq::has_solution, linsys.cpp:830-906
    bool has_solution(RMatT const & leq, RMatT const & eq, RMatT & vc, unsigned rhs_idx, bool is_int_sol, bool is_unique_sol)
    {
        if (leq.size() == 0 && eq.size() == 0) return false;
        const int cols = leq.size() ? (int)leq.get_col_size() : (int)eq.get_col_size();
        const int r = xpg_has_solution_rat32(ctx(), (const xpg_rat32 *)detail::data_of(leq), leq.size() ? (int)leq.get_row_size() : 0,
                                             (const xpg_rat32 *)detail::data_of(eq), eq.size() ? (int)eq.get_row_size() : 0,
                                             (const xpg_rat32 *)detail::data_of(vc), (int)vc.get_row_size(), cols, (int)rhs_idx,
                                             is_int_sol ? 1 : 0, is_unique_sol ? 1 : 0);
        return r == 1;
    }
    // Lineq::calcBound, linsys.cpp:1047-1078: the bounds of variable j alone (every other variable eliminated, inner to
    // outer) for every j, all chains on the device in one call.
    //   bool calcBound(List<RMat*> & limits)   -- the reference's own signature (linsys.h:151): `limits` holds m_rhs_idx
    //                                             matrices the caller owns, *limits.get_head_nth(j) receives variable j's
    //                                             (linsys.cpp:1072-1074), so the call site at linsys.cpp:312 compiles unchanged
    //   bool calcBound(std::vector<RMat> &)    -- the same with value semantics (resized to m_rhs_idx)
private:
    template <class Sink> bool calc_bound_into(Sink sink)
    {
        // the packed form: only the live rows of the nv bound systems travel, read straight out of the handle's pinned buffer
        const int rows = (int)m_coeff->get_row_size(), cols = (int)m_coeff->get_col_size(), nv = m_rhs_idx;
        int cap = 4 * rows + 16;
        for (int attempt = 0; attempt < 2; attempt++) {
            std::vector<long long> off((size_t)nv + 1, 0);
            const xpg_rat32 * view = 0;
            int32_t ok = 0;
            if (xpg_lineq_calc_bound_batch_packed_rat32(ctx(), 1, (const xpg_rat32 *)m_coeff->get_matrix(), rows, cols, nv, cap, (xpg_rat32 *)0, 0,
                                                        &view, off.data(), &ok) != 0) return false;
            if (ok < 0) { cap = -ok; continue; }
            if (ok == 0) return false;                       // "system inconsistency!" (linsys.cpp:1065-1068)
            for (int j = 0; j < nv; j++) {
                const int r = (int)(off[(size_t)j + 1] - off[(size_t)j]);
                std::vector<xpg_rat32> one;
                if (r) one.assign(view + (size_t)off[(size_t)j] * cols, view + (size_t)off[(size_t)j + 1] * cols);
                sink(j, one, r, cols);
            }
            return true;
        }
        return false;
    }
public:
    bool calcBound(std::vector<RMatT> & limits)
    {
        limits.resize((size_t)m_rhs_idx);
        return calc_bound_into([&](int j, const std::vector<xpg_rat32> & a, int r, int c) { put(limits[(size_t)j], a, r, c); });
    }
    template <class ListT, class = decltype(std::declval<ListT &>().get_head_nth(0u)), class = decltype(std::declval<ListT &>().get_elem_count())>
    bool calcBound(ListT & limits)
    {
        if ((int)limits.get_elem_count() != m_rhs_idx) return false;      // the reference ASSERTs (linsys.cpp:1050-1051)
        return calc_bound_into([&](int j, const std::vector<xpg_rat32> & a, int r, int c) { put(*limits.get_head_nth((unsigned)j), a, r, c); });
    }
};

// One SCoP's worth of dependence tests in ONE call: DepPoly::is_empty (src/eng/poly.cpp:530-573, called per polyhedron by
// DepGraph::rebuild, poly.cpp:268-314) on polyhedra of whatever shapes DepPolyMgr::build produced (poly.cpp:1120-1224).
// empty[k] = 1 / 0 as is_empty would return for *polys[k], XPG_ERR_REF_UNDEFINED where the reference is undefined.
template <class RMatT>
inline int dep_is_empty_all(const std::vector<RMatT *> & polys, std::vector<int32_t> & empty, xpg_ctx * c = 0)
{
    const int nb = (int)polys.size();
    std::vector<int32_t> rows((size_t)nb), cols((size_t)nb);
    std::vector<long long> off((size_t)nb + 1, 0);
    for (int b = 0; b < nb; b++) {
        rows[(size_t)b] = (int32_t)polys[(size_t)b]->get_row_size(); cols[(size_t)b] = (int32_t)polys[(size_t)b]->get_col_size();
        off[(size_t)b + 1] = off[(size_t)b] + (long long)rows[(size_t)b] * cols[(size_t)b];
    }
    std::vector<xpg_rat32> flat((size_t)off[(size_t)nb]);
    for (int b = 0; b < nb; b++)
        if (rows[(size_t)b] * cols[(size_t)b])
            std::memcpy((void *)(flat.data() + off[(size_t)b]), (const void *)polys[(size_t)b]->get_matrix(),
                        sizeof(xpg_rat32) * (size_t)rows[(size_t)b] * cols[(size_t)b]);
    empty.assign((size_t)nb, 0);
    return xpg_dep_is_empty_batch_ragged_rat32(c ? c : detail::shared_context(), nb, flat.data(), rows.data(), cols.data(), off.data(),
                                               empty.data(), (long long *)0);
}

// Many eliminations in ONE call: Lineq::fme(u[k], results[k]) (linsys.cpp:656-774) on systems of whatever shapes, the constant
// in column rhs_idx[k] (empty vector: the last column of each). The reference's hot callers eliminate level by level --
// loop-bound generation (src/eng/ldtran.cpp:178-193: for i = rhs_idx - 1 .. 1: fme(i)) and scanning (src/eng/poly.cpp:4803-4821) --
// one small system per call; collected over the statements of a SCoP a level becomes one launch. ok[k] is fme's bool.
// Returns 0 or a negative XPG_ERR_* code.
template <class RMatT>
inline int fme_all(const std::vector<RMatT *> & systems, const std::vector<int32_t> & u, const std::vector<int32_t> & rhs_idx,
                   std::vector<RMatT> & results, std::vector<int32_t> & ok, bool darkshadow = false, xpg_ctx * c = 0)
{
    const int nb = (int)systems.size();
    if ((int)u.size() != nb || (!rhs_idx.empty() && (int)rhs_idx.size() != nb)) return XPG_ERR_SHAPE;
    results.resize((size_t)nb); ok.assign((size_t)nb, 0);
    // empty systems stay empty (linsys.cpp:661-664) and do not travel
    std::vector<int> live;
    for (int b = 0; b < nb; b++) { if (systems[(size_t)b]->size() == 0) { results[(size_t)b].clean(); ok[(size_t)b] = 1; } else live.push_back(b); }
    const int nl = (int)live.size();
    if (nl == 0) return 0;
    std::vector<int32_t> rows((size_t)nl), cols((size_t)nl), uu((size_t)nl), rr((size_t)nl), orows((size_t)nl), ook((size_t)nl);
    std::vector<long long> off((size_t)nl + 1, 0), ooff((size_t)nl + 1, 0);
    for (int k = 0; k < nl; k++) {
        const RMatT & m = *systems[(size_t)live[(size_t)k]];
        rows[(size_t)k] = (int32_t)m.get_row_size(); cols[(size_t)k] = (int32_t)m.get_col_size();
        uu[(size_t)k] = u[(size_t)live[(size_t)k]];
        rr[(size_t)k] = rhs_idx.empty() ? cols[(size_t)k] - 1 : rhs_idx[(size_t)live[(size_t)k]];
        off[(size_t)k + 1] = off[(size_t)k] + (long long)rows[(size_t)k] * cols[(size_t)k];
    }
    std::vector<xpg_rat32> flat((size_t)off[(size_t)nl]);
    for (int k = 0; k < nl; k++)
        std::memcpy((void *)(flat.data() + off[(size_t)k]), (const void *)systems[(size_t)live[(size_t)k]]->get_matrix(),
                    sizeof(xpg_rat32) * (size_t)rows[(size_t)k] * cols[(size_t)k]);
    xpg_ctx * h = c ? c : detail::shared_context();
    // a buffer for the worst case of every result while that is small (one call), else a sizing call first
    long long capc = 0;
    for (int k = 0; k < nl; k++) capc += ((long long)rows[(size_t)k] * rows[(size_t)k] / 4 + rows[(size_t)k]) * cols[(size_t)k];
    std::vector<xpg_rat32> outs;
    int rc;
    if (capc * (long long)sizeof(xpg_rat32) <= (64ll << 20)) {
        outs.resize((size_t)(capc > 0 ? capc : 1));
        rc = xpg_lineq_fme_batch_ragged_rat32(h, nl, flat.data(), rows.data(), cols.data(), off.data(), rr.data(), uu.data(), darkshadow ? 1 : 0,
                                              outs.data(), capc, ooff.data(), orows.data(), ook.data());
    } else {
        rc = xpg_lineq_fme_batch_ragged_rat32(h, nl, flat.data(), rows.data(), cols.data(), off.data(), rr.data(), uu.data(), darkshadow ? 1 : 0,
                                              (xpg_rat32 *)0, 0, ooff.data(), orows.data(), ook.data());
        if (rc == XPG_ERR_SHAPE && ooff[(size_t)nl] > 0) {
            outs.resize((size_t)ooff[(size_t)nl]);
            rc = xpg_lineq_fme_batch_ragged_rat32(h, nl, flat.data(), rows.data(), cols.data(), off.data(), rr.data(), uu.data(), darkshadow ? 1 : 0,
                                                  outs.data(), ooff[(size_t)nl], ooff.data(), orows.data(), ook.data());
        }
    }
    if (rc != 0) return rc;
    for (int k = 0; k < nl; k++) {
        RMatT & res = results[(size_t)live[(size_t)k]];
        res.reinit(orows[(size_t)k], cols[(size_t)k]);
        if (orows[(size_t)k] * cols[(size_t)k])
            std::memcpy((void *)res.get_matrix(), (const void *)(outs.data() + ooff[(size_t)k]), sizeof(xpg_rat32) * (size_t)orows[(size_t)k] * cols[(size_t)k]);
        ok[(size_t)live[(size_t)k]] = ook[(size_t)k];
    }
    return 0;
}

// Lineq::calcBound (linsys.cpp:1047-1078) for many systems: limits[k][j] receives the bounds of variable j of *systems[k]
// alone (the constant in column rhs_idx[k]); ok[k] is calcBound's bool. Systems of one shape go up together -- one call per
// shape class (SCoPs have a handful), every elimination chain of a class on the device at once. Returns 0 or XPG_ERR_*.
template <class RMatT>
inline int calc_bound_all(const std::vector<RMatT *> & systems, const std::vector<int32_t> & rhs_idx,
                          std::vector<std::vector<RMatT> > & limits, std::vector<int32_t> & ok, xpg_ctx * c = 0)
{
    const int nb = (int)systems.size();
    if ((int)rhs_idx.size() != nb) return XPG_ERR_SHAPE;
    limits.assign((size_t)nb, std::vector<RMatT>());
    ok.assign((size_t)nb, 0);
    xpg_ctx * h = c ? c : detail::shared_context();
    std::vector<char> done((size_t)nb, 0);
    for (int b0 = 0; b0 < nb; b0++) {
        if (done[(size_t)b0]) continue;
        const int rows = (int)systems[(size_t)b0]->get_row_size(), cols = (int)systems[(size_t)b0]->get_col_size(), nv = rhs_idx[(size_t)b0];
        std::vector<int> cls;
        for (int b = b0; b < nb; b++)
            if (!done[(size_t)b] && (int)systems[(size_t)b]->get_row_size() == rows && (int)systems[(size_t)b]->get_col_size() == cols && rhs_idx[(size_t)b] == nv) {
                cls.push_back(b); done[(size_t)b] = 1;
            }
        if (rows == 0 || cols == 0 || nv < 1 || nv >= cols) return XPG_ERR_SHAPE;
        const int nc = (int)cls.size();
        std::vector<xpg_rat32> flat((size_t)nc * rows * cols);
        for (int k = 0; k < nc; k++)
            std::memcpy((void *)(flat.data() + (size_t)k * rows * cols), (const void *)systems[(size_t)cls[(size_t)k]]->get_matrix(), sizeof(xpg_rat32) * (size_t)rows * cols);
        int cap = 4 * rows + 16;
        for (int attempt = 0; attempt < 3; attempt++) {
            std::vector<long long> off((size_t)nc * nv + 1, 0);
            std::vector<int32_t> kok((size_t)nc, 0);
            const xpg_rat32 * view = 0;
            const int rc = xpg_lineq_calc_bound_batch_packed_rat32(h, nc, flat.data(), rows, cols, nv, cap, (xpg_rat32 *)0, 0, &view, off.data(), kok.data());
            if (rc != 0) return rc;
            int need = 0;
            for (int k = 0; k < nc; k++) if (kok[(size_t)k] < 0 && -kok[(size_t)k] > need) need = -kok[(size_t)k];
            if (need) { cap = need; continue; }                     // a step needed more rows than cap: once more with that many
            for (int k = 0; k < nc; k++) {
                const int b = cls[(size_t)k];
                ok[(size_t)b] = kok[(size_t)k];
                limits[(size_t)b].resize((size_t)nv);
                for (int j = 0; j < nv; j++) {
                    const long long lo = off[(size_t)k * nv + j], hi = off[(size_t)k * nv + j + 1];
                    RMatT & m = limits[(size_t)b][(size_t)j];
                    m.reinit((unsigned)(hi - lo), (unsigned)cols);
                    if (hi > lo) std::memcpy((void *)m.get_matrix(), (const void *)(view + (size_t)lo * cols), sizeof(xpg_rat32) * (size_t)(hi - lo) * cols);
                }
            }
            break;
        }
    }
    return 0;
}

} // namespace xpoly_amd
#endif
