// C++ adapter: classes with the names, signatures and status codes of xpoly's
// SIX<Mat,T> (src/com/lpsol.h:204-338) on top of the C ABI in ../xpoly_amd.h, so
// that an xpoly call site such as
//
//     SIX<RMat, Rational> six;                       // src/com/linsys.cpp:884
//     six.set_param(0, 10000);
//     UINT st = six.maxm(v, sol, tgtf, vc, eq, leq);  // src/com/lpsol.h:274-280
//
// compiles unchanged against `xpoly_amd::SIX` and runs on the GPU.  Header only;
// it relies on nothing but the duck-typed Matrix<T> contract the reference's own
// SIX relies on: get_row_size(), get_col_size(), get_matrix() (a row-major T*,
// src/com/matt.h:152-156, :281-283), size(), reinit(rows, cols), and on T being
// either an 8-byte fp64 wrapper (Float, flty.h:47-62) or an {int32 num; int32 den}
// pair (Rational, rational.h:66-67).  Include it AFTER the xpoly headers, or with
// any Matrix type that offers the same members.
#ifndef XPOLY_AMD_SIX_HPP
#define XPOLY_AMD_SIX_HPP

#include <cstddef>
#include <cstring>
#include <mutex>
#include <vector>
#include "../xpoly_amd.h"

namespace xpoly_amd {

// Which ABI flavour a scalar maps to. Specialise for your scalar type; the
// primary template guesses from the size-8 layouts the reference uses.
template <class T> struct scalar_kind { static const int value = -1; };

namespace detail {

// The context every default-constructed solver of this process shares (device 0). Created once, by whichever
// host thread constructs its first solver; later calls only read the pointer. The C ABI allows one thread
// per handle at a time: callers that solve from several threads give each solver its own context
// (`SIX(ctx)`, `MIP(ctx)`, `Lineq(m, rhs, ctx)`).
inline xpg_ctx * shared_context()
{
    static xpg_ctx * ctx = 0;
    static std::once_flag once;
    std::call_once(once, [] { if (xpg_create(&ctx, 0) != 0) ctx = 0; });
    return ctx;
}

// SIX::reviseTargetFunc / MIP::reviseTargetFunc, lpsol.h:2053-2074: a variable whose column is all zero in
// both systems gets objective coefficient 0 (it would make the problem unbounded for no reason). Host side in the
// reference too -- O(rows x cols) reads of the caller's matrices, no arithmetic -- so it stays on the host here.
template <class Mat> inline void revise_target(Mat & tgtf, Mat const & eq, Mat const & leq, int rhs_idx)
{
    for (int j = 0; j < rhs_idx; j++) {
        bool nonzero = false;
        if (leq.get_col_size() > 0 && !leq.is_colequ((unsigned)j, 0)) nonzero = true;
        if (eq.get_col_size() > 0 && !eq.is_colequ((unsigned)j, 0)) nonzero = true;
        if (!nonzero) tgtf.set(0, (unsigned)j, 0);
    }
}

template <class Mat> inline const void * data_of(Mat const & m)
{ return m.size() ? (const void *)m.get_matrix() : (const void *)0; }

} // namespace detail

template <class Mat, class T> class SIX {
    unsigned m_max_iter;
    unsigned m_indent;
    xpg_ctx * m_ctx;
    static_assert(sizeof(T) == 8, "xpoly_amd::SIX: T must be Float (fp64) or Rational (int32/int32)");
public:
    explicit SIX(xpg_ctx * ctx = 0) : m_max_iter(0xFFFFFFFFu), m_indent(0), m_ctx(ctx) {}
    void init() {}
    void destroy() {}
    // SIX::set_param, lpsol.h:380-385
    void set_param(unsigned indent, unsigned max_iter = 0xFFFFFFFFu) { m_indent = indent; m_max_iter = max_iter; }
    // SIX::reviseTargetFunc, lpsol.h:329-333 / :2053-2074 (call sites linsys.cpp:885, lpsol.h MIP)
    void reviseTargetFunc(Mat & tgtf, Mat const & eq, Mat const & leq, int rhs_idx) { detail::revise_target(tgtf, eq, leq, rhs_idx); }

    // SIX::maxm, lpsol.h:1993-2033 -- same argument order and meaning.
    unsigned maxm(T & maxv, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq, int rhs_idx = -1)
    { return solve(true, maxv, res, tgtf, vc, eq, leq, rhs_idx); }
    // SIX::minm, lpsol.h:1662-1732
    unsigned minm(T & minv, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq, int rhs_idx = -1)
    { return solve(false, minv, res, tgtf, vc, eq, leq, rhs_idx); }

    // SIX::TwoStageMethod, lpsol.h:291-301 / :1907-1930 -- same in/out arguments: newleq comes in as the
    // inequalities (A | b) of an x >= 0 problem and goes out as the slack tableau after at most max_iter pivots,
    // newtgtf as its objective row, newvc grown to the slack variables, the basis in the four vectors (anything
    // with set(i, v): Vector<bool> / Vector<INT>, sstl.h), new_rhs_idx the constant column of the tableau.
    // The tableau stays in HBM for the whole solve (xpg_lp_*); it is downloaded once at the end.
    template <class VecB, class VecI>
    unsigned TwoStageMethod(Mat & newleq, Mat & newvc, Mat & newtgtf, Mat & slack_sol, T & maxv, VecB & nvset, VecB & bvset,
                            VecI & bv2eqmap, VecI & eq2bvmap, int & new_rhs_idx)
    {
        const int kind = scalar_kind<T>::value;
        xpg_ctx * ctx = m_ctx ? m_ctx : detail::shared_context();
        if (!ctx) return (unsigned)XPG_ERR_NO_DEVICE;
        const int m = (int)newleq.get_row_size(), cols = (int)newleq.get_col_size(), n0 = cols - 1;
        if (kind < 0 || new_rhs_idx != n0 || (int)newvc.get_row_size() != n0 || (int)newvc.get_col_size() != cols ||
            (int)newtgtf.get_col_size() != cols)
            return (unsigned)XPG_ERR_SHAPE;
        std::vector<T> vcd((size_t)n0), vcr((size_t)n0);          // the only cells of vc the solver reads (lpsol.h:798-802)
        for (int i = 0; i < n0; i++) { vcd[(size_t)i] = newvc.get(i, i); vcr[(size_t)i] = newvc.get(i, n0); }
        xpg_lp * lp = 0;
        int st = xpg_lp_create(ctx, kind, detail::data_of(newleq), m, cols, detail::data_of(newtgtf), vcd.data(), vcr.data(), 0, &lp);
        if (st != 0) return (unsigned)st;
        st = xpg_lp_two_stage(lp, m_max_iter);
        int rows = 0, W = 0, rhs = 0;
        if (st >= 0 && xpg_lp_shape(lp, &rows, &W, &rhs) == 0 && st != XPG_SIX_NO_PRI_FEASIBLE_SOL) {
            std::vector<unsigned char> nv((size_t)rhs), bv((size_t)rhs);
            std::vector<int32_t> b2e((size_t)rhs), e2b((size_t)rows);
            newleq.reinit(rows, W); newtgtf.reinit(1, W); slack_sol.reinit(1, W);
            T mv; std::memset((void *)&mv, 0, sizeof(T));
            const int rc = xpg_lp_read(lp, (void *)newleq.get_matrix(), (void *)newtgtf.get_matrix(), nv.data(), bv.data(),
                                       b2e.data(), e2b.data(), (void *)&mv, (void *)slack_sol.get_matrix());
            if (rc != 0) st = rc;
            for (int i = 0; i < rhs; i++) { nvset.set(i, nv[(size_t)i] != 0); bvset.set(i, bv[(size_t)i] != 0); bv2eqmap.set(i, b2e[(size_t)i]); }
            for (int i = 0; i < rows; i++) eq2bvmap.set(i, e2b[(size_t)i]);
            // vc as SIX::slack leaves it (lpsol.h:1422-1431): the caller's rows widened, -x_s <= 0 for every slack
            Mat grown; grown.reinit(rhs, W);
            T zero_; std::memset((void *)&zero_, 0, sizeof(T));
            if (kind == 1) { int32_t z[2] = {0, 1}; std::memcpy((void *)&zero_, z, 8); }
            T neg1 = zero_;
            if (kind == 0) { const double d = -1.0; std::memcpy((void *)&neg1, &d, 8); } else { int32_t q[2] = {-1, 1}; std::memcpy((void *)&neg1, q, 8); }
            for (int i = 0; i < rhs; i++)
                for (int j = 0; j < W; j++) grown.set(i, j, zero_);
            for (int i = 0; i < rhs; i++) {
                grown.set(i, i, i < n0 ? vcd[(size_t)i] : neg1);
                if (i < n0) grown.set(i, rhs, vcr[(size_t)i]);
            }
            newvc = grown;
            maxv = mv;
            new_rhs_idx = rhs;
        }
        xpg_lp_destroy(lp);
        return (unsigned)st;
    }

private:
    unsigned solve(bool is_max, T & v, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq, int rhs_idx)
    {
        const int kind = scalar_kind<T>::value;
        xpg_ctx * ctx = m_ctx ? m_ctx : detail::shared_context();
        const int cols = (int)tgtf.get_col_size();
        // the reference only ASSERTs these (lpsol.h:1527-1557); we refuse instead
        if (!ctx) return (unsigned)XPG_ERR_NO_DEVICE;
        if (kind < 0 || (rhs_idx != -1 && rhs_idx != cols - 1)) return (unsigned)XPG_ERR_SHAPE;
        std::vector<T> out_sol((size_t)cols);
        T out_v;
        std::memset((void *)&out_v, 0, sizeof(T));
        int st;
        const int eq_rows = eq.size() ? (int)eq.get_row_size() : 0;
        const int leq_rows = leq.size() ? (int)leq.get_row_size() : 0;
        if (kind == 0) {
            st = (is_max ? xpg_six_maxm_f64 : xpg_six_minm_f64)(
                ctx, (const double *)detail::data_of(tgtf), (const double *)detail::data_of(vc),
                (int)vc.get_row_size(), (const double *)detail::data_of(eq), eq_rows,
                (const double *)detail::data_of(leq), leq_rows, cols, m_max_iter,
                (double *)&out_v, (double *)out_sol.data());
        } else {
            st = (is_max ? xpg_six_maxm_rat32 : xpg_six_minm_rat32)(
                ctx, (const xpg_rat32 *)detail::data_of(tgtf), (const xpg_rat32 *)detail::data_of(vc),
                (int)vc.get_row_size(), (const xpg_rat32 *)detail::data_of(eq), eq_rows,
                (const xpg_rat32 *)detail::data_of(leq), leq_rows, cols, m_max_iter,
                (xpg_rat32 *)&out_v, (xpg_rat32 *)out_sol.data());
        }
        v = out_v;                                   // 0 on non-success, lpsol.h:2024
        if (st == XPG_SIX_SUCC) {
            res.reinit(1, cols);                     // calcFinalSolution, lpsol.h:1880
            std::memcpy((void *)res.get_matrix(), (const void *)out_sol.data(), sizeof(T) * (size_t)cols);
        }
        return (unsigned)st;
    }
};

// MIP<Mat,T> (src/com/lpsol.h:2087-2157): maxm / minm with is_bin and a BMat-like
// rational_indicator (anything with get_col_size() and get(0, j) -> bool), same IP_* codes.
template <class Mat, class T> class MIP {
    xpg_ctx * m_ctx;
    static_assert(sizeof(T) == 8, "xpoly_amd::MIP: T must be Float (fp64) or Rational (int32/int32)");
public:
    explicit MIP(xpg_ctx * ctx = 0) : m_ctx(ctx) {}
    void init() {}
    void destroy() {}

    // MIP::maxm / minm, lpsol.h:2121-2140: `bool is_bin = false, BMat * rational_indicator = NULL, INT rhs_idx = -1`.
    // The indicator's type is the caller's (anything with get(0, j) -> bool), so it is a template parameter; the
    // reference's own call sites pass a literal NULL (linsys.cpp:864-866, :873-875), from which no pointer type can be
    // deduced -- those bind to the std::nullptr_t overloads (NULL and nullptr both convert), which also carry the
    // reference's defaults (poly.cpp:5131-5134 passes six arguments).
    template <class BoolMat>
    unsigned maxm(T & maxv, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq,
                  bool is_bin, BoolMat * rational_indicator, int rhs_idx = -1)
    { return solve(true, maxv, res, tgtf, vc, eq, leq, is_bin, rational_indicator, rhs_idx); }
    template <class BoolMat>
    unsigned minm(T & minv, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq,
                  bool is_bin, BoolMat * rational_indicator, int rhs_idx = -1)
    { return solve(false, minv, res, tgtf, vc, eq, leq, is_bin, rational_indicator, rhs_idx); }
    unsigned maxm(T & maxv, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq, bool is_bin = false,
                  std::nullptr_t = nullptr, int rhs_idx = -1)
    { return solve(true, maxv, res, tgtf, vc, eq, leq, is_bin, (no_indicator *)0, rhs_idx); }
    unsigned minm(T & minv, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq, bool is_bin = false,
                  std::nullptr_t = nullptr, int rhs_idx = -1)
    { return solve(false, minv, res, tgtf, vc, eq, leq, is_bin, (no_indicator *)0, rhs_idx); }
    // MIP::reviseTargetFunc, lpsol.h:2141-2145 (it forwards to SIX's; call sites linsys.cpp:861, poly.cpp:5127)
    void reviseTargetFunc(Mat & tgtf, Mat const & eq, Mat const & leq, int rhs_idx) { detail::revise_target(tgtf, eq, leq, rhs_idx); }

private:
    struct no_indicator { bool get(int, int) const { return false; } };
    template <class BoolMat>
    unsigned solve(bool is_max, T & v, Mat & res, Mat const & tgtf, Mat & vc, Mat const & eq, Mat const & leq,
                   bool is_bin, BoolMat * ind, int rhs_idx)
    {
        const int kind = scalar_kind<T>::value;
        xpg_ctx * ctx = m_ctx ? m_ctx : detail::shared_context();
        const int cols = (int)tgtf.get_col_size();
        if (!ctx) return (unsigned)XPG_ERR_NO_DEVICE;
        if (kind < 0 || (rhs_idx != -1 && rhs_idx != cols - 1)) return (unsigned)XPG_ERR_SHAPE;
        std::vector<unsigned char> flags;
        if (ind) { flags.resize((size_t)cols); for (int j = 0; j < cols; j++) flags[(size_t)j] = ind->get(0, j) ? 1 : 0; }
        std::vector<T> out_sol((size_t)cols);
        T out_v;
        std::memset((void *)&out_v, 0, sizeof(T));
        const int eq_rows = eq.size() ? (int)eq.get_row_size() : 0;
        const int leq_rows = leq.size() ? (int)leq.get_row_size() : 0;
        int st;
        if (kind == 0)
            st = (is_max ? xpg_mip_maxm_f64 : xpg_mip_minm_f64)(
                ctx, (const double *)detail::data_of(tgtf), (const double *)detail::data_of(vc), (int)vc.get_row_size(),
                (const double *)detail::data_of(eq), eq_rows, (const double *)detail::data_of(leq), leq_rows, cols,
                is_bin ? 1 : 0, ind ? flags.data() : (const unsigned char *)0, (double *)&out_v, (double *)out_sol.data());
        else
            st = (is_max ? xpg_mip_maxm_rat32 : xpg_mip_minm_rat32)(
                ctx, (const xpg_rat32 *)detail::data_of(tgtf), (const xpg_rat32 *)detail::data_of(vc), (int)vc.get_row_size(),
                (const xpg_rat32 *)detail::data_of(eq), eq_rows, (const xpg_rat32 *)detail::data_of(leq), leq_rows, cols,
                is_bin ? 1 : 0, ind ? flags.data() : (const unsigned char *)0, (xpg_rat32 *)&out_v, (xpg_rat32 *)out_sol.data());
        v = out_v;
        if (st == XPG_IP_SUCC) {
            res.reinit(1, cols);
            std::memcpy((void *)res.get_matrix(), (const void *)out_sol.data(), sizeof(T) * (size_t)cols);
        }
        return (unsigned)st;
    }
};

} // namespace xpoly_amd
#endif
