// TEST INFRASTRUCTURE -- CPU restatement ("port") of the rational row-elimination
// path of xpoly: Lineq::removeIdenRow / reduce / fme / has_solution
// (src/com/linsys.cpp) and Matrix<Rational>::rank / det / inv (src/com/matt.h).
// Never included by the product (xpoly_amd/).
#ifndef XPOLY_ORACLE_LINEQ_H
#define XPOLY_ORACLE_LINEQ_H

#include <vector>
#include "oracle_scalar.h"
#include "oracle_lp.h"

namespace orc {

typedef Mat<R32> RM;

enum { CST_UNK = 1, CST_LT = 2, CST_GT = 3, CST_EQ = 4 };                 // linsys.h:55-58

// Lineq::compareConstIterm against a value (linsys.cpp:204-231).
inline int cmp_const_value(const RM & m, int rhs, int row, R32 v)
{
    for (int j = rhs + 1; j < m.c; j++) if (ne(m.at(row, j), R32(0))) return CST_UNK;
    R32 c = m.at(row, rhs);
    if (eq(c, v)) return CST_EQ;
    return lt(c, v) ? CST_LT : CST_GT;
}

// Lineq::compareConstIterm between two rows (linsys.cpp:235-274).
inline int cmp_const_rows(const RM & m, int rhs, int r1, int r2)
{
    bool s1 = false, s2 = false, same = true;
    for (int j = rhs + 1; j < m.c; j++) {
        if (ne(m.at(r1, j), R32(0))) s1 = true;
        if (ne(m.at(r2, j), R32(0))) s2 = true;
        if (ne(m.at(r1, j), m.at(r2, j))) { same = false; break; }
    }
    if ((!s1 && !s2) || same) {
        R32 a = m.at(r1, rhs), b = m.at(r2, rhs);
        if (eq(a, b)) return CST_EQ;
        return lt(a, b) ? CST_LT : CST_GT;
    }
    return CST_UNK;
}

// Lineq::move2var (linsys.cpp:1177-1200): columns first_sym..last_sym are taken out, multiplied by -1 with the
// scalar's own '*' (Matrix::mul, matt.h:1331-1348: so a coefficient such as 2/4 comes back as -1/2) and put
// back in front of column rhs_idx, in their order.
inline void move2var(RM & m, int rhs_idx, int first_sym, int last_sym)
{
    const int rows = m.r, cols = m.c;
    RM out(rows, cols);
    for (int i = 0; i < rows; i++) {
        int c = 0;
        for (int j = 0; j < rhs_idx; j++) out.at(i, c++) = m.at(i, j);
        for (int j = first_sym; j <= last_sym; j++) out.at(i, c++) = mul(m.at(i, j), R32(-1, 1));
        for (int j = rhs_idx; j < cols; j++)
            if (j < first_sym || j > last_sym) out.at(i, c++) = m.at(i, j);
    }
    m = out;
}

// Lineq::removeIdenRow (linsys.cpp:1209-1268): drop later duplicates of a row;
// row sums are a prefilter, equality is field-wise.
inline void remove_iden_rows(RM & m)
{
    std::vector<R32> sum(m.r);
    std::vector<char> gone(m.r, 0);
    for (int i = 0; i < m.r; i++) {
        R32 s = R32(0);
        for (int j = 0; j < m.c; j++) s = add(s, m.at(i, j));
        sum[i] = s;
    }
    for (int i = 0; i < m.r; i++) {
        if (gone[i]) continue;
        for (int k = i + 1; k < m.r; k++) {
            if (ne(sum[i], sum[k])) continue;
            bool same = true;
            for (int j = 0; j < m.c && same; j++) same = eq(m.at(i, j), m.at(k, j));
            if (same) gone[k] = 1;
        }
    }
    if (m.r == 0) return;
    RM t(0, m.c);
    for (int i = 0; i < m.r; i++) {
        if (gone[i]) continue;
        t.a.insert(t.a.end(), m.row(i), m.row(i) + m.c);
        t.r++;
    }
    m = t;
}

inline void scale_row(RM & m, int row, R32 x) { scale_cells(m.row(row), m.c, 1, x); }

// The per-variable bound bookkeeping of Lineq::reduce (X2V_MAP, linsys.cpp:44-142).
struct Bounds { std::vector<std::vector<int> > pos, neg; };

// One side (all-positive or all-negative unit bounds) of Lineq::reduce's
// pairwise tightening (linsys.cpp:433-497 / :505-573). 'negative' selects the
// second block, which normalises by the negated coefficient.
inline void tighten_side(RM & m, int rhs, int var, const std::vector<int> & rows, bool negative,
                         bool is_intersect, std::vector<char> & removed, bool & any_removed)
{
    const int last = (int)rows.size() - 1;
    for (int k1 = 0; k1 < last; k1++) {
        const int r1 = rows[k1];
        if (removed[r1]) continue;
        R32 c = m.at(r1, var);
        if (negative) c = neg(c);
        if (ne(c, R32(1))) scale_row(m, r1, div(R32(1), c));
        bool r1_gone = false;
        for (int k2 = k1 + 1; k2 <= last; k2++) {
            const int r2 = rows[k2];
            if (removed[r2]) continue;
            c = m.at(r2, var);
            if (negative) c = neg(c);
            if (ne(c, R32(1))) scale_row(m, r2, div(R32(1), c));
            const int cres = cmp_const_rows(m, rhs, r1, r2);
            if (is_intersect) {
                if (cres == CST_LT || cres == CST_EQ) { removed[r2] = 1; any_removed = true; }
                else if (cres == CST_GT) { removed[r1] = 1; any_removed = true; r1_gone = true; }
            } else {
                if (cres == CST_LT || cres == CST_EQ) { removed[r1] = 1; any_removed = true; r1_gone = true; }
                else if (cres == CST_GT) { removed[r2] = 1; any_removed = true; }
            }
            if (r1_gone) break;
        }
    }
}

// Lineq::reduce (linsys.cpp:359-626). Returns the consistency flag; m is rewritten.
inline bool reduce_system(RM & m, int rhs, bool is_intersect)
{
    remove_iden_rows(m);
    Bounds B;
    B.pos.resize(rhs); B.neg.resize(rhs);
    std::vector<char> removed(m.r, 0);
    bool any_removed = false;
    for (int i = 0; i < m.r; i++) {                                       // :378-421
        int vars = 0, single = -1;
        for (int j = 0; j < rhs; j++) if (ne(m.at(i, j), R32(0))) { vars++; single = j; }
        if (vars == 0) {
            const int s = cmp_const_value(m, rhs, i, R32(0));
            if (s == CST_LT) return false;
            if (s == CST_EQ || s == CST_GT) { any_removed = true; removed[i] = 1; }
        } else if (vars == 1) {
            R32 c = m.at(i, single);
            if (gt(c, R32(0))) B.pos[single].push_back(i);
            else if (lt(c, R32(0))) B.neg[single].push_back(i);
        }
    }
    for (int var = 0; var < rhs; var++) {
        const std::vector<int> & P = B.pos[var];
        const std::vector<int> & N = B.neg[var];
        if (!P.empty()) tighten_side(m, rhs, var, P, false, is_intersect, removed, any_removed);
        if (!N.empty()) tighten_side(m, rhs, var, N, true, is_intersect, removed, any_removed);
        if (is_intersect && !P.empty() && !N.empty()) {                   // :577-602
            for (size_t a = 0; a < P.size(); a++) {
                const int pi = P[a];
                R32 c = m.at(pi, var);
                if (ne(c, R32(1))) scale_row(m, pi, div(R32(1), c));
                for (size_t b = 0; b < N.size(); b++) {
                    const int ni = N[b];
                    c = neg(m.at(ni, var));
                    if (ne(c, R32(1))) scale_row(m, ni, div(R32(-1), c));
                    else scale_row(m, ni, R32(-1));
                    const int cres = cmp_const_rows(m, rhs, pi, ni);
                    scale_row(m, ni, R32(-1));
                    if (cres == CST_LT) return false;
                }
            }
        }
    }
    if (any_removed) {                                                    // :606-621
        RM t(0, m.c);
        for (int i = 0; i < m.r; i++) {
            if (removed[i]) continue;
            t.a.insert(t.a.end(), m.row(i), m.row(i) + m.c);
            t.r++;
        }
        m = t.r ? t : RM();            // Matrix::copy of an empty source clears to 0 x 0 (matt.h:1156-1158)
    }
    return true;
}

// Lineq::fme (linsys.cpp:656-774): Fourier-Motzkin elimination of variable u.
inline bool fme(const RM & coeff, int rhs, int u, bool darkshadow, RM & res)
{
    res = RM();
    if (coeff.size() == 0) return true;
    RM tmp = coeff;
    std::vector<int> pos, negs;
    res = RM(0, coeff.c);
    for (int i = 0; i < coeff.r; i++) {
        bool have_vars = false;
        for (int j = 0; j < rhs && !have_vars; j++) have_vars = ne(coeff.at(i, j), R32(0));
        if (!have_vars && cmp_const_value(coeff, rhs, i, R32(0)) == CST_LT) {               // :697-702
            // the reference leaves whatever it had appended so far in 'res' (0 x 0 if nothing)
            if (res.r == 0) res = RM();
            return false;
        }
        R32 c = coeff.at(i, u);
        if (ne(c, R32(0))) {
            if (gt(c, R32(0))) {
                pos.push_back(i);
                if (ne(c, R32(1))) scale_row(tmp, i, div(R32(1), c));
            } else {
                negs.push_back(i);
                if (ne(c, R32(-1))) scale_row(tmp, i, div(R32(1), neg(c)));
                if (darkshadow) tmp.at(i, rhs) = sub(tmp.at(i, rhs), R32(1));
            }
        } else {
            res.a.insert(res.a.end(), tmp.row(i), tmp.row(i) + tmp.c);
            res.r++;
        }
    }
    const int np = (int)pos.size(), nn = (int)negs.size();
    if (np + nn == 1) {                                                   // :735-745
        const int pi = np == 1 ? pos[0] : negs[0];
        res.a.insert(res.a.end(), tmp.row(pi), tmp.row(pi) + tmp.c);
        res.r++;
    } else if (np + nn > 1) {                                             // :746-764
        for (int a = 0; a < np; a++)
            for (int b = 0; b < nn; b++) {
                std::vector<R32> row(tmp.row(negs[b]), tmp.row(negs[b]) + tmp.c);
                for (int j = 0; j < tmp.c; j++) row[j] = add(tmp.at(pos[a], j), row[j]);
                res.a.insert(res.a.end(), row.begin(), row.end());
                res.r++;
            }
    }
    if (res.r > 0) return reduce_system(res, rhs, true);
    return true;                                   // 0 x cols, as res.reinit(0, cols) leaves it
}

// Lineq::calcBound (linsys.cpp:1047-1078): for each variable j eliminate every other
// variable (innermost first) by fme; what is left bounds j alone. Returns false at the
// first inconsistent elimination (bounds of earlier variables stay filled).
inline bool calc_bound(const RM & coeff, int rhs, std::vector<RM> & limits)
{
    limits.assign(rhs, RM());
    for (int j = 0; j < rhs; j++) {
        RM work = coeff, res;
        for (int i = rhs - 1; i >= 0; i--) {
            if (i == j) continue;
            if (!fme(work, rhs, i, false, res)) return false;
            work = res;
        }
        limits[j] = work;
    }
    return true;
}

// SIX::reviseTargetFunc (lpsol.h:2053-2074).
inline void revise_target(std::vector<R32> & tgtf, const RM & eqs, const RM & leq, int rhs)
{
    for (int j = 0; j < rhs; j++) {
        bool nz = false;
        if (leq.c > 0 && !col_is_zero(leq, j)) nz = true;
        if (eqs.c > 0 && !col_is_zero(eqs, j)) nz = true;
        if (!nz) tgtf[j] = R32(0);
    }
}

// Lineq::has_solution (linsys.cpp:830-906). Returns 1/0, or ORC_REF_UNDEFINED.
inline int has_solution(const RM & leq, const RM & eqs, const RM & vc, int rhs, bool is_int, bool is_unique)
{
    if (leq.size() == 0 && eqs.size() == 0) return 0;
    Problem<R32> Q;
    Q.cols = leq.size() ? leq.c : eqs.c;
    Q.tgtf.assign(Q.cols, R32(0));
    for (int i = 0; i < rhs; i++) Q.tgtf[i] = R32(1);
    Q.vc = vc; Q.eq = eqs; Q.leq = leq;
    revise_target(Q.tgtf, eqs, leq, rhs);
    R32 v; std::vector<R32> sol;
    for (int pass = 0; pass < 2; pass++) {
        int st = is_int ? mip_solve(Q, pass == 0, false, (const uint8_t *)0, v, sol)
                        : (pass == 0 ? six_maxm(Q, 0xFFFFFFFFu, v, sol) : six_minm(Q, 0xFFFFFFFFu, v, sol));
        if (st < 0) return st;
        if (st == 0) return 1;                       // IP_SUCC == SIX_SUCC == 0
        if (!is_unique && st == 1) return 1;         // *_UNBOUND == 1
    }
    return 0;
}

// ---- Gauss-Jordan family (matt.h) -------------------------------------------------------
inline R32 abs_r(R32 v) { return lt(v, R32(0)) ? neg(v) : v; }            // matt.h:206-212

inline void swap_rows(RM & m, int a, int b)
{
    if (a == b) return;
    for (int j = 0; j < m.c; j++) { R32 t = m.at(a, j); m.at(a, j) = m.at(b, j); m.at(b, j) = t; }
}
inline void axpy_row(RM & m, int from, R32 v, int to)                     // mul_and_add_row, matt.h:1493-1501
{
    for (int j = 0; j < m.c; j++) m.at(to, j) = add(mul(m.at(from, j), v), m.at(to, j));
}

// Matrix<Rational>::rank(basis, is_unitarize) (matt.h:2614-2726). Without unitarising, the
// eliminated trapezoid is returned only when the rank is full; otherwise `basis` becomes the
// ORIGINAL rows in pivot order (matt.h:2710-2719).
inline int rank_basis(const RM & in, bool unitarize, RM & basis)
{
    RM p = in;
    std::vector<int> rowpos(p.r);
    for (int i = 0; i < p.r; i++) rowpos[i] = i;
    int rankv = 0;
    for (int row = 0, col = 0; row < p.r && col < p.c; row++, col++) {
        int swap_row = -1;
        R32 pivot = R32(0);
        for (int w = col; w < p.c; w++) {
            for (int k = row; k < p.r; k++) {
                R32 t = p.at(k, w);
                if (eq(t, R32(0))) continue;
                if (swap_row == -1) {
                    swap_row = k; pivot = t;
                    if (eq(pivot, R32(1))) break;
                } else if (eq(t, R32(1))) {
                    swap_row = k; pivot = t;
                    break;
                } else if (lt(abs_r(pivot), abs_r(t))) {
                    swap_row = k; pivot = t;
                }
            }
            if (swap_row == -1) continue;
            swap_rows(p, swap_row, row);
            std::swap(rowpos[swap_row], rowpos[row]);
            col = w;
            break;
        }
        if (swap_row == -1) break;
        if (unitarize && ne(p.at(row, col), R32(1))) scale_row(p, row, div(R32(1), p.at(row, col)));
        for (int i = unitarize ? 0 : row + 1; i < p.r; i++) {
            if (i == row || eq(p.at(i, col), R32(0))) continue;
            R32 t = div(neg(p.at(i, col)), p.at(row, col));
            axpy_row(p, row, t, i);
        }
        rankv++;
    }
    if (!unitarize && rankv < in.r) {
        p = RM(rankv, in.c);
        for (int i = 0; i < rankv; i++)
            for (int j = 0; j < in.c; j++) p.at(i, j) = in.at(rowpos[i], j);
    }
    basis = p;
    return rankv;
}

// Matrix<Rational>::null (matt.h:2546-2584): column-convention basis of the null space.
inline void null_of(const RM & in, RM & ns)
{
    RM tmp;
    rank_basis(in, true, tmp);
    ns = RM(in.c, in.c);
    for (int i = 0; i < in.c; i++) for (int j = 0; j < in.c; j++) ns.at(i, j) = R32(i == j ? 1 : 0);
    for (int nsrow = 0, row = 0; row < tmp.r; row++) {
        int col;
        bool found = false;
        for (col = row; col < tmp.c; col++)
            if (!eq(tmp.at(row, col), R32(0))) { nsrow = col; found = true; break; }
        if (!found) break;
        ns.at(nsrow, col) = R32(0);
        for (int k = col + 1; k < tmp.c; k++) ns.at(nsrow, k) = neg(tmp.at(row, k));
    }
}

// ---- INTMat (xmat.cpp:853-1030): 32-bit two's-complement integers -------------------------
typedef Mat<int32_t> IM;
inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t iabs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }

inline int32_t exgcd_rec(int32_t a, int32_t b, int32_t & x, int32_t & y)     // comf.cpp:295-307
{
    if (b == 0) { x = 1; y = 0; return a; }
    int32_t x1, y1;
    int32_t g = exgcd_rec(b, a % b, x1, y1);
    x = y1;
    y = wadd(x1, -wmul(a / b, y1));
    return g;
}
inline int32_t exgcd(int32_t a, int32_t b, int32_t & x, int32_t & y)         // comf.cpp:312-321
{
    int32_t g = exgcd_rec(a, b, x, y);
    if (g < 0) { g = -g; x = -x; y = -y; }
    return g;
}
inline int32_t sgcd(int32_t x, int32_t y)                                     // comf.cpp:226-243
{
    if (x < 0) x = -x;
    if (y < 0) y = -y;
    if (x > y) std::swap(x, y);
    while (x) { int32_t t = x; x = y % x; y = t; }
    return y;
}

inline IM imul(const IM & a, const IM & b)                                    // xmat.cpp:99-117
{
    IM c(a.r, b.c);
    for (int i = 0; i < a.r; i++)
        for (int j = 0; j < b.c; j++) {
            int32_t t = 0;
            for (int k = 0; k < a.c; k++) t = wadd(t, wmul(a.at(i, k), b.at(k, j)));
            c.at(i, j) = t;
        }
    return c;
}
inline IM ieye(int n) { IM e(n, n); for (int i = 0; i < n; i++) e.at(i, i) = 1; return e; }

// INTMat::hnf (xmat.cpp:912-992): h = this * u, h lower triangular. Written with the
// reference's own full elimination-matrix products. Returns 0, or ORC_REF_UNDEFINED where the
// x86-64 reference divides by zero (zero diagonal below row 0, xmat.cpp:956-980) or multiplies
// by a rows x cols "identity" read out of bounds (cols > rows and a negative diagonal, :936-941).
inline int int_hnf(const IM & a, IM & h, IM & u)
{
    const int n = a.c;
    u = ieye(n);
    h = a;
    const int lim = a.r < a.c ? a.r : a.c;
    for (int i = 0; i < lim; i++) {
        for (int j = i + 1; j < n; j++) {
            if (h.at(i, j) == 0) continue;
            int32_t aii = h.at(i, i), aij = h.at(i, j), x, y;                 // gen_elim_mat, :853-868
            if ((aii == INT32_MIN || aij == INT32_MIN)) return ORC_REF_UNDEFINED;
            int32_t g = exgcd(aii, aij, x, y);
            IM elim = ieye(n);
            elim.at(i, i) = x; elim.at(j, i) = y;
            elim.at(i, j) = -aij / g; elim.at(j, j) = aii / g;
            u = imul(u, elim);
            h = imul(h, elim);
        }
        if (h.at(i, i) < 0) {
            if (a.c > a.r) return ORC_REF_UNDEFINED;
            IM neg = ieye(n);
            neg.at(i, i) = -1;
            h = imul(h, neg);
            u = imul(u, neg);
        }
        for (int j = 0; j < i; j++) {
            if (h.at(i, j) >= 0) continue;
            if (h.at(i, i) == 0 || h.at(i, j) == INT32_MIN) return ORC_REF_UNDEFINED;
            int32_t v = iabs(h.at(i, j)) <= iabs(h.at(i, i)) ? 1 : iabs(h.at(i, j) / h.at(i, i)) + 1;
            IM elim = ieye(n);
            elim.at(i, j) = v;
            h = imul(h, elim);
            u = imul(u, elim);
        }
        for (int j = 0; j < i; j++) {
            if (h.at(i, j) < h.at(i, i)) continue;
            if (h.at(i, i) == 0) return ORC_REF_UNDEFINED;
            int32_t d = h.at(i, j) / h.at(i, i);
            IM elim = ieye(n);
            elim.at(i, j) = -d;
            h = imul(h, elim);
            u = imul(u, elim);
        }
    }
    return 0;
}

// INTMat::gcd (xmat.cpp:996-1030): divide each row by the gcd of its nonzero magnitudes.
inline void int_gcd(IM & m)
{
    if (m.c == 1) return;
    for (int i = 0; i < m.r; i++) {
        uint32_t mn = (uint32_t)-1;
        bool allzero = true;
        for (int j = 0; j < m.c; j++) {
            uint32_t x = (uint32_t)iabs(m.at(i, j));
            if (x != 0) { mn = mn < x ? mn : x; allzero = false; }
        }
        if (mn == 1 || mn == 0 || allzero) continue;
        uint32_t g = mn;
        for (int j = 0; j < m.c; j++) {
            uint32_t q = (uint32_t)iabs(m.at(i, j));
            if (q != 0 && q != g) {
                g = (uint32_t)sgcd((int32_t)g, (int32_t)q);
                if (g == 1) break;
            }
        }
        if (g == 1) continue;
        for (int j = 0; j < m.c; j++) m.at(i, j) = m.at(i, j) / (int32_t)g;
    }
}

// Matrix<Rational>::rank with basis == NULL (matt.h:2614-2726): unitarising Gauss-Jordan.
inline int rank_of(const RM & in)
{
    RM p = in;
    int rankv = 0;
    for (int row = 0, col = 0; row < p.r && col < p.c; row++, col++) {
        int swap_row = -1;
        R32 pivot = R32(0);
        for (int w = col; w < p.c; w++) {
            for (int k = row; k < p.r; k++) {
                R32 t = p.at(k, w);
                if (eq(t, R32(0))) continue;
                if (swap_row == -1) {
                    swap_row = k; pivot = t;
                    if (eq(pivot, R32(1))) break;
                } else if (eq(t, R32(1))) {
                    swap_row = k; pivot = t;
                    break;
                } else if (lt(abs_r(pivot), abs_r(t))) {
                    swap_row = k; pivot = t;
                }
            }
            if (swap_row == -1) continue;
            swap_rows(p, swap_row, row);
            col = w;
            break;
        }
        if (swap_row == -1) break;
        if (ne(p.at(row, col), R32(1))) scale_row(p, row, div(R32(1), p.at(row, col)));
        for (int i = 0; i < p.r; i++) {
            if (i == row || eq(p.at(i, col), R32(0))) continue;
            R32 t = div(neg(p.at(i, col)), p.at(row, col));
            axpy_row(p, row, t, i);
        }
        rankv++;
    }
    return rankv;
}

inline bool upper_tri(const RM & m)
{ for (int j = 0; j < m.c; j++) for (int i = j + 1; i < m.r; i++) if (!eq(m.at(i, j), R32(0))) return false; return true; }
inline bool lower_tri(const RM & m)
{ for (int i = 0; i < m.r; i++) for (int j = i + 1; j < m.c; j++) if (!eq(m.at(i, j), R32(0))) return false; return true; }
inline bool anti_upper_tri(const RM & m)
{ for (int i = 0; i < m.r; i++) for (int j = 0; j < m.c - 1 - i; j++) if (!eq(m.at(i, j), R32(0))) return false; return true; }
inline bool anti_lower_tri(const RM & m)
{ for (int j = 0; j < m.c; j++) for (int i = m.r - 1; i > m.r - 1 - j; i--) if (!eq(m.at(i, j), R32(0))) return false; return true; }

// Matrix<Rational>::det (matt.h:1621-1736).
inline R32 det_of(const RM & m)
{
    if (m.r != m.c) return R32(0);
    const int n = m.r;
    if (n == 1) return m.at(0, 0);
    if (n == 2) return sub(mul(m.at(0, 0), m.at(1, 1)), mul(m.at(0, 1), m.at(1, 0)));
    if (n == 3) {
        if (upper_tri(m) || lower_tri(m)) return mul(mul(m.at(0, 0), m.at(1, 1)), m.at(2, 2));
        if (anti_upper_tri(m) || anti_lower_tri(m))
            return mul(mul(mul(m.at(2, 0), m.at(1, 1)), m.at(0, 2)), R32(-1));   // (3*2/2) odd => -1
        R32 d = mul(mul(m.at(0, 0), m.at(1, 1)), m.at(2, 2));
        d = add(d, mul(mul(m.at(1, 0), m.at(2, 1)), m.at(0, 2)));
        d = add(d, mul(mul(m.at(0, 1), m.at(1, 2)), m.at(2, 0)));
        d = sub(d, mul(mul(m.at(0, 2), m.at(1, 1)), m.at(2, 0)));
        d = sub(d, mul(mul(m.at(0, 1), m.at(1, 0)), m.at(2, 2)));
        d = sub(d, mul(mul(m.at(2, 1), m.at(1, 2)), m.at(0, 0)));
        return d;
    }
    R32 d = R32(1);
    if (upper_tri(m) || lower_tri(m)) {
        for (int i = 0; i < n; i++) d = mul(d, m.at(i, i));
        return d;
    }
    if (anti_upper_tri(m) || anti_lower_tri(m)) {
        for (int i = 0; i < n; i++) d = mul(d, m.at(i, n - 1 - i));
        return d;
    }
    RM a = m;
    int swaps = 0;
    for (int j = 0; j < n; j++) {
        int swap_row = -1;
        R32 entry;
        for (int k = j; k < n; k++) {
            R32 t = a.at(k, j);
            if (eq(t, R32(0))) continue;
            if (swap_row == -1) { swap_row = k; entry = t; if (eq(entry, R32(1))) break; }
            else if (eq(t, R32(1))) { swap_row = k; break; }
            else if (lt(abs_r(entry), abs_r(t))) { swap_row = k; entry = t; }
        }
        if (swap_row == -1) return R32(0);
        if (swap_row != j) { swap_rows(a, swap_row, j); swaps++; }
        for (int i = j + 1; i < n; i++)
            if (!eq(a.at(i, j), R32(0))) axpy_row(a, j, neg(div(a.at(i, j), a.at(j, j))), i);
    }
    for (int j = 0; j < n; j++) d = mul(d, a.at(j, j));
    if (swaps & 1) d = neg(d);
    return d;
}

// Matrix<Rational>::inv (matt.h:1743-1845). Returns false when singular.
inline bool inverse_of(const RM & m, RM & e)
{
    if (m.r != m.c) return false;
    const int n = m.r;
    RM p = m;
    e = RM(n, n);
    if (n == 1) { e.at(0, 0) = div(R32(1), p.at(0, 0)); return true; }
    if (n == 2) {
        R32 k = sub(mul(p.at(0, 0), p.at(1, 1)), mul(p.at(0, 1), p.at(1, 0)));
        if (eq(k, R32(0))) return false;
        k = div(R32(1), k);
        e.at(0, 0) = p.at(1, 1); e.at(1, 1) = p.at(0, 0);
        e.at(0, 1) = mul(R32(-1), p.at(0, 1)); e.at(1, 0) = mul(R32(-1), p.at(1, 0));
        // Matrix::mul(k), matt.h:1331-1348: zero test first, then the == 1 shortcut
        if (eq(k, R32(0))) { for (size_t t = 0; t < e.a.size(); t++) e.a[t] = R32(0); }
        else if (!eq(k, R32(1))) { for (size_t t = 0; t < e.a.size(); t++) e.a[t] = mul(e.a[t], k); }
        return true;
    }
    for (int i = 0; i < n; i++) e.at(i, i) = R32(1);
    for (int j = 0; j < n; j++) {
        int swap_row = -1;
        R32 entry;
        for (int k = j; k < n; k++) {
            R32 t = p.at(k, j);
            if (eq(t, R32(0))) continue;
            if (swap_row == -1) { swap_row = k; entry = t; if (eq(entry, R32(1))) break; }
            else if (eq(t, R32(1))) { swap_row = k; break; }
            else if (lt(abs_r(entry), abs_r(t))) { swap_row = k; entry = t; }
        }
        if (swap_row == -1) return false;
        if (swap_row != j) { swap_rows(p, swap_row, j); swap_rows(e, swap_row, j); }
        if (ne(p.at(j, j), R32(1))) {
            R32 t = div(R32(1), p.at(j, j));
            scale_row(p, j, t); scale_row(e, j, t);
        }
        for (int i = 0; i < n; i++) {
            if (i == j || eq(p.at(i, j), R32(0))) continue;
            R32 t = mul(R32(-1), p.at(i, j));
            axpy_row(p, j, t, i); axpy_row(e, j, t, i);
        }
    }
    return true;
}

} // namespace orc
#endif
