// TEST INFRASTRUCTURE -- drop-in check with the REAL xpoly types.
// Compiled in the authoring container against /root/reference/src/com (headers
// and objects where they lie) by `make -C oracle ref`; the binary lands in
// oracle/_ref/dropin_demo and travels to the GPU box. It solves the same problems
// once with the reference's own xcom::SIX<Mat,T> (CPU) and once with
// xpoly_amd::SIX<Mat,T> (include/xpoly_amd/six.hpp -> C ABI -> GPU), on the
// reference's FloatMat / RMat objects, and compares status, optimum and solution
// bit for bit. Exit code 0 = all equal.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <signal.h>
#include <unistd.h>
#include <vector>

#include "ltype.h"
#include "comf.h"
#include "smempool.h"
#include "strbuf.h"
#include "rational.h"
#include "flty.h"
#include "sstl.h"
#include "matt.h"
#include "xmat.h"
#include "bs.h"
#include "sbs.h"
#include "sgraph.h"
#include "lpsol.h"
#include "linsys.h"

#include "xpoly_amd/six.hpp"
#include "xpoly_amd/lineq.hpp"

namespace xpoly_amd {
template <> struct scalar_kind<xcom::Float> { static const int value = 0; };
template <> struct scalar_kind<xcom::Rational> { static const int value = 1; };
}

using namespace xcom;

// Lineq::has_solution's body (src/com/linsys.cpp:842-906) exactly as the reference has it, compiled with the
// substitution INTEGRATION.md section 2 prescribes and nothing else. The text is cut out of the reference by
// oracle/Makefile while this file is being compiled and is not kept (HAS_SOLUTION_BODY names the temporary).
namespace subst {
using xpoly_amd::SIX;
using xpoly_amd::MIP;
static bool has_solution(RMat const& leq, RMat const& eq, RMat & vc, UINT rhs_idx, bool is_int_sol, bool is_unique_sol)
#include HAS_SOLUTION_BODY
}

// where the demo is, for the handler below: the reference divides integers by zero on some inputs (rational.cpp's
// reduce / operator/ have no guard), which must be reported as the reference's crash, not as a silent exit
static const char * g_where = "start";
static int g_case = -1;
static void on_fpe(int) { fprintf(stderr, "SIGFPE (integer division by zero) in: %s, case %d\n", g_where, g_case); _exit(3); }

static int same_cells(RMat const & a, RMat const & b)
{
    if (a.get_row_size() != b.get_row_size()) return 0;
    if (a.size() == 0 || b.size() == 0) return a.size() == b.size();
    return a.get_col_size() == b.get_col_size() && memcmp(a.get_matrix(), b.get_matrix(), sizeof(Rational) * a.size()) == 0;
}

static unsigned long long rng_state = 88172645463325252ULL;
static unsigned long long xs()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static int irand(int lo, int hi) { return lo + (int)(xs() % (unsigned long long)(hi - lo + 1)); }

template <class Mat, class T> static int compare_one(Mat & tgtf, Mat & vc, Mat & eq, Mat & leq, bool is_max, const char * tag)
{
    T v_ref, v_gpu;
    Mat s_ref, s_gpu;
    xcom::SIX<Mat, T> ref;
    xpoly_amd::SIX<Mat, T> gpu;
    UINT a = is_max ? ref.maxm(v_ref, s_ref, tgtf, vc, eq, leq) : ref.minm(v_ref, s_ref, tgtf, vc, eq, leq);
    UINT b = is_max ? gpu.maxm(v_gpu, s_gpu, tgtf, vc, eq, leq) : gpu.minm(v_gpu, s_gpu, tgtf, vc, eq, leq);
    int bad = (a != b) || memcmp(&v_ref, &v_gpu, sizeof(T)) != 0;
    if (!bad && a == SIX_SUCC)
        bad = s_ref.get_col_size() != s_gpu.get_col_size() ||
              memcmp(s_ref.get_matrix(), s_gpu.get_matrix(), sizeof(T) * s_ref.get_col_size()) != 0;
    if (bad) printf("MISMATCH %s %s: reference status %u, xpoly_amd status %u\n", tag, is_max ? "maxm" : "minm", a, b);
    return bad;
}

// ---- --one-call: the reference's own call pattern -- ONE tiny problem per call -- timed on one host core with the reference's
// classes and with the adapter classes (include/xpoly_amd/*.hpp -> C ABI -> GPU) on the same objects. bench.py leg `one_call`.
#include <time.h>
static double now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
template <class F> static double time_us(F && f, double budget_us = 3e5, int min_reps = 5)
{
    f();                                                              // warm (first launch of a kernel, caches)
    int reps = 0;
    const double t0 = now_us();
    double t1 = t0;
    while (reps < min_reps || t1 - t0 < budget_us) { f(); reps++; t1 = now_us(); if (reps >= 20000) break; }
    return (t1 - t0) / reps;
}
static void fill_lp(RMat & leq, RMat & tgtf, RMat & vc, int m, int nv)
{
    leq.reinit(m, nv + 1); tgtf.reinit(1, nv + 1); vc.reinit(nv, nv + 1);
    for (int i = 0; i < m; i++) { for (int j = 0; j < nv; j++) leq.setr(i, j, irand(1, 9), 1); leq.setr(i, nv, nv * irand(3, 5), 1); }
    for (int j = 0; j < nv; j++) { tgtf.setr(0, j, irand(1, 9), 1); vc.setr(j, j, -1, 1); }
}
static int one_call()
{
    printf("{\"cases\": [");
    const char * sep = "";
    for (int sz = 0; sz < 2; sz++) {                                  // SIX<RMat,Rational>::maxm, dense integer data (14 x 6, 28 x 12)
        const int m = sz ? 28 : 14, nv = sz ? 12 : 6;
        RMat leq, tgtf, vc, eq;
        fill_lp(leq, tgtf, vc, m, nv);
        Rational v1, v2; RMat s1, s2;
        UINT a = 0, b = 0;
        const double tr = time_us([&] { xcom::SIX<RMat, Rational> six; a = six.maxm(v1, s1, tgtf, vc, eq, leq); });
        const double tg = time_us([&] { xpoly_amd::SIX<RMat, Rational> six; b = six.maxm(v2, s2, tgtf, vc, eq, leq); });
        const int same = a == b && memcmp(&v1, &v2, sizeof(Rational)) == 0;
        printf("%s{\"call\": \"SIX<RMat,Rational>::maxm\", \"shape\": \"%dx%d\", \"status\": %u, \"reference_us\": %.2f, \"adapter_us\": %.2f, \"same_result\": %s}",
               sep, m, nv, a, tr, tg, same ? "true" : "false");
        sep = ", ";
    }
    {   // Lineq::has_solution(integer, unique) on a dependence-polyhedron-like system (12 rows, 4 variables)
        const int rows = 12, nv = 4;
        RMat sys(rows, nv + 1), vc(nv, nv + 1), eq;
        for (int i = 0; i < rows; i++) for (int j = 0; j <= nv; j++) sys.setr(i, j, j < nv ? irand(-3, 3) : irand(0, 9), 1);
        for (int j = 0; j < nv; j++) vc.setr(j, j, -1, 1);
        bool ha = false, hb = false;
        xcom::Lineq ref(NULL); xpoly_amd::Lineq<RMat> gpu(NULL);
        const double tr = time_us([&] { RMat v = vc; ha = ref.has_solution(sys, eq, v, nv, true, true); });
        const double tg = time_us([&] { RMat v = vc; hb = gpu.has_solution(sys, eq, v, nv, true, true); });
        printf("%s{\"call\": \"Lineq::has_solution(int, unique)\", \"shape\": \"%dx%d\", \"status\": %d, \"reference_us\": %.2f, \"adapter_us\": %.2f, \"same_result\": %s}",
               sep, rows, nv + 1, (int)ha, tr, tg, ha == hb ? "true" : "false");
    }
    {   // MIP<RMat,Rational>::maxm, 0-1 knapsack of 24 variables with two capacity rows (bench.py leg `mip`)
        const int nv = 24, m = 2;
        RMat leq(m + nv, nv + 1), tgtf(1, nv + 1), vc(nv, nv + 1), eq;
        for (int i = 0; i < m; i++) { int sum = 0; for (int j = 0; j < nv; j++) { const int x = irand(1, 9); sum += x; leq.setr(i, j, x, 1); } leq.setr(i, nv, sum / 2, 1); }
        for (int j = 0; j < nv; j++) { leq.setr(m + j, j, 1, 1); leq.setr(m + j, nv, 1, 1); tgtf.setr(0, j, irand(1, 9), 1); vc.setr(j, j, -1, 1); }
        Rational v1, v2; RMat s1, s2;
        UINT a = 0, b = 0;
        const double tr = time_us([&] { xcom::MIP<RMat, Rational> mip; a = mip.maxm(v1, s1, tgtf, vc, eq, leq, true, NULL); }, 1e6, 2);
        const double tg = time_us([&] { xpoly_amd::MIP<RMat, Rational> mip; b = mip.maxm(v2, s2, tgtf, vc, eq, leq, true, NULL); }, 1e6, 2);
        const int same = a == b && memcmp(&v1, &v2, sizeof(Rational)) == 0;
        printf(", {\"call\": \"MIP<RMat,Rational>::maxm(is_bin)\", \"shape\": \"%dx%d\", \"status\": %u, \"reference_us\": %.2f, \"adapter_us\": %.2f, \"same_result\": %s}",
               m + nv, nv + 1, a, tr, tg, same ? "true" : "false");
    }
    for (int sz = 0; sz < 2; sz++) {                                  // Lineq::reduce and Lineq::fme on ONE system
        const int rows = sz ? 40 : 16, nv = sz ? 12 : 8;
        RMat sys(rows, nv + 1);
        for (int i = 0; i < rows; i++) for (int j = 0; j <= nv; j++) sys.setr(i, j, j < nv ? (irand(0, 9) < 7 ? irand(-3, 3) : 0) : irand(-5, 8), 1);
        bool ra = false, rb = false;
        xcom::Lineq ref(NULL); xpoly_amd::Lineq<RMat> gpu(NULL);
        const double tr = time_us([&] { RMat w = sys; ra = ref.reduce(w, nv, true); });
        const double tg = time_us([&] { RMat w = sys; rb = gpu.reduce(w, nv, true); });
        printf(", {\"call\": \"Lineq::reduce\", \"shape\": \"%dx%d\", \"status\": %d, \"reference_us\": %.2f, \"adapter_us\": %.2f, \"same_result\": %s}",
               rows, nv + 1, (int)ra, tr, tg, ra == rb ? "true" : "false");
        RMat f1, f2;
        bool fa = false, fb = false;
        const double tr2 = time_us([&] { xcom::Lineq r2(&sys, nv); fa = r2.fme(0, f1, false); });
        const double tg2 = time_us([&] { xpoly_amd::Lineq<RMat> g2(&sys, nv); fb = g2.fme(0, f2, false); });
        printf(", {\"call\": \"Lineq::fme\", \"shape\": \"%dx%d\", \"status\": %d, \"reference_us\": %.2f, \"adapter_us\": %.2f, \"same_result\": %s}",
               rows, nv + 1, (int)fa, tr2, tg2, (fa == fb && same_cells(f1, f2)) ? "true" : "false");
    }
    printf("]}\n");
    return 0;
}

int main(int argc, char ** argv)
{
    signal(SIGFPE, on_fpe);
    if (argc > 1 && !strcmp(argv[1], "--one-call")) return one_call();
    int bad = 0, n = 0, undefined = 0;
    {   // src/example/example.cpp:54-93
        FloatMat leq(2, 3), tgtf(1, 3), vc(2, 3), eq;
        double l[6] = {2, -1, 2, 1, -5, -4}, t[3] = {2, -1, 0};
        for (int i = 0; i < 6; i++) leq.set(i / 3, i % 3, Float(l[i]));
        for (int i = 0; i < 3; i++) tgtf.set(0, i, Float(t[i]));
        vc.set(0, 0, Float(-1.0)); vc.set(1, 1, Float(-1.0));
        bad += compare_one<FloatMat, Float>(tgtf, vc, eq, leq, true, "example-float"); n++;
    }
    {   // src/example/example.cpp:106-174
        RMat leq(8, 6), tgtf(1, 6), vc(5, 6), eq;
        int l[48] = {-1,0,0,0,0,-10, -1,-1,0,0,0,-8, -1,-1,-1,0,0,-9, -1,-1,-1,-1,0,-11,
                     0,-1,-1,-1,-1,-13, 0,0,-1,-1,-1,-8, 0,0,0,-1,-1,-5, 0,0,0,0,-1,-3};
        for (int i = 0; i < 48; i++) leq.setr(i / 6, i % 6, l[i], 1);
        for (int i = 0; i < 5; i++) { tgtf.setr(0, i, 1, 1); vc.setr(i, i, -1, 1); }
        bad += compare_one<RMat, Rational>(tgtf, vc, eq, leq, true, "example-rational"); n++;
        bad += compare_one<RMat, Rational>(tgtf, vc, eq, leq, false, "example-rational"); n++;
    }
    for (int it = 0; it < 40; it++) {   // random small integer LPs, both scalars
        int m = irand(1, 7), nv = irand(1, 7);
        RMat rl(m, nv + 1), rt(1, nv + 1), rv(nv, nv + 1), req;
        FloatMat fl(m, nv + 1), ft(1, nv + 1), fv(nv, nv + 1), feq;
        for (int i = 0; i < m; i++)
            for (int j = 0; j <= nv; j++) {
                int x = j < nv ? irand(-3, 6) : irand(-2, 20);
                rl.setr(i, j, x, 1); fl.set(i, j, Float((double)x));
            }
        for (int j = 0; j < nv; j++) {
            int c = irand(-2, 6);
            rt.setr(0, j, c, 1); ft.set(0, j, Float((double)c));
            rv.setr(j, j, -1, 1); fv.set(j, j, Float(-1.0));
        }
        for (int mx = 0; mx < 2; mx++) {
            bad += compare_one<RMat, Rational>(rt, rv, req, rl, mx == 0, "random-rational"); n++;
            bad += compare_one<FloatMat, Float>(ft, fv, feq, fl, mx == 0, "random-float"); n++;
        }
    }
    for (int it = 0; it < 24; it++) {   // MIP<RMat,Rational> on small integer / 0-1 knapsack-like problems
        bool is_bin = (it & 1) != 0;
        int m = is_bin && (it & 2) ? 1 : irand(1, 4), nv = irand(2, 5);
        RMat leq(m + (is_bin ? nv : 0), nv + 1), tgtf(1, nv + 1), vc(nv, nv + 1), eq;
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < nv; j++) leq.setr(i, j, irand(1, 6), 1);
            leq.setr(i, nv, irand(4, 4 * nv + 2), 1);
        }
        if (is_bin)
            for (int j = 0; j < nv; j++) { leq.setr(m + j, j, 1, 1); leq.setr(m + j, nv, 1, 1); }
        for (int j = 0; j < nv; j++) { tgtf.setr(0, j, irand(1, 8), 1); vc.setr(j, j, -1, 1); }
        Rational v_ref, v_gpu;
        RMat s_ref, s_gpu;
        xcom::MIP<RMat, Rational> ref;
        xpoly_amd::MIP<RMat, Rational> gpu;
        UINT a = ref.maxm(v_ref, s_ref, tgtf, vc, eq, leq, is_bin, NULL);
        UINT b = gpu.maxm(v_gpu, s_gpu, tgtf, vc, eq, leq, is_bin, NULL);   // the reference's spelling (linsys.cpp:864-866)
        if (b == (UINT)XPG_ERR_REF_UNDEFINED) {
            // convertEq2Ineq reads the equality row at the inequality's row index (lpsol.h:1232);
            // with more inequality rows than columns that is a read past the buffer in the
            // reference, so there is nothing to compare with -- xpoly_amd refuses the input.
            undefined++;
            continue;
        }
        int mis = (a != b) || memcmp(&v_ref, &v_gpu, sizeof(Rational)) != 0;
        if (!mis && a == IP_SUCC)
            mis = memcmp(s_ref.get_matrix(), s_gpu.get_matrix(), sizeof(Rational) * s_ref.get_col_size()) != 0;
        if (mis) printf("MISMATCH MIP %s: reference status %u, xpoly_amd status %u\n", is_bin ? "0-1" : "integer", a, b);
        bad += mis; n++;
    }
    // ---- SIX::TwoStageMethod (lpsol.h:291-301) with the reference's own Vector<bool> / Vector<INT> in/outs
    for (int it = 0; it < 24; it++) {
        int m = irand(2, 7), nv = irand(2, 7);
        for (int flavour = 0; flavour < 2; flavour++) {
            unsigned K = it % 3 == 0 ? 2u : 0xFFFFFFFFu;
            if (flavour == 0) {
                RMat l1(m, nv + 1), t1(1, nv + 1), v1(nv, nv + 1), l2, t2, v2, s1, s2;
                for (int i = 0; i < m; i++) for (int j = 0; j <= nv; j++) l1.setr(i, j, j < nv ? irand(0, 6) : irand(3, 20), 1);
                for (int j = 0; j < nv; j++) { t1.setr(0, j, irand(1, 6), 1); v1.setr(j, j, -1, 1); }
                l2 = l1; t2 = t1; v2 = v1;
                Rational mv1, mv2; Vector<bool> n1, b1, n2, b2; Vector<INT> be1, eb1, be2, eb2; INT r1 = nv, r2 = nv;
                xcom::SIX<RMat, Rational> ref; ref.set_param(0, K);
                xpoly_amd::SIX<RMat, Rational> gpu; gpu.set_param(0, K);
                UINT a = ref.TwoStageMethod(l1, v1, t1, s1, mv1, n1, b1, be1, eb1, r1);
                UINT b = gpu.TwoStageMethod(l2, v2, t2, s2, mv2, n2, b2, be2, eb2, r2);
                int mis = a != b || r1 != r2 || l1.get_row_size() != l2.get_row_size() || l1.get_col_size() != l2.get_col_size();
                if (!mis) mis = memcmp(l1.get_matrix(), l2.get_matrix(), sizeof(Rational) * l1.size()) != 0 ||
                                memcmp(t1.get_matrix(), t2.get_matrix(), sizeof(Rational) * t1.size()) != 0;
                for (INT i = 0; !mis && i < r1; i++) mis = n1.get(i) != n2.get(i) || b1.get(i) != b2.get(i) || be1.get(i) != be2.get(i);
                for (UINT i = 0; !mis && i < l1.get_row_size(); i++) mis = eb1.get(i) != eb2.get(i);
                if (!mis && a == SIX_SUCC) mis = memcmp(&mv1, &mv2, sizeof(Rational)) != 0;
                if (mis) printf("MISMATCH TwoStageMethod rational: reference status %u, xpoly_amd status %u\n", a, b);
                bad += mis; n++;
            } else {
                FloatMat l1(m, nv + 1), t1(1, nv + 1), v1(nv, nv + 1), l2, t2, v2, s1, s2;
                for (int i = 0; i < m; i++) for (int j = 0; j <= nv; j++) l1.set(i, j, Float((double)(j < nv ? irand(0, 6) : irand(3, 20))));
                for (int j = 0; j < nv; j++) { t1.set(0, j, Float((double)irand(1, 6))); v1.set(j, j, Float(-1.0)); }
                l2 = l1; t2 = t1; v2 = v1;
                Float mv1, mv2; Vector<bool> n1, b1, n2, b2; Vector<INT> be1, eb1, be2, eb2; INT r1 = nv, r2 = nv;
                xcom::SIX<FloatMat, Float> ref; ref.set_param(0, K);
                xpoly_amd::SIX<FloatMat, Float> gpu; gpu.set_param(0, K);
                UINT a = ref.TwoStageMethod(l1, v1, t1, s1, mv1, n1, b1, be1, eb1, r1);
                UINT b = gpu.TwoStageMethod(l2, v2, t2, s2, mv2, n2, b2, be2, eb2, r2);
                int mis = a != b || r1 != r2 || l1.get_col_size() != l2.get_col_size();
                if (!mis) mis = memcmp(l1.get_matrix(), l2.get_matrix(), sizeof(Float) * l1.size()) != 0 ||
                                memcmp(t1.get_matrix(), t2.get_matrix(), sizeof(Float) * t1.size()) != 0;
                for (UINT i = 0; !mis && i < l1.get_row_size(); i++) mis = eb1.get(i) != eb2.get(i);
                if (mis) printf("MISMATCH TwoStageMethod float: reference status %u, xpoly_amd status %u\n", a, b);
                bad += mis; n++;
            }
        }
    }
    // ---- the Lineq call sites of the dependence test (src/eng/poly.cpp:530-573, src/com/linsys.cpp:884) through
    // xcom::Lineq and xpoly_amd::Lineq<RMat>: reduce, has_solution, fme
    for (int it = 0; it < 40; it++) {
        int rows = irand(2, 9), nv = irand(1, 4);
        RMat sys1(rows, nv + 1), sys2, vc(nv, nv + 1), eq;
        for (int i = 0; i < rows; i++) for (int j = 0; j <= nv; j++) sys1.setr(i, j, j < nv ? irand(-3, 3) : irand(-4, 9), 1);
        for (int j = 0; j < nv; j++) vc.setr(j, j, -1, 1);
        sys2 = sys1;
        xcom::Lineq ref(NULL);
        xpoly_amd::Lineq<RMat> gpu(NULL);
        g_where = "Lineq::reduce"; g_case = it;
        bool ra = ref.reduce(sys1, nv, true), rb = gpu.reduce(sys2, nv, true);
        int mis = ra != rb || sys1.get_row_size() != sys2.get_row_size() ||
                  (sys1.size() && memcmp(sys1.get_matrix(), sys2.get_matrix(), sizeof(Rational) * sys1.size()) != 0);
        if (!mis && ra && sys1.get_row_size() > 0 && sys1.get_row_size() <= sys1.get_col_size()) {
            // (more rows than columns: convertEq2Ineq is out of bounds in the reference, lpsol.h:1232)
            RMat vc2 = vc;
            g_where = "Lineq::has_solution(int, unique)";
            bool ha = ref.has_solution(sys1, eq, vc, nv, true, true), hb = gpu.has_solution(sys2, eq, vc2, nv, true, true);
            mis = ha != hb;
            // ... and the reference's own has_solution BODY running on xpoly_amd::SIX / MIP (reviseTargetFunc, the literal
            // NULL indicator, rhs_idx), integer and rational question, unique or not
            for (int q = 0; !mis && q < 4; q++) {
                RMat vc3 = vc, vc4 = vc;
                g_where = (q & 1) ? "has_solution body, integer" : "has_solution body, rational";
                bool hr = ref.has_solution(sys1, eq, vc3, nv, (q & 1) != 0, (q & 2) != 0);
                bool hs = subst::has_solution(sys2, eq, vc4, nv, (q & 1) != 0, (q & 2) != 0);
                if (hr != hs) { mis = 1; printf("  substituted has_solution body: reference %d, on xpoly_amd %d (int %d, unique %d)\n", (int)hr, (int)hs, q & 1, (q >> 1) & 1); }
            }
            n += 4;
        }
        if (!mis && ra && sys1.get_row_size() > 1) {
            RMat f1, f2;
            g_where = "Lineq::fme";
            xcom::Lineq r2(&sys1, nv); xpoly_amd::Lineq<RMat> g2(&sys2, nv);
            bool fa = r2.fme(0, f1, false), fb = g2.fme(0, f2, false);
            mis = fa != fb || f1.get_row_size() != f2.get_row_size() ||
                  (f1.size() && memcmp(f1.get_matrix(), f2.get_matrix(), sizeof(Rational) * f1.size()) != 0);
        }
        if (!mis && ra && sys1.get_row_size() > 1) {
            // Lineq::calcBound with the reference's own signature, List<RMat*> (linsys.h:151; call site linsys.cpp:312)
            RMat a1[4], a2[4];
            g_where = "Lineq::calcBound";
            List<RMat*> lim1, lim2;
            for (int j = 0; j < nv; j++) { lim1.append_tail(&a1[j]); lim2.append_tail(&a2[j]); }
            xcom::Lineq r3(&sys1, nv); xpoly_amd::Lineq<RMat> g3(&sys2, nv);
            bool ca = r3.calcBound(lim1), cb = g3.calcBound(lim2);
            mis = ca != cb;
            for (int j = 0; !mis && ca && j < nv; j++) {
                // (an empty bound: the reference leaves whatever shape its last fme left, 0 x 0 or 0 x cols -- no cells either way)
                mis = a1[j].get_row_size() != a2[j].get_row_size() || (a1[j].size() != 0 && a1[j].get_col_size() != a2[j].get_col_size()) ||
                      (a1[j].size() && memcmp(a1[j].get_matrix(), a2[j].get_matrix(), sizeof(Rational) * a1[j].size()) != 0);
                if (mis) printf("  variable %d: reference %u x %u, xpoly_amd %u x %u\n", j, a1[j].get_row_size(), a1[j].get_col_size(),
                                a2[j].get_row_size(), a2[j].get_col_size());
            }
            if (mis) printf("MISMATCH calcBound(List<RMat*>) on system %d (reference %d, xpoly_amd %d)\n", it, (int)ca, (int)cb);
        }
        if (mis) printf("MISMATCH Lineq adapter on system %d\n", it);
        bad += mis; n++;
    }
    // ---- the host-side members next to them: reviseTargetFunc (lpsol.h:2053-2074, :2412-2420), appendEquation
    // (linsys.cpp:922), formatBound (:948), initVarConstraint (:803), is_consistent (:779)
    for (int it = 0; it < 40; it++) {
        int rows = irand(2, 7), nv = irand(1, 4), nsym = it % 3 == 0 ? 1 : 0, cols = nv + 1 + nsym;
        RMat sys1(rows, cols), sys2, e(irand(1, 2), cols), t1(1, cols), t2, t3, none;
        for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) { const int x = j < nv ? (irand(0, 2) ? irand(-3, 3) : 0) : irand(-4, 9); sys1.set(i, j, x == 0 ? Rational(0) : Rational(x, irand(1, 3))); }   // (canonical cells: 0/3 is not == 0 to the reference, and 1 / it divides by zero)
        if (it % 4 == 0) for (int i = 0; i < rows; i++) sys1.setr(i, irand(0, nv - 1) , 0, 1);      // an all-zero column now and then
        for (UINT i = 0; i < e.get_row_size(); i++) for (int j = 0; j < cols; j++) e.setr(i, j, it % 4 == 0 && j < nv ? 0 : irand(-2, 2), 1);
        for (int j = 0; j < nv; j++) t1.setr(0, j, 1, 1);
        sys2 = sys1; t2 = t1; t3 = t1;
        int mis = 0;
        g_where = "reviseTargetFunc"; g_case = it;
        {   xcom::SIX<RMat, Rational> rs; xpoly_amd::SIX<RMat, Rational> gs; xpoly_amd::MIP<RMat, Rational> gm;
            RMat ta = t1, tb = t1;
            rs.reviseTargetFunc(t1, e, sys1, nv); gs.reviseTargetFunc(t2, e, sys2, nv); gm.reviseTargetFunc(t3, e, sys2, nv);
            rs.reviseTargetFunc(ta, none, sys1, nv); gs.reviseTargetFunc(tb, none, sys2, nv);
            mis = !same_cells(t1, t2) || !same_cells(t1, t3) || !same_cells(ta, tb);
            if (mis) printf("MISMATCH reviseTargetFunc on system %d\n", it);
        }
        if (!mis) {
            RMat a1 = sys1, a2 = sys2;
            g_where = "appendEquation";
            xcom::Lineq r(&a1, nv); xpoly_amd::Lineq<RMat> g(&a2, nv);
            r.appendEquation(e); g.appendEquation(e);
            mis = !same_cells(a1, a2);
            if (mis) printf("MISMATCH appendEquation on system %d\n", it);
        }
        for (int u = 0; !mis && u < nv; u++) {
            RMat f1, f2;
            g_where = "formatBound";
            xcom::Lineq r(&sys1, nv); xpoly_amd::Lineq<RMat> g(&sys2, nv);
            r.formatBound(u, f1); g.formatBound(u, f2);
            mis = !same_cells(f1, f2);
            if (mis) printf("MISMATCH formatBound(%d) on system %d\n", u, it);
        }
        if (!mis) {
            RMat v1, v2, v3, v4;
            g_where = "initVarConstraint";
            Vector<INT> sign;
            for (int j = 0; j < nv; j++) sign.set(j, irand(-1, 1));
            xcom::Lineq r(NULL); xpoly_amd::Lineq<RMat> g(NULL);
            r.initVarConstraint(&sign, v1, nv); g.initVarConstraint(&sign, v2, nv);
            r.initVarConstraint(NULL, v3, nv); g.initVarConstraint(NULL, v4, nv);
            mis = !same_cells(v1, v2) || !same_cells(v3, v4);
            if (mis) printf("MISMATCH initVarConstraint on system %d\n", it);
        }
        if (!mis && nsym == 0) {
            xcom::Lineq r(&sys1, nv); xpoly_amd::Lineq<RMat> g(&sys2, nv);
            g_where = "is_consistent";
            bool c1 = r.is_consistent(), c2 = g.is_consistent();
            mis = c1 != c2 || !same_cells(sys1, sys2);
            if (mis) printf("MISMATCH is_consistent on system %d: reference %d, xpoly_amd %d\n", it, (int)c1, (int)c2);
        }
        bad += mis; n++;
    }
    // ---- the collectors for the other hot callers (src/eng/ldtran.cpp:178-193, src/eng/poly.cpp:4803-4821): many systems of mixed
    // shapes, one elimination level / one calcBound per call, against the reference's Lineq called system by system
    {
        const int NS = 36;
        std::vector<RMat> sys((size_t)NS);
        std::vector<RMat *> ptr;
        std::vector<int32_t> u, rhs;
        for (int k = 0; k < NS; k++) {
            const int rows = 3 + (k % 3) * 3, nv = 2 + (k % 2);                       // 3 x 3, 6 x 4, 9 x 3, ... (six shape classes)
            sys[(size_t)k].reinit(rows, nv + 1);
            for (int i = 0; i < rows; i++) for (int j = 0; j <= nv; j++) sys[(size_t)k].setr(i, j, j < nv ? irand(-3, 3) : irand(-4, 9), 1);
            ptr.push_back(&sys[(size_t)k]); u.push_back(irand(0, nv - 1)); rhs.push_back(nv);
        }
        std::vector<RMat> res; std::vector<int32_t> ok;
        g_where = "fme_all"; g_case = 0;
        int rc = xpoly_amd::fme_all(ptr, u, rhs, res, ok);
        int mis = rc != 0;
        for (int k = 0; !mis && k < NS; k++) {
            RMat f1; xcom::Lineq r(&sys[(size_t)k], rhs[(size_t)k]);
            g_case = k;
            const bool fa = r.fme((UINT)u[(size_t)k], f1, false);
            mis = fa != (ok[(size_t)k] != 0) || !same_cells(f1, res[(size_t)k]);
            if (mis) printf("MISMATCH fme_all on system %d (reference %d, collector %d)\n", k, (int)fa, (int)ok[(size_t)k]);
        }
        bad += mis; n++;
        std::vector<std::vector<RMat> > lim; std::vector<int32_t> cok;
        g_where = "calc_bound_all";
        rc = xpoly_amd::calc_bound_all(ptr, rhs, lim, cok);
        mis = rc != 0;
        for (int k = 0; !mis && k < NS; k++) {
            RMat a1[4]; List<RMat*> l1;
            for (int j = 0; j < rhs[(size_t)k]; j++) l1.append_tail(&a1[j]);
            g_case = k;
            xcom::Lineq r(&sys[(size_t)k], rhs[(size_t)k]);
            const bool ca = r.calcBound(l1);
            mis = ca != (cok[(size_t)k] > 0);
            for (int j = 0; !mis && ca && j < rhs[(size_t)k]; j++)
                mis = a1[j].get_row_size() != lim[(size_t)k][(size_t)j].get_row_size() ||
                      (a1[j].size() && memcmp(a1[j].get_matrix(), lim[(size_t)k][(size_t)j].get_matrix(), sizeof(Rational) * a1[j].size()) != 0);
            if (mis) printf("MISMATCH calc_bound_all on system %d (reference %d, collector %d)\n", k, (int)ca, (int)cok[(size_t)k]);
        }
        bad += mis; n++;
    }
    printf("dropin_demo: %d solves through xcom::SIX / MIP and xpoly_amd::SIX / MIP on the reference's own matrix types, "
           "%d refused as undefined in the reference, %d mismatches\n", n, undefined, bad);
    return bad ? 1 : 0;
}
