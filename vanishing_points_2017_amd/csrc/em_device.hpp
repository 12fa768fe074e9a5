// em_device.hpp -- device-resident EM refinement of vanishing points, one workgroup per image.
//
// MI355X-first design: the whole EM of one image (setup, E-step, N x N smoothing, M-step,
// split / merge / finalisation control flow) runs inside ONE persistent workgroup, so an
// iteration costs barriers instead of kernel launches and the only HBM stream that matters is
// the fp64 N x N line-similarity matrix read once per E-step (B_EM).  Hundreds
// of images run concurrently (one per workgroup), which is where the throughput comes from.
//
// All arithmetic is fp64 (fp32 only where the reference is fp32: the prior weights).  The
// translation unit is compiled with -ffp-contract=off so that products and sums round like the
// reference's separate NumPy ufunc calls; the two bandwidth-bound accumulation loops use an
// explicit fma().
//
// Conventions: every VPK_DEVFN below is called by ALL threads of the workgroup with uniform
// arguments, expects its inputs to be visible (a barrier has happened) and ends with a barrier.
// Citations are file:line under /root/reference.
#ifndef VPK_EM_DEVICE_HPP_
#define VPK_EM_DEVICE_HPP_

#include "wave_prims.hpp"
#include "../../include/vpk.h"
#include "em_layout.hpp"

namespace vpk {

constexpr int MAXM = 64;            // capacity of simultaneously live VP hypotheses
constexpr int GRIDN = 20;           // CNN output grid (cnn/deploy.prototxt:283-296)
constexpr int NCELL = GRIDN * GRIDN;
constexpr int MAXCOMP = 100;        // prior keeps the 100 strongest cells (probability_functions.py:87)
constexpr int MT = 8;               // VP tile of the smoothing kernel (accumulators per column)
constexpr int PART_DOUBLES = 2048;  // LDS scratch of the setup phases (16 KiB): the head of the smoother's panel, not yet in use then
constexpr int WT_DOUBLES = 6144;    // LDS operand tile of the smoother (48 KiB)
constexpr int KNN1 = 10;            // line_rating_knn k1 (vp_localisation.py:34,230)
constexpr int TRACE_COLS = 12;       // trace row: M, max_err, M_end, events, us_estep, us_smooth, us_mstep, us_total,
                                     //            us_split_select, us_split_cluster, us_split_fit, us_merge
constexpr int KNN2 = 4;             // k2=4 at the call site (:230)
constexpr double PI_D = 3.141592653589793238462643383279502884;

struct Shared {
    double cur[MAXM * 3];   // v[i]   of the reference's history array
    double nxt[MAXM * 3];   // v[i+1]
    double s[MAXM];         // per-VP variance
    double pv[MAXM];        // prior p(v)
    double vx[MAXM], vy[MAXM];  // VP projected to the image plane
    double k2[MAXM];        // 1 / sqrt(2 pi s)
    double cnt[MAXM], cntw[MAXM], err[MAXM];
    int removed[MAXM];
    int icnt[MAXM];
    double red_v[32];
    int red_i[32];
    double pma[MAXCOMP], pmb[MAXCOMP], pw[MAXCOMP];  // prior mixture (alpha, beta, weight)
    float wts[NCELL];
    unsigned char mx[NCELL];
    int ncomp;
    int M;
    int status;
    unsigned flags;
    int ibuf[8];
    double dbuf[16];
    double sigma_prior;
    double active_us;       // device time spent on this image in earlier time slices
};

constexpr size_t SH_BYTES = (sizeof(Shared) + 15) / 16 * 16;
static_assert(sizeof(Shared) % 8 == 0 && sizeof(Shared) <= EM_STATE_DOUBLES * 8, "Shared must fit the slot's state region");
// LDS layout of every EM kernel: [Shared | smoother operand panel]
VPK_DEV Shared& SH() { return *reinterpret_cast<Shared*>(lds_base()); }
VPK_DEV double* WT() { return reinterpret_cast<double*>(lds_base() + SH_BYTES); }
VPK_DEV double* SCRATCH() { return WT(); }   // PART_DOUBLES doubles; every launch gives the panel at least that much

struct EmCtx {
    int N;
    int ldn;   // row stride of the [m][n] arrays (N rounded up to 8)
    int ld;    // row stride of lsim
    int mcap;  // row stride of wsrc ([n][m]); multiple of MT
    gdp l;
    cgdp lp;
    cgfp cnn;
    cgbp sphere;
    int ssize;
    cgdp init_vp;
    int n_init;
    vpk_em_params prm;
    // per-slot global scratch
    gdp lsim;     // N x ld
    gdp pdist;    // N x ld : closest distance of every pair of segments (setup scratch)
    gdp den;      // N   : 1 + bias * lweight[k] * sum_j lsim[j][k]
    gdp lweight;  // N
    gdp langle;   // N
    gdp lscore;   // N
    gdp lvsq;     // [m][n]
    gdp pvl;      // [m][n]
    gdp w;        // [m][n]
    gdp wsrc;     // [n][mcap] : p_vl * lweight, VP index contiguous (broadcast reads)
    gdp drow;     // 6 x ldn: per-line constants of the E-step (midpoint, direction, norm) and p_l
    gdp cl;       // split: Nw x Nw cluster distances (NULL when do_split == 0)
    gip assoc;    // N
    gip idx;      // 3N (split: gathered line indices, cluster membership)
    gdp rowsum;   // N : sum_j lsim[j][k]
    int wt_doubles;   // its capacity (WT_DOUBLES, or more when the launch gives the workgroup a whole CU)
    gdp part;     // global: nwaves x mcap x ldn row-slice partial sums of the smoother
    gdp lcopy;    // N x 3 normalised lines (l points here once the setup has run)
    gdp lpcopy;   // N x 4 segment end points (lp likewise)
    gdp state;    // snapshot of Shared while the image is suspended
    int smoother = 0; // 0: the row-sliced smoother wherever it applies; 1: always the round-1/2 kernels; 2: the sparse smoother
                      // where it applies (slower, see smooth_sparse), the row-sliced one elsewhere -- same bits under all three
};

// point the context's scratch pointers into one slot
VPK_DEV void bind_scratch(EmCtx& c, double* base_, const EmLayout& L, bool do_split) {
    gdp base = (gdp)base_;
    c.ldn = L.ldn; c.ld = L.ld; c.mcap = L.mcap;
    c.lsim = base + L.lsim; c.pdist = base + L.pdist; c.den = base + L.den; c.lweight = base + L.lweight;
    c.langle = base + L.langle; c.lscore = base + L.lscore; c.lvsq = base + L.lvsq;
    c.pvl = base + L.pvl; c.w = base + L.w; c.wsrc = base + L.wsrc; c.drow = base + L.drow;
    c.cl = do_split ? base + L.cl : (gdp) nullptr;
    c.rowsum = base + L.rowsum;
    c.part = base + L.part;
    c.lcopy = base + L.lcopy; c.lpcopy = base + L.lpcopy; c.state = base + L.state;
    c.assoc = (gip)(base + L.assoc);
    c.idx = (gip)(base + L.idx);
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
VPK_DEV double clip(double x, double lo, double hi) {  // np.clip (NaN passes through)
    return x < lo ? lo : (x > hi ? hi : x);
}
// phase stopwatch (thread 0, after a barrier): returns microseconds since the previous call
VPK_DEV double lap(long long& t) {
    long long now = clock_ticks();
    double us = (double)(now - t) * CLOCK_US;
    t = now;
    return us;
}
VPK_DEV bool is_nan(double x) { return x != x; }
// The reference's scalar code calls np.dot / np.linalg.norm on 2- and 3-vectors; NumPy's BLAS
// evaluates those as a fused chain  fma(x_{n-1}, y_{n-1}, ... fma(x1, y1, x0*y0))  (verified on the
// build container's NumPy 2.2.6 / OpenBLAS).  These helpers round the same way, which matters when a
// VP collapses onto a single line and 1 - |cos| is 0 or 1 ulp (sigma^2 at its 1e-200 floor).
VPK_DEV double dot2(double ax, double ay, double bx, double by) { return fma(ay, by, ax * bx); }
VPK_DEV double dot3(double ax, double ay, double az, double bx, double by, double bz) {
    return fma(az, bz, fma(ay, by, ax * bx));
}
VPK_DEV double norm2(double x, double y) { return sqrt(dot2(x, y, x, y)); }
VPK_DEV double norm3(double x, double y, double z) { return sqrt(dot3(x, y, z, x, y, z)); }
VPK_DEV double sign_np(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : (x == 0 ? 0.0 : x)); }

// workgroup-wide lexicographic (value, index) minimum; result to every thread
VPK_DEVFN void block_argmin(Shared&, double& v, int& idx) {
    Shared& sh = SH();
    wave_argmin(v, idx);
    if (lane() == 0) {
        sh.red_v[wave_id()] = v;
        sh.red_i[wave_id()] = idx;
    }
    block_sync();
    double bv = sh.red_v[0];
    int bi = sh.red_i[0];
    for (int k = 1; k < nwaves(); ++k) {
        double u = sh.red_v[k];
        int j = sh.red_i[k];
        bool take = (u < bv) || (u == bv && j < bi) || (bv != bv && u == u);
        bv = take ? u : bv;
        bi = take ? j : bi;
    }
    block_sync();
    v = bv;
    idx = bi;
}
VPK_DEVFN double block_max(Shared&, double v) {
    Shared& sh = SH();
    v = wave_max(v);
    if (lane() == 0) sh.red_v[wave_id()] = v;
    block_sync();
    double b = sh.red_v[0];
    for (int k = 1; k < nwaves(); ++k) b = nanmax(b, sh.red_v[k]);
    block_sync();
    return b;
}

// exp for arguments that are usually far below the underflow threshold (a VP against a distant mixture
// component or line): exp(x) is exactly 0 for x < -745.14 in glibc and in ocml, so the ~50-instruction
// evaluation is skipped there -- whole waves take the short path most of the time.
VPK_DEV double exp_underflow(double x) { return x < -746.0 ? 0.0 : exp(x); }

// symmetric 3x3 eigen-solver (cyclic Jacobi): A = J diag(ev) J^T, J orthogonal (columns = eigenvectors)
VPK_DEV void eig3_full(double a00, double a01, double a02, double a11, double a12, double a22,
                       double ev[3], double J[3][3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) J[i][k] = (i == k) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (!(off > 0)) break;
        for (int p = 0; p < 2; ++p) {
            for (int q = p + 1; q < 3; ++q) {
                double apq = A[p][q];
                if (apq == 0) continue;
                double g = 100.0 * fabs(apq);
                if (fabs(A[p][p]) + g == fabs(A[p][p]) && fabs(A[q][q]) + g == fabs(A[q][q])) {
                    A[p][q] = 0; A[q][p] = 0;                 // negligible against both diagonals
                    continue;
                }
                double theta = (A[q][q] - A[p][p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                if (!(fabs(theta) < 1e150)) t = 0.5 / theta;  // avoid overflow of theta^2
                double cth = 1 / sqrt(t * t + 1);
                double sth = t * cth;
                double app = A[p][p], aqq = A[q][q];
                A[p][p] = app - t * apq;
                A[q][q] = aqq + t * apq;
                A[p][q] = 0;
                A[q][p] = 0;
                int r = 3 - p - q;
                double arp = A[r][p], arq = A[r][q];
                A[r][p] = A[p][r] = cth * arp - sth * arq;
                A[r][q] = A[q][r] = sth * arp + cth * arq;
                for (int k = 0; k < 3; ++k) {
                    double vkp = J[k][p], vkq = J[k][q];
                    J[k][p] = cth * vkp - sth * vkq;
                    J[k][q] = sth * vkp + cth * vkq;
                }
            }
        }
    }
    ev[0] = A[0][0]; ev[1] = A[1][1]; ev[2] = A[2][2];
}

// Smallest right singular vector of the row-weighted line matrix diag(r) * L (N x 3), cooperatively
// by one aligned group of G lanes (G = WAVE: the whole wave) -- stands in for V[:,2] of
// numpy.linalg.svd (vp_localisation.py:466,595).  All lanes of the group must call it together.
// rw(n) returns the row weight r_n (0 = row not selected).  Pass 0 diagonalises the 3x3 scatter
// sum r^2 l l^T (normal equations: error ~ eps * cond^2 in the small direction); every further pass
// re-accumulates the scatter IN THE ROTATED BASIS V^T l, where the entries that couple to the small
// direction are sums of small numbers (no cancellation against the large ones), and applies the
// Jacobi correction -- an implicit one-sided Jacobi SVD, accurate like LAPACK's after 2-3 passes.
template <int G, int LB = 4, class RowWeight>
VPK_DEV void group_null_vector(cgdp l, int N, RowWeight rw, double out[3]) {
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double ev[3] = {0, 0, 0};
    for (int pass = 0; pass < 5; ++pass) {
        double g00 = 0, g01 = 0, g02 = 0, g11 = 0, g12 = 0, g22 = 0;
        // LB = four lines per step with all their loads issued first: with 16 lanes per VP a lane walks N/16
        // lines, and one L2 round trip per line was most of the M-step.  (Round 6 measured LB = 8 -- same chains, same
        // bits -- SLOWER: M-step 57.9 -> 60.3 ms of workgroup time per YUD batch; the walk is bound by its divisions
        // and the eigen-solve, not by loads in flight.)
        for (int n0 = lane() % G; n0 < N; n0 += LB * G) {
            double r[LB], a0[LB], a1[LB], a2[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int n = n0 + u * G;
                const bool in = n < N;
                r[u] = in ? rw(n) : 0.0;
                cgdp ln = l + 3 * (size_t)(in ? n : 0);
                a0[u] = ln[0]; a1[u] = ln[1]; a2[u] = ln[2];
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                if (r[u] == 0) continue;
                double y0, y1, y2;
                if (pass == 0) {
                    y0 = r[u] * a0[u]; y1 = r[u] * a1[u]; y2 = r[u] * a2[u];
                } else {
                    y0 = r[u] * (a0[u] * V[0][0] + a1[u] * V[1][0] + a2[u] * V[2][0]);
                    y1 = r[u] * (a0[u] * V[0][1] + a1[u] * V[1][1] + a2[u] * V[2][1]);
                    y2 = r[u] * (a0[u] * V[0][2] + a1[u] * V[1][2] + a2[u] * V[2][2]);
                }
                g00 += y0 * y0; g01 += y0 * y1; g02 += y0 * y2;
                g11 += y1 * y1; g12 += y1 * y2; g22 += y2 * y2;
            }
        }
        g00 = group_sum<G>(g00); g01 = group_sum<G>(g01); g02 = group_sum<G>(g02);
        g11 = group_sum<G>(g11); g12 = group_sum<G>(g12); g22 = group_sum<G>(g22);
        const double tol = 4e-16;
        const bool conv = pass > 0 && fabs(g01) <= tol * sqrt(g00 * g11) && fabs(g02) <= tol * sqrt(g00 * g22) &&
                          fabs(g12) <= tol * sqrt(g11 * g22);
        double J[3][3];
        eig3_full(g00, g01, g02, g11, g12, g22, ev, J);
        double Vn[3][3];
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) Vn[i][k] = V[i][0] * J[0][k] + V[i][1] * J[1][k] + V[i][2] * J[2][k];
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) V[i][k] = Vn[i][k];
        if (conv) break;
        if (pass == 0) {
            // normal-equations error of the bottom eigenvector ~ eps * ev_max / (ev_mid - ev_min):
            // below 1e-13 when the two larger eigenvalues are within 1e3 -> no refinement needed
            double lo = ev[0] < ev[1] ? ev[0] : ev[1]; lo = lo < ev[2] ? lo : ev[2];
            double hi = ev[0] > ev[1] ? ev[0] : ev[1]; hi = hi > ev[2] ? hi : ev[2];
            double mid = ev[0] + ev[1] + ev[2] - lo - hi;
            if (mid - lo > 1e-3 * hi) break;
        }
    }
    int b = 0;
    if (ev[1] < ev[b]) b = 1;
    if (ev[2] < ev[b]) b = 2;
    double x = V[0][b], y = V[1][b], z = V[2][b];
    double nrm = norm3(x, y, z);                              // vp /= np.linalg.norm(vp) (:472)
    out[0] = x / nrm; out[1] = y / nrm; out[2] = z / nrm;
}

template <class RowWeight>
VPK_DEV void wave_null_vector(cgdp l, int N, RowWeight rw, double out[3]) {
    group_null_vector<WAVE>(l, N, rw, out);
}

// Third right singular vector of a 1 x 3 matrix [a b c] as LAPACK returns it (numpy.linalg.svd with
// full_matrices on one row: dgesdd -> dgelqf -> one Householder reflector H = I - tau v v^T with
// beta = -sign(a)|x|, tau = (beta - a)/beta, v = (1, b/(a-beta), c/(a-beta)); V^T = H up to the sign
// of its first row).  The reference reaches this in the hard-assignment M-step when a VP wins a
// single line (vp_localisation.py:353-369) and its `err > 1.5` test (:387) depends on this vector.
VPK_DEV void lapack_null_1row(double a, double b, double c, double out[3]) {
    double nrm = sqrt(a * a + b * b + c * c);
    double beta = a >= 0 ? -nrm : nrm;
    if (a == 0 && 1.0 / a < 0) beta = nrm;                   // sign(-0.0)
    double tau = (beta - a) / beta;
    double v1 = b / (a - beta), v2 = c / (a - beta);
    out[0] = -tau * v2;
    out[1] = -tau * v2 * v1;
    out[2] = 1 - tau * v2 * v2;
}

// ---------------------------------------------------------------------------------------------
// segment geometry (vp_localisation.py:700-776)
// ---------------------------------------------------------------------------------------------
// vp_localisation.py:743-758: the reference squares the NORM of (b - a) (:747)
VPK_DEV double seg_point_dist(double ax, double ay, double bx, double by, double px, double py) {
    double dx = bx - ax, dy = by - ay;
    double nrm = norm2(dx, dy);
    double param = dot2(px - ax, py - ay, dx, dy) / (nrm * nrm);
    double cx, cy;
    if (param < 0) {
        cx = ax; cy = ay;
    } else if (param > 1) {
        cx = bx; cy = by;
    } else {
        cx = ax + param * dx; cy = ay + param * dy;
    }
    double ex = cx - px, ey = cy - py;
    return norm2(ex, ey);
}
// Per-line quantities reused by every pair this line takes part in (all as the reference rounds them)
struct LineGeom {
    double x1, y1, x2, y2;   // end points
    double dx, dy;           // (x2 - x1, y2 - y1): segment vector used by line_segment_point_distance
    double nn;               // np.square(norm(d)) (:747)
    double vx, vy;           // (x1 - x2, y1 - y2): direction used by lines_points_cosangle (:716)
    double nv;               // norm(v)
};
VPK_DEV LineGeom line_geom(const double a[4]) {
    LineGeom g;
    g.x1 = a[0]; g.y1 = a[1]; g.x2 = a[2]; g.y2 = a[3];
    g.dx = a[2] - a[0]; g.dy = a[3] - a[1];
    const double nrm = norm2(g.dx, g.dy);
    g.nn = nrm * nrm;
    g.vx = a[0] - a[2]; g.vy = a[1] - a[3];
    g.nv = norm2(g.vx, g.vy);
    return g;
}
// squared distance from point p to segment s (vp_localisation.py:743-758 before the final sqrt)
VPK_DEV double seg_point_dist_sq(const LineGeom& s, double px, double py) {
    const double param = dot2(px - s.x1, py - s.y1, s.dx, s.dy) / s.nn;
    double cx, cy;
    if (param < 0) {
        cx = s.x1; cy = s.y1;
    } else if (param > 1) {
        cx = s.x2; cy = s.y2;
    } else {
        cx = s.x1 + param * s.dx; cy = s.y1 + param * s.dy;
    }
    const double ex = cx - px, ey = cy - py;
    return dot2(ex, ey, ex, ey);
}
// vp_localisation.py:727-740.  sqrt is monotonic and correctly rounded, so min(sqrt(a..d)) ==
// sqrt(min(a..d)) bit for bit: one square root per pair instead of four.
VPK_DEV double line_distance_closest(const LineGeom& a, const LineGeom& b) {
    const double d1 = seg_point_dist_sq(a, b.x1, b.y1);
    const double d2 = seg_point_dist_sq(a, b.x2, b.y2);
    const double d4 = seg_point_dist_sq(b, a.x1, a.y1);
    const double d5 = seg_point_dist_sq(b, a.x2, a.y2);
    const double m = d1 < d2 ? d1 : d2;
    const double q = d4 < d5 ? d4 : d5;
    return sqrt(m < q ? m : q);
}
// cos(clip(9 * acos(c), -pi/2, pi/2)) for c in [0, 1] without acos/cos (vp_localisation.py:721-722 with
// f = 9): with s = sin(phi) = sqrt((1 - c)(1 + c)), cos(9 phi) = Re((c + i s)^9), evaluated by repeated
// squaring (unit-modulus products: ~1e-15 absolute error, the same order as libm's last-ulp noise through
// the ill-conditioned acos near c = 1).  9 phi >= pi/2  <=>  c <= cos(pi/18): the clipped branch returns
// numpy's cos(pi/2) = 6.123233995736766e-17.
VPK_DEV double cos9_of_cos(double c) {
    const double COS_PI_18 = 0.98480775301220802;     // cos(pi / 18)
    if (!(c > COS_PI_18)) return (c != c) ? c : 6.123233995736766e-17;
    if (c > 1.0) c = 1.0;                             // np.clip(cosdphi, -1, 1)
    const double s = sqrt((1.0 - c) * (1.0 + c));
    double re = c, im = s;                            // z
    double r2 = re * re - im * im, i2 = 2 * re * im;  // z^2
    double r4 = r2 * r2 - i2 * i2, i4 = 2 * r2 * i2;  // z^4
    double r8 = r4 * r4 - i4 * i4, i8 = 2 * r4 * i4;  // z^8
    return r8 * re - i8 * im;                         // Re(z^9)
}
VPK_DEV double lines_cosangle(const LineGeom& a, const LineGeom& b, double f) {   // :715-724, f = 9 only
    const double c = fabs(dot2(a.vx, a.vy, b.vx, b.vy) / (a.nv * b.nv));
    (void)f;
    return cos9_of_cos(c);
}
// vp_localisation.py:727-740
VPK_DEV double line_distance_closest(const double a[4], const double b[4]) {
    double d1 = seg_point_dist(a[0], a[1], a[2], a[3], b[0], b[1]);
    double d2 = seg_point_dist(a[0], a[1], a[2], a[3], b[2], b[3]);
    double d4 = seg_point_dist(b[0], b[1], b[2], b[3], a[0], a[1]);
    double d5 = seg_point_dist(b[0], b[1], b[2], b[3], a[2], a[3]);
    double m = d1 < d2 ? d1 : d2;           // np.min of [d1,d2,d4,d5]; NaN handling not replicated
    double q = d4 < d5 ? d4 : d5;
    return m < q ? m : q;
}
// vp_localisation.py:715-724
VPK_DEV double lines_cosangle(const double a[4], const double b[4], double f) {
    double v1x = a[0] - a[2], v1y = a[1] - a[3];
    double v2x = b[0] - b[2], v2y = b[1] - b[3];
    double n1 = norm2(v1x, v1y), n2 = norm2(v2x, v2y);
    double c = fabs(dot2(v1x, v1y, v2x, v2y) / (n1 * n2));
    double dphi = fabs(acos(clip(c, -1.0, 1.0)));
    return cos(clip(f * dphi, -PI_D / 2, PI_D / 2));
}
VPK_DEV double line_length(const double a[4]) {
    return norm2(a[0] - a[2], a[1] - a[3]);
}
// vp_localisation.py:708-712 with the distance supplied
VPK_DEV double proximity(double d, double len_a, double len_b, double sigma) {
    double sg = sigma * (len_a < len_b ? len_a : len_b);
    return exp(-(d * d) / (2 * sg * sg));
}

// ---------------------------------------------------------------------------------------------
// setup: line normalisation, pairwise similarity + kNN score, weights
// ---------------------------------------------------------------------------------------------
// vp_localisation.py:185-186 and again :226 (the second pass divides by ~1)
VPK_DEVFN void normalise_lines(EmCtx& c) {
    for (int n = tid(); n < c.N; n += nthreads()) {
        gdp r = c.l + 3 * (size_t)n;
        for (int pass = 0; pass < 2; ++pass) {
            double nr = norm3(r[0], r[1], r[2]);
            r[0] /= nr; r[1] /= nr; r[2] /= nr;
        }
    }
    block_sync();
}

// calc_lsim (vp_localisation.py:87-108, sigma=1 at :178) and line_rating_knn (:34-84, k2=4 at
// :230) from ONE evaluation of every pair: the closest distance is shared by both.  Also lines_angles (:765-776).
//
// With weights (want_lsim) every UNORDERED pair is evaluated once, like the reference does (:102-108 compute
// lines_similarity(lp[i], lp[j]) for j < i and :95-97 mirror it): pass 1, one wave per row i, lanes over the
// columns j < i, writes the similarity and the distance to (i, j) and (j, i); pass 2, one wave per row, sums the
// row in the order the one-pass version did (so rowsum keeps its bits) and rates the line from its distance row.
// The pair functions are symmetric bit for bit (tests/test_gpu_em.py asserts lsim == lsim.T), so nothing moves.
// Without weights only the distances matter and the one-pass version below runs (every ordered pair, no matrix).
//
// LARGE images (N >= PW_TILED_MIN; round 6): pass 1 walks TILES of 16 rows x 64 columns instead of whole rows.  Row by row, the mirrored
// store (j, i) of a row's pairs touches N different 128-byte lines with 8 bytes each, and the other 15 entries of such a line arrive with
// the next 15 rows -- other waves, tens of microseconds later.  At N = 1000 that is 128 KB of partially written lines in flight per
// row, 256 workgroups share 32 MB of L2, the lines are evicted half-written and written again: profiles/r05_pmc_traffic.json counts
// 25.8 GB of HBM writes per 512-image stress launch against 7.6 GB algorithmic (DESIGN r5 blamed the smoother's partials,
// which N = 1000 does not have).  In a tile, one wave writes all 16 entries of a line within 16 consecutive pairs, the line is complete
// before it can be evicted, and the geometry of column j's line is computed once per 16 pairs instead of once per pair.  Same pair
// function, same argument order, same positions: every byte of lsim / pdist is the row-by-row version's (tests/test_gpu_em.py).
// The mirrored half of an interior tile goes through a wave-private LDS transpose: lane L then stores 16 bytes of row j = 8 q + L / 8 at
// columns i0 + 2 (L % 8) -- eight rows x 128 contiguous bytes per store instruction, whole lines like the direct half.  (Measured with the
// mirrored entries as 8-byte stores, 64 rows per instruction: the set-up of 256 stress images wrote 11.1 GB where the matrices are
// 4.1 GB -- profiles/r06_pmc_em_traffic.txt -- the L2 writes such sectors out more than once.)
constexpr int PW_TILED_MIN = 512;
constexpr int PW_RB = 16;                                      // rows of a tile
constexpr int PW_TLD = WAVE + 1;                               // row stride of the transpose buffers (doubles)
constexpr int PW_TBUF = 2 * PW_RB * PW_TLD;                    // per wave: similarity and distance tiles
VPK_DEV void pairwise_tiles(EmCtx& c) {
    const int N = c.N;
    double* gs = SCRATCH() + wave_id() * (PW_RB * 10);          // this wave's 16 row geometries (LineGeom = 10 doubles)
    static_assert(8 * PW_RB * 10 <= PART_DOUBLES, "row geometries of eight waves in the setup scratch");
    // the transpose buffers lie behind the setup scratch, if the launch's LDS has room for them (the whole-CU configuration has)
    const bool tbuf_ok = WAVE == 64 && PART_DOUBLES + nwaves() * PW_TBUF <= c.wt_doubles;
    double* tb = SCRATCH() + PART_DOUBLES + wave_id() * PW_TBUF;
    const int nb = (N + PW_RB - 1) / PW_RB;
    int t = 0;
    for (int I = 0; I < nb; ++I) {
        const int i0 = I * PW_RB;
        const int ilast = (i0 + PW_RB < N ? i0 + PW_RB : N) - 1;    // the block's last row; its pairs are the columns j < ilast
        const int nch = (ilast + WAVE - 1) / WAVE;
        for (int jc = 0; jc < nch; ++jc, ++t) {
            if (t % nwaves() != wave_id()) continue;            // tiles are dealt to the waves in turn
            if (lane() < PW_RB) {
                const int i = i0 + lane() < N ? i0 + lane() : N - 1;
                double a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = c.lp[4 * (size_t)i + q];
                const LineGeom g = line_geom(a);
                double* o = gs + lane() * 10;
                o[0] = g.x1; o[1] = g.y1; o[2] = g.x2; o[3] = g.y2; o[4] = g.dx; o[5] = g.dy; o[6] = g.nn; o[7] = g.vx; o[8] = g.vy; o[9] = g.nv;
            }
            wave_lds_order();
            const int j = jc * WAVE + lane();
            const int jj = j < N ? j : N - 1;
            double b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = c.lp[4 * (size_t)jj + q];
            const LineGeom gb = line_geom(b);
            auto row_geom = [&](int r) {
                const double* o = gs + r * 10;
                LineGeom g;
                g.x1 = o[0]; g.y1 = o[1]; g.x2 = o[2]; g.y2 = o[3]; g.dx = o[4]; g.dy = o[5]; g.nn = o[6]; g.vx = o[7]; g.vy = o[8]; g.nv = o[9];
                return g;
            };
            auto put = [&](int i, double d, double sim) {      // pair (i, j), j < i: both halves of the symmetric matrices
                if (!(j < i && i < N)) return;
                c.lsim[(size_t)i * c.ld + j] = sim;
                c.lsim[(size_t)j * c.ld + i] = sim;
                c.pdist[(size_t)i * c.ld + j] = d;
                c.pdist[(size_t)j * c.ld + i] = d;
            };
            // interior tile: every column of the chunk lies in front of the block's first row and every row exists
            const bool interior = tbuf_ok && jc * WAVE + WAVE - 1 < i0 && i0 + PW_RB <= N;
            for (int r = 0; r < PW_RB; r += 2) {                // two independent pairs per trip (see the row-by-row loop)
                const LineGeom g0 = row_geom(r), g1 = row_geom(r + 1);
                const double d0 = line_distance_closest(g0, gb);
                const double d1 = line_distance_closest(g1, gb);
                const double s0 = lines_cosangle(g0, gb, 9.0) * proximity(d0, g0.nv, gb.nv, 1.0);
                const double s1 = lines_cosangle(g1, gb, 9.0) * proximity(d1, g1.nv, gb.nv, 1.0);
                if (interior) {
                    c.lsim[(size_t)(i0 + r) * c.ld + j] = s0;   c.pdist[(size_t)(i0 + r) * c.ld + j] = d0;
                    c.lsim[(size_t)(i0 + r + 1) * c.ld + j] = s1; c.pdist[(size_t)(i0 + r + 1) * c.ld + j] = d1;
                    tb[r * PW_TLD + lane()] = s0;               tb[(PW_RB + r) * PW_TLD + lane()] = d0;
                    tb[(r + 1) * PW_TLD + lane()] = s1;         tb[(PW_RB + r + 1) * PW_TLD + lane()] = d1;
                } else {
                    put(i0 + r, d0, s0);
                    put(i0 + r + 1, d1, s1);
                }
            }
            wave_lds_order();                                   // (the next tile overwrites the row geometries)
            if (interior) {
                const int cp = lane() & 7, jr = lane() >> 3;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int jl = 8 * q + jr;                  // column of the tile = row of the mirrored entries
                    const size_t at = (size_t)(jc * WAVE + jl) * c.ld + i0 + 2 * cp;
                    store_cols2(c.lsim + at, tb[(2 * cp) * PW_TLD + jl], tb[(2 * cp + 1) * PW_TLD + jl]);
                    store_cols2(c.pdist + at, tb[(PW_RB + 2 * cp) * PW_TLD + jl], tb[(PW_RB + 2 * cp + 1) * PW_TLD + jl]);
                }
                wave_lds_order();
            }
        }
    }
    for (int i = tid(); i < N; i += nthreads()) {
        c.lsim[(size_t)i * c.ld + i] = 0.0;                     // :104 (the row's own entry stays 0)
        c.pdist[(size_t)i * c.ld + i] = 4.0;                    // :82
    }
}

VPK_DEVFN void pairwise_setup(EmCtx& c, bool want_lsim) {
    const int N = c.N;
    if (want_lsim) {
        // Row i has i pairs and row N - 1 - i has N - 1 - i: a wave takes the two together, N - 1 pairs for every
        // wave (whole trips of 2 x 64 pairs; row by row the short rows leave most lanes idle at N ~ 100..400).
        if (N >= PW_TILED_MIN && WAVE == 64 && c.smoother != 1) pairwise_tiles(c);
        else
        for (int r = wave_id(); 2 * r < N; r += nwaves()) {
            const int i0 = r, i1 = N - 1 - r;                      // i0 <= i1
            const int len = i0 == i1 ? i0 : i0 + i1;
            double a0[4], a1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a0[q] = c.lp[4 * (size_t)i0 + q]; a1[q] = c.lp[4 * (size_t)i1 + q]; }
            const LineGeom g0 = line_geom(a0), g1 = line_geom(a1);
            // two independent pairs per lane per trip: the per-pair arithmetic is one long dependent fp64 chain
            // (divisions, square roots, exp), so the second pair fills its issue bubbles
            auto one = [&](int t, double& d, double& sim) {
                const bool first = t < i0;
                const int j = t < len ? (first ? t : t - i0) : 0;
                const LineGeom& ga = first ? g0 : g1;              // pair (i, j): a = line i, b = line j < i (:102-106)
                double b[4] = {c.lp[4 * (size_t)j], c.lp[4 * (size_t)j + 1], c.lp[4 * (size_t)j + 2], c.lp[4 * (size_t)j + 3]};
                const LineGeom gb = line_geom(b);
                d = line_distance_closest(ga, gb);
                sim = lines_cosangle(ga, gb, 9.0) * proximity(d, ga.nv, gb.nv, 1.0);
            };
            auto put = [&](int t, double d, double sim) {
                if (t >= len) return;
                const int i = t < i0 ? i0 : i1, j = t < i0 ? t : t - i0;
                c.lsim[(size_t)i * c.ld + j] = sim;
                c.lsim[(size_t)j * c.ld + i] = sim;
                c.pdist[(size_t)i * c.ld + j] = d;
                c.pdist[(size_t)j * c.ld + i] = d;
            };
            for (int t = lane(); t < len; t += 2 * WAVE) {
                double d0, s0, d1, s1;
                one(t, d0, s0);
                one(t + WAVE, d1, s1);
                put(t, d0, s0);
                put(t + WAVE, d1, s1);
            }
            if (lane() == 0) {
                c.lsim[(size_t)i0 * c.ld + i0] = 0.0;              // :104 (the row's own entry stays 0)
                c.pdist[(size_t)i0 * c.ld + i0] = 4.0;             // :82
                c.lsim[(size_t)i1 * c.ld + i1] = 0.0;
                c.pdist[(size_t)i1 * c.ld + i1] = 4.0;
            }
        }
        block_sync();
        // pass 2a: row sums, one wave per row, each lane adds its columns in ascending order and the wave reduces --
        // the order the one-pass version used (lsim is symmetric: row sum == column sum, :522)
        for (int i = wave_id(); i < N; i += nwaves()) {
            cgdp srow = c.lsim + (size_t)i * c.ld;
            double rsum = 0.0;
            for (int j = lane(); j < N; j += WAVE) rsum += srow[j];
            rsum = wave_sum(rsum);
            if (lane() == 0) c.rowsum[i] = rsum;
        }
        // pass 2b: line_rating_knn from the stored distance rows, ROWG lanes per row and WAVE / ROWG rows per wave at a
        // time: the k1 selection rounds (a lexicographic minimum over the group and a pop) and the serial tail are
        // per-row costs that a whole wave per row paid ~14 us for; a 16-lane minimum is four DPP steps.
        constexpr int RPW = WAVE / ROWG;
        const int grp = lane() / ROWG, gl = lane() % ROWG;
        const int k1 = N < KNN1 ? N : KNN1;
        const int k2 = N < KNN2 ? N : KNN2;
        double* ks = SCRATCH() + (wave_id() * RPW + grp) * (4 * KNN1 + KNN2);   // idx, dist, cos, prox per neighbour + term by rank
        for (int base = wave_id() * RPW; base < N; base += nwaves() * RPW) {
            const int i = base + grp;
            const bool valid = i < N;
            const int ii = valid ? i : 0;
            double td[KNN1];
            int tj[KNN1];
#pragma unroll
            for (int q = 0; q < KNN1; ++q) { td[q] = 1e300; tj[q] = 0x7fffffff; }
            cgdp drow = c.pdist + (size_t)ii * c.ld;
            for (int j0 = gl; j0 < N; j0 += 4 * ROWG) {            // four loads in flight
                double dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) dv[u] = (j0 + u * ROWG < N) ? drow[j0 + u * ROWG] : 1e300;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double nd = dv[u];
                    int nj = (j0 + u * ROWG < N) ? j0 + u * ROWG : 0x7fffffff;
#pragma unroll
                    for (int q = 0; q < KNN1; ++q) {
                        const bool lt = (nd < td[q]) || (nd == td[q] && nj < tj[q]);
                        const double od = td[q];
                        const int oj = tj[q];
                        td[q] = lt ? nd : od;
                        tj[q] = lt ? nj : oj;
                        nd = lt ? od : nd;
                        nj = lt ? oj : nj;
                    }
                }
            }
            for (int r = 0; r < k1; ++r) {                          // k1 nearest overall, by (distance, index)
                double bd = td[0];
                int bj = tj[0];
                row16_argmin(bd, bj);
                if (tj[0] == bj && td[0] == bd) {                   // the winning lane pops its head
#pragma unroll
                    for (int q = 0; q + 1 < KNN1; ++q) { td[q] = td[q + 1]; tj[q] = tj[q + 1]; }
                    td[KNN1 - 1] = 1e300;
                    tj[KNN1 - 1] = 0x7fffffff;
                }
                if (gl == 0) { ks[r] = (double)bj; ks[KNN1 + r] = bd; }
            }
            wave_sync();
            double a[4] = {c.lp[4 * (size_t)ii], c.lp[4 * (size_t)ii + 1], c.lp[4 * (size_t)ii + 2], c.lp[4 * (size_t)ii + 3]};
            const double len_a = norm2(a[0] - a[2], a[1] - a[3]);
            for (int q = gl; q < k1; q += ROWG) {
                int j = (int)ks[q];
                j = (valid && j >= 0 && j < N) ? j : 0;
                double b[4] = {c.lp[4 * (size_t)j], c.lp[4 * (size_t)j + 1], c.lp[4 * (size_t)j + 2], c.lp[4 * (size_t)j + 3]};
                ks[2 * KNN1 + q] = lines_cosangle(a, b, 9.0);                          // :55
                ks[3 * KNN1 + q] = proximity(ks[KNN1 + q], len_a, line_length(b), 1.0);  // :65
            }
            wave_sync();
            // np.argsort(cosphi)[::-1][0:k2] (:57-59): descending, ties -> later position first
            for (int q = gl; q < k1; q += ROWG) {
                const double cq = ks[2 * KNN1 + q];
                int rank = 0;
                for (int p = 0; p < k1; ++p) {
                    const double cp = ks[2 * KNN1 + p];
                    rank += (cp > cq) || (cp == cq && p > q);
                }
                if (rank < k2) ks[4 * KNN1 + rank] = ks[3 * KNN1 + q] * cq;                 // :66
            }
            wave_sync();
            if (gl == 0 && valid) {
                double sum = 0.0;
                for (int r = 0; r < k2; ++r) sum += ks[4 * KNN1 + r];                       // :68, in rank order
                c.lscore[i] = sum / k2;                                                 // :70
                double vx = a[0] - a[2], vy = a[1] - a[3];                              // lines_angles (:765-776)
                double nr = norm2(vx, vy);
                double phi = fabs(acos(clip(vx / nr, -1.0, 1.0)));
                c.langle[i] = phi > PI_D / 2 ? PI_D - phi : phi;
            }
            wave_sync();
        }
        block_sync();
        return;
    }
    // per-wave kNN scratch carved from the partial-sum buffer: [k1] idx(as double), dist, cos, prox
    double* ks = SCRATCH() + wave_id() * (4 * KNN1 + KNN2);   // idx, dist, cos, prox per neighbour + term by rank
    const int k1 = N < KNN1 ? N : KNN1;
    const int k2 = N < KNN2 ? N : KNN2;
    for (int i = wave_id(); i < N; i += nwaves()) {
        double a[4] = {c.lp[4 * (size_t)i], c.lp[4 * (size_t)i + 1], c.lp[4 * (size_t)i + 2],
                       c.lp[4 * (size_t)i + 3]};
        const LineGeom ga = line_geom(a);
        const double len_a = ga.nv;               // line_length == norm of the direction (:761-762)
        double rsum = 0.0;
        // per-lane sorted list of this lane's KNN1 nearest columns, by (distance, index): filled by a
        // branch-free insertion during the pair loop, merged across the wave afterwards -- the distance
        // row never goes to memory
        double td[KNN1];
        int tj[KNN1];
#pragma unroll
        for (int q = 0; q < KNN1; ++q) { td[q] = 1e300; tj[q] = 0x7fffffff; }
        // two independent pairs per lane per trip: the per-pair arithmetic is one long dependent fp64
        // chain (divisions, square roots, exp), so the second pair fills its issue bubbles
        auto pair_eval = [&](int j, double& d, double& sim) {
            const int jj = j < N ? j : i;          // clamp: lanes past the end recompute the diagonal
            double b[4] = {c.lp[4 * (size_t)jj], c.lp[4 * (size_t)jj + 1], c.lp[4 * (size_t)jj + 2],
                           c.lp[4 * (size_t)jj + 3]};
            const LineGeom gb = line_geom(b);
            d = line_distance_closest(ga, gb);
            sim = want_lsim ? lines_cosangle(ga, gb, 9.0) * proximity(d, len_a, gb.nv, 1.0) : 0.0;
        };
        auto pair_commit = [&](int j, double d, double sim) {
            if (j >= N) return;
            if (want_lsim) {
                sim = (i == j) ? 0.0 : sim;
                c.lsim[(size_t)i * c.ld + j] = sim;
                rsum += sim;
            }
            double nd = (i == j) ? 4.0 : d;       // :82
            int nj = j;
#pragma unroll
            for (int q = 0; q < KNN1; ++q) {
                const bool lt = (nd < td[q]) || (nd == td[q] && nj < tj[q]);
                const double od = td[q];
                const int oj = tj[q];
                td[q] = lt ? nd : od;
                tj[q] = lt ? nj : oj;
                nd = lt ? od : nd;
                nj = lt ? oj : nj;
            }
        };
        for (int j = lane(); j < N; j += 2 * WAVE) {
            double d0, s0, d1, s1;
            pair_eval(j, d0, s0);
            pair_eval(j + WAVE, d1, s1);
            pair_commit(j, d0, s0);
            pair_commit(j + WAVE, d1, s1);
        }
        rsum = wave_sum(rsum);                    // lsim is symmetric: row sum == column sum (:522)
        if (lane() == 0) c.rowsum[i] = rsum;
        // k1 nearest overall: k1 rounds of lexicographic (distance, index) minimum over the lane heads
        for (int r = 0; r < k1; ++r) {
            double bd = td[0];
            int bj = tj[0];
            wave_argmin(bd, bj);
            if (tj[0] == bj && td[0] == bd) {     // the winning lane pops its head
#pragma unroll
                for (int q = 0; q + 1 < KNN1; ++q) { td[q] = td[q + 1]; tj[q] = tj[q + 1]; }
                td[KNN1 - 1] = 1e300;
                tj[KNN1 - 1] = 0x7fffffff;
            }
            if (lane() == 0) { ks[r] = (double)bj; ks[KNN1 + r] = bd; }
        }
        wave_sync();
        for (int q = lane(); q < k1; q += WAVE) {
            int j = (int)ks[q];
            double b[4] = {c.lp[4 * (size_t)j], c.lp[4 * (size_t)j + 1], c.lp[4 * (size_t)j + 2],
                           c.lp[4 * (size_t)j + 3]};
            ks[2 * KNN1 + q] = lines_cosangle(a, b, 9.0);                          // :55
            ks[3 * KNN1 + q] = proximity(ks[KNN1 + q], len_a, line_length(b), 1.0);  // :65
        }
        wave_sync();
        // np.argsort(cosphi)[::-1][0:k2] (:57-59): descending, ties -> later position first.  Each of the
        // k1 neighbour lanes computes its own rank; the k2 best publish prox * cos under their rank.
        for (int q = lane(); q < k1; q += WAVE) {
            const double cq = ks[2 * KNN1 + q];
            int rank = 0;
            for (int p = 0; p < k1; ++p) {
                const double cp = ks[2 * KNN1 + p];
                rank += (cp > cq) || (cp == cq && p > q);
            }
            if (rank < k2) ks[4 * KNN1 + rank] = ks[3 * KNN1 + q] * cq;                 // :66
        }
        wave_sync();
        if (lane() == 0) {
            double sum = 0.0;
            for (int r = 0; r < k2; ++r) sum += ks[4 * KNN1 + r];                       // :68, in rank order
            c.lscore[i] = sum / k2;                                                 // :70
            // lines_angles (:765-776)
            double vx = a[0] - a[2], vy = a[1] - a[3];
            double nr = norm2(vx, vy);
            double phi = fabs(acos(clip(vx / nr, -1.0, 1.0)));
            c.langle[i] = phi > PI_D / 2 ? PI_D - phi : phi;
        }
        wave_sync();
    }
    block_sync();
}

// lweight = len * clip(lscore, 0.2, 1) (vp_localisation.py:227-233) and the hoisted denominator
// of weight_matrix (:522): den[k] = 1 + bias * lweight[k] * sum_j lsim[j][k]
VPK_DEVFN void weights_setup(EmCtx& c) {
    const int N = c.N;
    const bool uw = c.prm.use_weights != 0;
    for (int n = tid(); n < N; n += nthreads()) {
        double a[4] = {c.lp[4 * (size_t)n], c.lp[4 * (size_t)n + 1], c.lp[4 * (size_t)n + 2],
                       c.lp[4 * (size_t)n + 3]};
        c.lweight[n] = uw ? line_length(a) * clip(c.lscore[n], 0.2, 1.0) : 1.0;
    }
    block_sync();
    for (int k = tid(); k < N; k += nthreads()) {
        c.den[k] = 1 + c.prm.wbias * c.lweight[k] * (uw ? c.rowsum[k] : 0.0);
        // a NaN / Inf in lsim (a segment of length 0) shows in its row sum: the sparse smoother, which leaves out the
        // terms 0 * lsim, then stands back (0 * Inf is NaN, not 0)
        if (uw && !(fabs(c.rowsum[k]) <= 1.7976931348623157e308)) SH().ibuf[2] = 1;
    }
    block_sync();
}

// ---------------------------------------------------------------------------------------------
// prior parameters and initial VPs from the CNN grid
// ---------------------------------------------------------------------------------------------
// numpy's float32 pairwise summation (np.sum over a contiguous float32 array), needed because
// the prior weights are normalised in float32 (probability_functions.py:82-90)
VPK_DEV float np_pairwise_block_f32(const float* a, int n) {   // 8 <= n <= 128
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}
// n = 400 splits as (96 + 104) + (96 + 104): n2 = n/2 rounded down to a multiple of 8 at each level
VPK_DEV float np_pairwise_sum_f32_400(const float* a) {
    float lo = np_pairwise_block_f32(a, 96) + np_pairwise_block_f32(a + 96, 104);
    float hi = np_pairwise_block_f32(a + 200, 96) + np_pairwise_block_f32(a + 296, 104);
    return lo + hi;
}

// np.linspace(-(A-1)/A*pi/2, (A-1)/A*pi/2, A)[i] (probability_functions.py:73,75)
VPK_DEV double grid_centre(int i) {
    double start = -(GRIDN - 1.0) / GRIDN * PI_D / 2;
    double stop = (GRIDN - 1.0) / GRIDN * PI_D / 2;
    double step = (stop - start) / (GRIDN - 1);
    return i == GRIDN - 1 ? stop : i * step + start;
}

// pdf_params (probability_functions.py:62-96): keep the 100 strongest cells, normalise in f32.
VPK_DEVFN void prior_setup(EmCtx& c) {
    Shared& sh = SH();
    for (int i = tid(); i < NCELL; i += nthreads()) sh.wts[i] = c.cnn[i];
    block_sync();
    float* keep = (float*)SCRATCH();  // 400 floats
    for (int i = tid(); i < NCELL; i += nthreads()) {
        float wi = sh.wts[i];
        int rank = 0;  // position in argsort(weights)[::-1]: ties -> higher index first
        for (int j = 0; j < NCELL; ++j) {
            float wj = sh.wts[j];
            rank += (wj > wi) || (wj == wi && j > i);
        }
        keep[i] = rank < MAXCOMP ? wi : 0.f;
    }
    block_sync();
    if (tid() == 0) {
        sh.sigma_prior = PI_D / (1.282 * GRIDN);  // :71
        ((float*)SCRATCH())[NCELL] = np_pairwise_sum_f32_400(keep);
    }
    block_sync();
    // normalise every cell in parallel, then list the positive ones in cell-index order (calc_pdf visits cells in
    // index order, :20-21): position = positive cells in earlier waves + positive cells in lower lanes.  (One thread
    // walking the 400 cells with two f32 divisions each was ~50 us per image.)
    {
        const float sum = ((float*)SCRATCH())[NCELL];
        const float dv = (float)(2 * PI_D * sh.sigma_prior * sh.sigma_prior);
        int* wcount = (int*)(SCRATCH() + 256);            // per-wave counts (behind the 401 floats)
        for (int base = 0; base < NCELL; base += nthreads()) {   // one round for 512 threads
            const int i = base + tid();
            float w = 0.f;
            if (i < NCELL) {
                w = keep[i] / sum;
                w = w / dv;
            }
            const unsigned long long pos = wave_ballot(i < NCELL && w > 0);
            if (lane() == 0) wcount[wave_id()] = popcount64(pos);
            block_sync();
            int before = (base == 0) ? 0 : sh.ncomp;
            for (int k = 0; k < wave_id(); ++k) before += wcount[k];
            const int nc = before + popcount64(pos & lanes_below());
            if (i < NCELL) {
                sh.wts[i] = w;
                if (w > 0 && nc < MAXCOMP) {
                    sh.pma[nc] = grid_centre(i % GRIDN);   // means[:,0] = alpha, varies along columns
                    sh.pmb[nc] = grid_centre(i / GRIDN);   // means[:,1] = beta, varies along rows
                    sh.pw[nc] = (double)w;
                }
            }
            block_sync();
            if (tid() == 0) {
                int tot = before;                              // thread 0: wave 0, lane 0 -> before = carried count
                for (int k = 0; k < nwaves(); ++k) tot += wcount[k];
                sh.ncomp = tot < MAXCOMP ? tot : MAXCOMP;
            }
            block_sync();
        }
    }
}

// find_maxima (vp_localisation.py:13-31) + find_initial_vps (:111-165).  Leaves the VPs in
// sh.cur (row-major cell order), sh.M = count.  Uses SCRATCH().
VPK_DEVFN void initial_vps(EmCtx& c) {
    Shared& sh = SH();
    cgfp r = c.cnn;
    for (int i = tid(); i < NCELL; i += nthreads()) {
        int b = i / GRIDN, a = i % GRIDN;
        float vm = r[i];
        float vu = (a + 1 < GRIDN) ? r[b * GRIDN + a + 1] : 0.f;
        float vd = (a - 1 > 0) ? r[b * GRIDN + a - 1] : 0.f;   // quirk: index 0 never a neighbour
        float vl = (b - 1 > 0) ? r[(b - 1) * GRIDN + a] : 0.f;
        float vr = (b + 1 < GRIDN) ? r[(b + 1) * GRIDN + a] : 0.f;
        sh.mx[i] = (vm > vu && vm > vd && vm > vl && vm > vr) ? 1 : 0;
    }
    block_sync();
    unsigned char* keep = (unsigned char*)SCRATCH();        // 400 bytes
    double* cand = SCRATCH() + 64;                          // 400 x 4 doubles (x,y,z,valid)
    const int num_max = c.prm.num_init_vp;
    for (int i = tid(); i < NCELL; i += nthreads()) {
        int k = 0;
        if (sh.mx[i]) {
            int rank = 0;  // argsort(resp[maxima])[::-1]: ties -> later maximum first (:123-125)
            float vi = r[i];
            for (int j = 0; j < NCELL; ++j)
                if (sh.mx[j]) rank += (r[j] > vi) || (r[j] == vi && j > i);
            k = rank < num_max;
        }
        keep[i] = (unsigned char)k;
        cand[4 * i + 3] = 0.0;
    }
    block_sync();
    const int S = c.ssize;
    for (int cell = wave_id(); cell < NCELL; cell += nwaves()) {
        if (!keep[cell]) continue;
        int ra = cell / GRIDN, rb = cell % GRIDN;
        int r0 = ra * S / GRIDN, r1 = (ra + 1) * S / GRIDN;   // rows of the FLIPPED image (:114,:133)
        int c0 = rb * S / GRIDN, c1 = (rb + 1) * S / GRIDN;
        int bw = c1 - c0, npix = (r1 - r0) * bw;
        // the slice's pixels are loaded ONCE, ten per lane with all loads in flight (25 x 25 pixels at S = 500); the
        // maximum and the positions that reach it come out of registers.  (Two passes of dependent byte loads were
        // ~20 memory round trips per cell.)
        constexpr int PV = 10;
        int mxv = 0, cntp = 0, sr = 0, sc = 0;
        if (npix <= PV * WAVE) {
            int vals[PV];
#pragma unroll
            for (int q = 0; q < PV; ++q) {
                const int p = lane() + q * WAVE;
                const int pc = p < npix ? p : 0;
                const int v = c.sphere[(size_t)(S - 1 - (r0 + pc / bw)) * S + c0 + pc % bw];
                vals[q] = p < npix ? v : -1;
            }
#pragma unroll
            for (int q = 0; q < PV; ++q) mxv = vals[q] > mxv ? vals[q] : mxv;
            mxv = wave_max_int(mxv);
            if (mxv == 0) continue;                           // :137-142
#pragma unroll
            for (int q = 0; q < PV; ++q) {
                const int p = lane() + q * WAVE;
                if (vals[q] == mxv) { ++cntp; sr += p / bw; sc += p % bw; }
            }
        } else {
            for (int p = lane(); p < npix; p += WAVE) {
                int rr = r0 + p / bw, cc = c0 + p % bw;
                int v = c.sphere[(size_t)(S - 1 - rr) * S + cc];
                mxv = v > mxv ? v : mxv;
            }
            mxv = wave_max_int(mxv);
            if (mxv == 0) continue;                           // :137-142
            for (int p = lane(); p < npix; p += WAVE) {
                int rr = p / bw, cc = p % bw;
                int v = c.sphere[(size_t)(S - 1 - (r0 + rr)) * S + c0 + cc];
                if (v == mxv) { ++cntp; sr += rr; sc += cc; }
            }
        }
        cntp = wave_sum_int(cntp);
        sr = wave_sum_int(sr);
        sc = wave_sum_int(sc);
        if (lane() == 0) {
            double avg_r = (double)sr / cntp, avg_c = (double)sc / cntp;   // :148-151
            double ia = avg_c + c0, ib = avg_r + r0;                       // :155-158 (col,row)
            double alpha = (ia - 0.5 * S + 0.5) * PI_D / S;                // coordinate_conversion.py:14-15
            double beta = (ib - 0.5 * S + 0.5) * PI_D / S;
            double px = sin(alpha) * cos(beta), py = sin(beta), pz = cos(alpha) * cos(beta);
            double sg = sign_np(pz);                                       // :48
            cand[4 * cell + 0] = px * sg;
            cand[4 * cell + 1] = py * sg;
            cand[4 * cell + 2] = pz * sg;
            cand[4 * cell + 3] = 1.0;
        }
    }
    block_sync();
    if (tid() == 0) {
        int m = 0;
        for (int cell = 0; cell < NCELL; ++cell) {
            if (cand[4 * cell + 3] != 0.0 && m < MAXM) {
                sh.cur[3 * m + 0] = cand[4 * cell + 0];
                sh.cur[3 * m + 1] = cand[4 * cell + 1];
                sh.cur[3 * m + 2] = cand[4 * cell + 2];
                ++m;
            }
        }
        sh.M = m;
    }
    block_sync();
}

// Per-line constants of the E-step (calc_lvsq_angle :165-172 evaluates them again for every VP and every
// iteration): midpoint, direction and its norm, in c.drow as [5][ldn]; the line probabilities p_l of the
// current E-step follow at [5].
VPK_DEVFN void line_geometry_setup(EmCtx& c) {
    for (int n = tid(); n < c.N; n += nthreads()) {
        cgdp q = c.lp + 4 * (size_t)n;
        const double v2x = q[0] - q[2], v2y = q[1] - q[3];
        c.drow[n] = 0.5 * (q[0] + q[2]);
        c.drow[(size_t)c.ldn + n] = 0.5 * (q[1] + q[3]);
        c.drow[2 * (size_t)c.ldn + n] = v2x;
        c.drow[3 * (size_t)c.ldn + n] = v2y;
        c.drow[4 * (size_t)c.ldn + n] = norm2(v2x, v2y);
    }
    block_sync();
}

// ---- geometry of the row-sliced smoother (smooth_rows) -----------------------------------------------------------
// The N rows of lsim are cut into EIGHT slices of jch = ceil(N / 8) consecutive rows (the summation order every stored
// result was produced with: per (column, VP) eight ascending fma chains, then ((((p0 + p1) + p2) + ...) + p7).  The
// operand panel w_[line][vp] is kept slice by slice, [slice][row in slice][W], with the slice stride padded to 16 mod 32
// doubles so that the two slices whose rows one ds_read_b64 touches (lanes 0-31: two rows of 16 lanes) lie in
// different halves of the 64 banks.
constexpr int RS_TT = 4;                               // VPs per reduction round (one output per lane and round)
constexpr int RS_RED_DOUBLES = RS_TT * 16 * 9;         // per wave: [vp][column][8 slices + 1 pad]
constexpr int RS_PANEL_FLAG = 0x100;                   // sh.ibuf[5] = RS_PANEL_FLAG + W: the E-step left this layout
VPK_DEV int rs_jchunk(int N) { return (N + 7) >> 3; }
VPK_DEV int rs_sstride(int jch, int W) { const int q = jch * W; return q + ((16 - q) & 31); }
VPK_DEV int rs_panel_doubles(int jch, int W) { return 8 * rs_sstride(jch, W) + 32; }   // + slack: lanes read 16 + i past a row
VPK_DEV int rs_row(int n, int jch, int S, int W) { const int sl = n / jch; return sl * S + (n - sl * jch) * W; }
// Which smoother the next smooth() takes for M hypotheses -- decided in ONE place because the E-step writes the panel in
// that smoother's layout.  0: none in LDS (wsrc in HBM), 1: smooth_full's [line][W], 2: smooth_rows' sliced layout.
// VPs per pass of smooth_rows when the whole panel does not fit: the widest multiple of 8 (<= 32) whose sliced panel and
// the reduction scratch fit the LDS budget; 0 = not even 8
VPK_DEV int rs_wfit(const EmCtx& c) {
    const int jch = rs_jchunk(c.N);
    for (int w = 32; w >= MT; w -= MT)
        if (rs_panel_doubles(jch, w) + 8 * RS_RED_DOUBLES <= c.wt_doubles) return w;
    return 0;
}
// Which smoother the next smooth() takes for M hypotheses -- decided in ONE place because the E-step writes the panel in
// that smoother's layout.  0: none in LDS (wsrc in HBM; smooth_full in passes or smooth_blocks), 1: smooth_full's
// [line][W], 2: smooth_rows' sliced layout, 3: wsrc in HBM, smooth_rows in passes of rs_wfit() VPs.
VPK_DEV int smooth_plan(const EmCtx& c, int M) {
    const int N = c.N;
    if (!c.prm.use_weights || M <= 0) return 0;
    const int Wp = ((M + MT - 1) / MT) * MT;
    const bool rows_ok = WAVE == 64 && nwaves() == 8 && c.smoother != 1;
    const int colw = N > WAVE ? 2 * WAVE : WAVE;                   // smooth_full's column groups: when they divide evenly
    const bool direct = (((N + colw - 1) / colw) % 8) == 0;        // among the waves it sums ALL rows in one chain
    if (M <= 32) {
        if (rows_ok && !direct && rs_panel_doubles(rs_jchunk(N), Wp) + 8 * RS_RED_DOUBLES <= c.wt_doubles) return 2;
        if ((long long)N * Wp <= c.wt_doubles) return 1;
    }
    // In passes: smooth_rows keeps smooth_full's eight-slice order, so it may stand in wherever smooth_full would run
    // (a panel of at least 8 VPs fits the OLD layout: wfit >= 8), never for smooth_blocks (one chain per column).
    if (rows_ok && !direct && (c.wt_doubles / N) / MT >= 1 && rs_wfit(c) >= MT) return 3;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// E-step: calc_probabilities (probability_functions.py:99-147, "angle" branch)
// ---------------------------------------------------------------------------------------------
// X points at sh.cur or sh.nxt.  Writes lvsq[m][n], pvl[m][n], wsrc[n][m]; floors sh.s (:139).
// (Round 6, measured: the body inlined into em_run's main loop -- to save the callee-saved register traffic of one call per iteration,
//  which did pay for the smoother's thin wrappers -- makes the E-step 2.5 x SLOWER, 51 -> 130 ms of workgroup time per YUD batch: inside
//  em_run's register allocation the line loop spills.  The phases stay functions of their own.)
VPK_DEVFN void estep(EmCtx& c, const double* X) {
    Shared& sh = SH();
    const int M = sh.M, N = c.N;
    const double kk = -0.5 / (sh.sigma_prior * sh.sigma_prior);
    long long tq_ = clock_ticks();
    // prior p(v): a group of VPG lanes per VP (four VPs per wave: the asin/cos chains of four VPs run in one
    // wave's lanes), lanes over mixture components (calc_angles :252-259, calc_pdf :8-40)
    {
        constexpr int G = VPG;
        const int gl = lane() % G;
        const int per_round = nwaves() * (WAVE / G);
        for (int m = wave_id() * (WAVE / G) + lane() / G; m < M; m += per_round) {
            double x0 = X[3 * m], x1 = X[3 * m + 1], x2 = X[3 * m + 2];
            double beta = asin(x1);
            double inner = x0 / cos(beta);
            inner = inner < 1 ? inner : (is_nan(inner) ? inner : 1.0);
            inner = inner > -1 ? inner : (is_nan(inner) ? inner : -1.0);
            double alpha = asin(inner);
            double acc = 0.0;
            for (int q = gl; q < sh.ncomp; q += G) {
                double ma = sh.pma[q], mb = sh.pmb[q];
                double d1 = (alpha - ma) * (alpha - ma) + (beta - mb) * (beta - mb);
                double d2 = (alpha - ma + PI_D) * (alpha - ma + PI_D) + (beta + mb) * (beta + mb);
                double d3 = (alpha - ma - PI_D) * (alpha - ma - PI_D) + (beta + mb) * (beta + mb);
                double d4 = (alpha + ma) * (alpha + ma) + (beta - mb - PI_D) * (beta - mb - PI_D);
                double e4 = exp_underflow(d4 * kk);          // the fifth term duplicates the fourth (:25-26)
                double p = (((exp_underflow(d1 * kk) + exp_underflow(d2 * kk)) + exp_underflow(d3 * kk)) + e4) + e4;
                acc += p * sh.pw[q];
            }
            acc = group_sum<G>(acc);
            if (gl == 0) {
                sh.pv[m] = acc;
                sh.vx[m] = x0 / x2;                          // calc_lvsq_angle :165-166
                sh.vy[m] = x1 / x2;
                double sm = sh.s[m];
                sm = sm > 1e-200 ? sm : 1e-200;              // calc_plv :139 (in place)
                sh.s[m] = sm;
                sh.k2[m] = 1.0 / sqrt(2 * PI_D * sm);        // :145
            }
        }
    }
    block_sync();
    if (tid() == 0) sh.dbuf[14] += lap(tq_);
    // When the smoother's whole operand panel fits in LDS the weights go there directly ([line][vp],
    // Wp = M rounded to the VP tile) as well as to HBM, and smooth_full skips its staging pass.
    const int Wp = ((M + MT - 1) / MT) * MT;
    const int plan = smooth_plan(c, M);                      // 1: [line][Wp] for smooth_full, 2: slice by slice for smooth_rows
    const bool panel = plan == 1 || plan == 2;
    const int rs_jch = rs_jchunk(N), rs_S = rs_sstride(rs_jch, Wp);
    double* wt = WT();
    // one thread per line; the VP loop is unrolled four deep with the four sqrt/div/exp chains written
    // side by side (independent until the ordered sum), because a lone wave per SIMD is bound by the
    // latency of that dependent chain, not by issue
    cgdp gmx = c.drow, gmy = c.drow + c.ldn, gvx = c.drow + 2 * (size_t)c.ldn, gvy = c.drow + 3 * (size_t)c.ldn,
         gn2 = c.drow + 4 * (size_t)c.ldn;
    constexpr int EU = 4;
    // LANES PER LINE (round 6).  One thread per line leaves 512 - N threads idle and the busy ones with M dependent sqrt / div / exp
    // chains each: at the YUD shape (N ~ 250, M ~ 22) the line part took as long as the smoother's row loops.  When the panel is in
    // LDS and T N <= 512, T = 2, 4 or 8 ADJACENT lanes share a line, each a contiguous run of ceil(M / T) VPs.  Every (line, VP) value
    // is the same expression as below; p_l (:116) is still ONE chain over the VPs in ascending order -- lane h takes the running sum
    // from lane h - 1 and continues it over its own terms, re-read from the line's panel row -- so every output has the same bits.
    int T = 1;
    if (panel && WAVE == 64 && c.smoother != 1)              // (vpk_em_set_smoother(1): the forms of the earlier rounds, for the bit-equality test)
        while (T < 8 && 2 * T * N <= nthreads()) T *= 2;
    if (T > 1) {
        const int n_ = tid() / T, h = tid() - n_ * T;
        const bool on = n_ < N;
        const int n = on ? n_ : N - 1;
        const int Mh = (M + T - 1) / T;
        const int m_lo = on ? (h * Mh < M ? h * Mh : M) : 0, m_hi = on ? (m_lo + Mh < M ? m_lo + Mh : M) : 0;
        const double lmx = gmx[n], lmy = gmy[n], v2x = gvx[n], v2y = gvy[n], n2 = gn2[n];
        const double lw = c.lweight[n];
        gdp lvq = c.lvsq + n, pvq = c.pvl + n;
        double* wl = wt + (plan == 2 ? (size_t)rs_row(n, rs_jch, rs_S, Wp) : (size_t)n * Wp);
        int m = m_lo;
        for (; m + EU <= m_hi; m += EU) {
            double lv[EU], tt[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const double v1x = lmx - sh.vx[m + u], v1y = lmy - sh.vy[m + u];
                const double n1 = norm2(v1x, v1y);
                const double cc = 1 - fabs(dot2(v1x, v1y, v2x, v2y) / (n1 * n2));
                lv[u] = cc * cc;                             // :174
            }
#pragma unroll
            for (int u = 0; u < EU; ++u)
                tt[u] = (exp_underflow(-(lv[u] / (2 * sh.s[m + u]))) * sh.k2[m + u]) * sh.pv[m + u];   // calc_plv :137-145
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                lvq[(size_t)(m + u) * c.ldn] = lv[u];
                wl[m + u] = tt[u];
            }
        }
        for (; m < m_hi; ++m) {
            const double v1x = lmx - sh.vx[m], v1y = lmy - sh.vy[m];
            const double n1 = norm2(v1x, v1y);
            const double cc = 1 - fabs(dot2(v1x, v1y, v2x, v2y) / (n1 * n2));
            const double lv1 = cc * cc;
            lvq[(size_t)m * c.ldn] = lv1;
            wl[m] = (exp_underflow(-(lv1 / (2 * sh.s[m]))) * sh.k2[m]) * sh.pv[m];
        }
        double pl = 0.0;                                     // p_l = dot(p_lv, p_v) :116, in VP order, handed from lane to lane
        for (int hh = 0; hh < T; ++hh) {
            const double prev = wave_bcast(pl, (lane() + WAVE - 1) & (WAVE - 1));
            if (h == hh) {
                if (hh > 0) pl = prev;
                for (m = m_lo; m < m_hi; ++m) pl += wl[m];
            }
        }
        pl = wave_bcast(pl, lane() | (T - 1));               // the line's last lane holds the whole sum
        pl = (pl > 1e-12 || is_nan(pl)) ? pl : 1e-12;        // :117
        m = m_lo;
        for (; m + EU <= m_hi; m += EU) {
            double q[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) q[u] = wl[m + u];
#pragma unroll
            for (int u = 0; u < EU; ++u) q[u] = q[u] / pl;   // calc_pvl :128
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                pvq[(size_t)(m + u) * c.ldn] = q[u];
                wl[m + u] = q[u] * lw;                       // weight_matrix :519
            }
        }
        for (; m < m_hi; ++m) {
            const double q1 = wl[m] / pl;
            pvq[(size_t)m * c.ldn] = q1;
            wl[m] = q1 * lw;
        }
        if (on && h == T - 1)
            for (m = M; m < Wp; ++m) wl[m] = 0.0;            // padding of the last VP tile
    } else
    for (int n = tid(); n < N; n += nthreads()) {
        const double lmx = gmx[n], lmy = gmy[n], v2x = gvx[n], v2y = gvy[n], n2 = gn2[n];
        gdp lvq = c.lvsq + n, pvq = c.pvl + n;
        double* wl = wt + (plan == 2 ? (size_t)rs_row(n, rs_jch, rs_S, Wp) : (size_t)n * Wp);   // this line's panel row; parks p_lv p_v until p_l is known
        double pl = 0.0;
        int m = 0;
        for (; m + EU <= M; m += EU) {
            double lv[EU], tt[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const double v1x = lmx - sh.vx[m + u], v1y = lmy - sh.vy[m + u];
                const double n1 = norm2(v1x, v1y);
                const double cc = 1 - fabs(dot2(v1x, v1y, v2x, v2y) / (n1 * n2));
                lv[u] = cc * cc;                             // :174
            }
#pragma unroll
            for (int u = 0; u < EU; ++u)
                tt[u] = (exp_underflow(-(lv[u] / (2 * sh.s[m + u]))) * sh.k2[m + u]) * sh.pv[m + u];   // calc_plv :137-145
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                lvq[(size_t)(m + u) * c.ldn] = lv[u];
                if (panel) wl[m + u] = tt[u]; else pvq[(size_t)(m + u) * c.ldn] = tt[u];
                pl += tt[u];                                 // p_l = dot(p_lv, p_v) :116, in VP order
            }
        }
        for (; m < M; ++m) {
            const double v1x = lmx - sh.vx[m], v1y = lmy - sh.vy[m];
            const double n1 = norm2(v1x, v1y);
            const double cc = 1 - fabs(dot2(v1x, v1y, v2x, v2y) / (n1 * n2));
            const double lv1 = cc * cc;
            lvq[(size_t)m * c.ldn] = lv1;
            const double t1 = (exp_underflow(-(lv1 / (2 * sh.s[m]))) * sh.k2[m]) * sh.pv[m];
            if (panel) wl[m] = t1; else pvq[(size_t)m * c.ldn] = t1;
            pl += t1;
        }
        pl = (pl > 1e-12 || is_nan(pl)) ? pl : 1e-12;        // :117
        const double lw = c.lweight[n];
        gdp ws = c.wsrc + (size_t)n * c.mcap;
        m = 0;
        for (; m + EU <= M; m += EU) {
            double q[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) q[u] = panel ? wl[m + u] : pvq[(size_t)(m + u) * c.ldn];
#pragma unroll
            for (int u = 0; u < EU; ++u) q[u] = q[u] / pl;   // calc_pvl :128
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                pvq[(size_t)(m + u) * c.ldn] = q[u];
                if (panel) wl[m + u] = q[u] * lw;            // weight_matrix :519 (the HBM copy has no reader when the
                else ws[m + u] = q[u] * lw;                  //   smoother takes the whole panel from LDS in one pass)
            }
        }
        for (; m < M; ++m) {
            const double q1 = (panel ? wl[m] : pvq[(size_t)m * c.ldn]) / pl;
            pvq[(size_t)m * c.ldn] = q1;
            if (panel) wl[m] = q1 * lw; else ws[m] = q1 * lw;
        }
        for (m = M; m < Wp; ++m) {                           // padding of the last VP tile
            if (panel) wl[m] = 0.0; else ws[m] = 0.0;
        }
    }
    if (plan == 2)                                           // zero operand rows where a short or empty slice has no line
        for (int p = N * Wp + tid(); p < 8 * rs_jch * Wp; p += nthreads()) {
            const int j = p / Wp;
            wt[rs_row(j, rs_jch, rs_S, Wp) + (p - j * Wp)] = 0.0;
        }
    if (tid() == 0) sh.dbuf[15] += lap(tq_);
    if (tid() == 0) sh.ibuf[5] = plan == 2 ? RS_PANEL_FLAG + Wp : (plan == 1 ? Wp : 0);   // consumed (and cleared) by smooth()
    block_sync();
}

// ---------------------------------------------------------------------------------------------
// smoothing: weight_matrix (vp_localisation.py:515-524), the (M x N) . (N x N) product
// ---------------------------------------------------------------------------------------------
// w[m][k] = (w_[m][k] + bias*lweight[k] * sum_j w_[m][j] lsim[j][k]) / den[k].
// Work decomposition: an output block = (64*C consecutive columns) x (MT VPs); every wave owns
// whole blocks and walks ALL rows j for them, so no cross-wave reduction is needed and the result
// is deterministic.  A lane holds C adjacent columns (C = 2: one 16-byte load per row, a wave reads
// 1 KiB of contiguous lsim per row) and MT accumulators per column; rows are unrolled UNR deep so
// UNR independent loads are in flight per lane (HBM latency is hidden by bytes in flight, not by
// occupancy).  The w_ operand (wsrc[j][m]) is staged through LDS in row chunks and read as a
// wave-uniform broadcast.
template <int C, int UNR>
VPK_DEVFN void smooth_blocks(EmCtx& c) {
    Shared& sh = SH();
    const int M = sh.M, N = c.N;
    const int colw = WAVE * C;
    const int ncg = (N + colw - 1) / colw;
    const int ntile = (M + MT - 1) / MT;
    const int W = ntile * MT;                       // staged VPs per row (<= mcap)
    const int nblk = ncg * ntile;
    int JC = c.wt_doubles / W;                      // rows per LDS chunk
    if (JC > N) JC = N;
    const double bias = c.prm.wbias;
    double* wt = WT();
    for (int b0 = 0; b0 < nblk; b0 += nwaves()) {
        const int b = b0 + wave_id();
        const bool have = b < nblk;
        const int cg = have ? b % ncg : 0, tile = have ? b / ncg : 0;
        const int k = cg * colw + lane() * C;
        const bool live = have && k < N;
        double acc[MT][C];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < C; ++q) acc[t][q] = 0.0;
        for (int jc = 0; jc < N; jc += JC) {
            const int jn = (N - jc) < JC ? (N - jc) : JC;
            block_sync();                           // the previous chunk has been consumed
            for (int p = tid(); p < jn * W; p += nthreads()) {
                int j = p / W, m = p - j * W;
                wt[p] = c.wsrc[(size_t)(jc + j) * c.mcap + m];
            }
            block_sync();
            if (live) {
                cgdp lrow = c.lsim + (size_t)jc * c.ld + k;
                const double* wrow = wt + tile * MT;
                int j = 0;
                for (; j + UNR <= jn; j += UNR) {
                    double a[UNR][C];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) load_cols<C>(lrow + (size_t)(j + u) * c.ld, a[u]);
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            const double wv = wrow[(j + u) * W + t];
#pragma unroll
                            for (int q = 0; q < C; ++q) acc[t][q] = fma(wv, a[u][q], acc[t][q]);
                        }
                }
                for (; j < jn; ++j) {
                    double a1[C];
                    load_cols<C>(lrow + (size_t)j * c.ld, a1);
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const double wv = wrow[j * W + t];
#pragma unroll
                        for (int q = 0; q < C; ++q) acc[t][q] = fma(wv, a1[q], acc[t][q]);
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const int kk = k + q;
                if (kk < N) {
                    const double lw = c.lweight[kk], dn = c.den[kk];
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const int m = tile * MT + t;
                        if (m < M)
                            c.w[(size_t)m * c.ldn + kk] =
                                (c.wsrc[(size_t)kk * c.mcap + m] + bias * lw * acc[t][q]) / dn;
                    }
                }
            }
        }
    }
    block_sync();
}

// Single-pass smoother for images whose whole operand panel fits in LDS (N x W doubles).
// lsim is read exactly ONCE per call: every lane keeps NT*8 VP accumulators for its C columns.
// Work split: wave w owns row slice w (all waves equally loaded for any N) and walks every
// column group; the row-slice partials go through an L2-resident scratch and are summed in a fixed
// order (deterministic).  When the column groups divide evenly among the waves (ncg % nwaves == 0,
// e.g. N = 1000 with C = 2) each wave instead owns whole column groups and writes results directly.
// Loads are software-pipelined two batches deep so the L2/HBM latency of batch b+1 hides under the
// FMAs of batch b.
template <int NT, int C>
VPK_DEVFN void smooth_full(EmCtx& c, int m0) {
    Shared& sh = SH();
    constexpr int W = NT * MT;
    // rows per prefetch batch: two batches are in flight per lane (16 rows x 16 B at C = 2 -- the bytes
    // in flight, not occupancy, are what hides the ~1.5 us loaded memory latency), fewer when the
    // accumulators already take most of the register file
    constexpr int UNR = (NT * C >= 8) ? 4 : 8;
    const int N = c.N;
    const int M = sh.M - m0 < W ? sh.M - m0 : W;    // VPs handled by this call: [m0, m0 + M)
    const double bias = c.prm.wbias;
    double* wt = WT();
    long long tq_ = clock_ticks();
    if (!(m0 == 0 && sh.ibuf[5] == W)) {            // not left in place by the E-step
        for (int p = tid(); p < N * W; p += nthreads()) {
            const int j = p / W, m = p - j * W;
            wt[p] = (m < M) ? c.wsrc[(size_t)j * c.mcap + m0 + m] : 0.0;
        }
        block_sync();
    }
    if (tid() == 0) sh.dbuf[8] += lap(tq_);
    const int colw = WAVE * C;
    const int ncg = (N + colw - 1) / colw;
    const int nw = nwaves();
    const bool direct = (ncg % nw) == 0;            // whole column groups per wave, no row slicing
    const int R = direct ? 1 : nw;                  // (the reduction below handles up to 8 row slices)
    const int slice = direct ? 0 : wave_id();
    const int jchunk = (N + R - 1) / R;
    const int j0 = slice * jchunk;
    const int j1 = (j0 + jchunk) < N ? (j0 + jchunk) : N;
    for (int cg = direct ? wave_id() : 0; cg < ncg; cg += direct ? nw : 1) {
        const int k = cg * colw + lane() * C;
        const bool live = k < N;
        double acc[W][C];
#pragma unroll
        for (int t = 0; t < W; ++t)
#pragma unroll
            for (int q = 0; q < C; ++q) acc[t][q] = 0.0;
        if (live) {
            cgdp lcol = c.lsim + k;
            double a0[UNR][C], a1[UNR][C];
            double wb[2][MT];                       // operand double buffer: one 8-VP group ahead
            int j = j0;
            const int nfull = (j1 - j0) / UNR;      // full batches
            if (nfull > 0) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) load_cols<C>(lcol + (size_t)(j + u) * c.ld, a0[u]);
#pragma unroll
                for (int t = 0; t < MT; ++t) wb[0][t] = wt[(size_t)j * W + t];
            }
            for (int b = 0; b < nfull; ++b) {
                const bool more = b + 1 < nfull;
                if (more) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) load_cols<C>(lcol + (size_t)(j + UNR + u) * c.ld, a1[u]);
                }
                // UNR * NT steps, each: prefetch the next step's 8 operands, then 8*C FMAs on the current
                // ones.  pin8 keeps the steps in order (registers stay bounded), the prefetch hides the
                // LDS latency under the FMAs.
#pragma unroll
                for (int st = 0; st < UNR * NT; ++st) {
                    const int u = st / NT, g = st % NT;
                    const int nu = (st + 1) / NT, ng = (st + 1) % NT;
                    int nrow = j + nu;                          // row of the next step
                    if (st + 1 == UNR * NT) nrow = more ? j + UNR : j;   // last step: next batch (or a harmless re-read)
                    const double* nw = wt + (size_t)nrow * W + ng * MT;
#pragma unroll
                    for (int t = 0; t < MT; ++t) wb[(st + 1) & 1][t] = nw[t];
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int q = 0; q < C; ++q)
                            acc[g * MT + t][q] = fma(wb[st & 1][t], a0[u][q], acc[g * MT + t][q]);
#pragma unroll
                    for (int q = 0; q < C; ++q)
                        pin8(acc[g * MT][q], acc[g * MT + 1][q], acc[g * MT + 2][q], acc[g * MT + 3][q],
                             acc[g * MT + 4][q], acc[g * MT + 5][q], acc[g * MT + 6][q], acc[g * MT + 7][q]);
                }
                if (more) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
#pragma unroll
                        for (int q = 0; q < C; ++q) a0[u][q] = a1[u][q];
                }
                j += UNR;
            }
            // the slice's last rows (fewer than a batch): all their loads are issued before the first is used -- one
            // memory round trip instead of one per row (N = 245: 7 such rows per slice and column group, a third of the
            // phase's time); same rows in the same order
            const int rem = j1 - j;
            if (rem > 0) {
#pragma unroll
                for (int u = 0; u < UNR - 1; ++u)
                    if (u < rem) load_cols<C>(lcol + (size_t)(j + u) * c.ld, a0[u]);
#pragma unroll
                for (int u = 0; u < UNR - 1; ++u) {
                    if (u >= rem) break;
                    const double* wr = wt + (size_t)(j + u) * W;
#pragma unroll
                    for (int t = 0; t < W; ++t) {
                        const double wv = wr[t];
#pragma unroll
                        for (int q = 0; q < C; ++q) acc[t][q] = fma(wv, a0[u][q], acc[t][q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const int kk = k + q;
                if (kk >= N) continue;
                if (direct) {
                    const double lw = c.lweight[kk], dn = c.den[kk];
#pragma unroll
                    for (int t = 0; t < W; ++t)
                        if (t < M)
                            c.w[(size_t)(m0 + t) * c.ldn + kk] =
                                (wt[(size_t)kk * W + t] + bias * lw * acc[t][q]) / dn;
                } else {
#pragma unroll
                    for (int t = 0; t < W; ++t)
                        if (t < M) c.part[((size_t)slice * c.mcap + t) * c.ldn + kk] = acc[t][q];
                }
            }
        }
    }
    if (!direct) {
        block_sync();
        if (tid() == 0) sh.dbuf[9] += lap(tq_);
        // Work items = (column, batch of RB VPs), columns fastest (coalesced), dealt round-robin to ALL threads: with one
        // thread per column only N of the 512 threads worked, each through M / RB dependent batches of L2 round trips.
        // All the partials of a batch are loaded before any is used (the stores to w keep the compiler from hoisting
        // loads); each (column, VP) is still summed over the slices in the fixed order 0..7.
        constexpr int RB = 4;
        const int nbatch = (M + RB - 1) / RB;
        for (int item = tid(); item < N * nbatch; item += nthreads()) {
            const int t0 = (item / N) * RB, kk = item - (item / N) * N;
            const double blw = bias * c.lweight[kk], dn = c.den[kk];
            cgdp pcol = c.part + kk;
            double v[RB][8];
#pragma unroll
            for (int u = 0; u < RB; ++u)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    v[u][r] = (t0 + u < M && r < R) ? pcol[((size_t)r * c.mcap + t0 + u) * c.ldn] : 0.0;
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if (t0 + u >= M) break;
                double sum = 0.0;
#pragma unroll
                for (int r = 0; r < 8; ++r) sum += v[u][r];                                   // fixed order
                c.w[(size_t)(m0 + t0 + u) * c.ldn + kk] = (wt[(size_t)kk * W + t0 + u] + blw * sum) / dn;
            }
        }
    }
    block_sync();
    if (tid() == 0) sh.dbuf[10] += lap(tq_);
}

// lsim carries 8 rows more than the image has lines; the rows N .. 8 ceil(N / 8) - 1 are zero (smooth_rows walks them
// with zero operands where a slice is short or empty).  Once per image, after the matrix is in place.
VPK_DEVFN void zero_tail_rows(EmCtx& c) {
    const int N = c.N, jend = 8 * rs_jchunk(N);
    for (int p = tid(); p < (jend - N) * c.ld; p += nthreads()) c.lsim[(size_t)N * c.ld + p] = 0.0;
    block_sync();
}

// Row-sliced smoother: the same eight row slices and the same summation order as smooth_full, but no partial sum ever
// leaves the wave.  A wave owns 16 columns; its four rows of 16 lanes own the slices d and d + 4 (d = lane / 16), so the
// eight partials of a (column, VP) live in the four lanes {column, 16 + column, ...} of ONE wave and are summed through a
// 4.6 KB wave-private LDS scratch in the fixed order 0..7 -- no HBM/L2 round trip of the partials, no workgroup barrier
// before the results are written.  The w_ operands no longer come as wave-uniform broadcast reads (W / 2 ds_read_b128
// per row, as many LDS cycles as the FMAs take SIMD cycles): lane i of a row of 16 reads operand i (and 16 + i) of its
// slice's row ONCE and the FMAs take them through DPP row_newbcast (fmac8_row_bcast).  Per lane and row of a slice:
// one 8-byte lsim load (a row of 16 lanes = one 128-byte line), one or two 8-byte LDS reads, W FMAs.
template <int NT>
VPK_DEVFN void smooth_rows(EmCtx& c, int m0) {
    Shared& sh = SH();
    constexpr int W = NT * MT;
    constexpr int UNR = 4;                          // rows per load batch and slice; two batches are in flight
    const int N = uniform_int(c.N);
    m0 = uniform_int(m0);
    const int M = uniform_int(sh.M) - m0 < W ? uniform_int(sh.M) - m0 : W;    // VPs of this pass: [m0, m0 + M)
    const double bias = c.prm.wbias;
    double* wt = WT();
    long long tq_ = clock_ticks();
    const int jch = rs_jchunk(N), S = rs_sstride(jch, W);
    if (m0 != 0 || sh.ibuf[5] != RS_PANEL_FLAG + W) {   // not left in place by the E-step (passes; vpk_weight_matrix): stage it
        for (int p = tid(); p < N * W; p += nthreads()) {
            const int j = p / W, m = p - j * W;
            wt[rs_row(j, jch, S, W) + m] = (m < M) ? c.wsrc[(size_t)j * c.mcap + m0 + m] : 0.0;
        }
        for (int p = N * W + tid(); p < 8 * jch * W; p += nthreads()) {   // rows a short or empty slice does not have
            const int j = p / W;
            wt[rs_row(j, jch, S, W) + (p - j * W)] = 0.0;
        }
        block_sync();
    }
    if (tid() == 0) sh.dbuf[8] += lap(tq_);
    double* red = wt + rs_panel_doubles(jch, W) + wave_id() * RS_RED_DOUBLES;
    const int d = lane() >> 4, i = lane() & 15;
    const int jA0 = d * jch, jB0 = (d + 4) * jch;
    const double* oA = wt + (size_t)d * S + i;      // operand i of row r of the slice: oA[r * W] (and oA[r * W + 16])
    const double* oB = wt + (size_t)(d + 4) * S + i;
    const size_t ld = (size_t)uniform_int(c.ld), ldn = (size_t)uniform_int(c.ldn);
    cgdp lsim = c.lsim, lweight = c.lweight, den = c.den;   // (locals: the compiler barriers below would make it re-read c)
    gdp wout = c.w;
    // Every lane walks jch rows of both of its slices, also where a slice is short or empty (the last ones): the rows
    // N .. 8 jch - 1 exist in lsim as zeros (zero_tail_rows) and the operand rows of those "lines" are zero in the panel
    // (estep / the staging pass above), and fma(0, 0, acc) returns acc bit for bit (acc is never -0: it starts at +0
    // and a zero product is absorbed).  So the loop has no divergent branch, every load is unconditional with the
    // address (scalar row base) + (per-lane constant), and the compiler can count its waits.  The loads run one batch
    // of UNR rows ahead of the FMAs ACROSS column blocks: the last batch of a block requests the first rows of the
    // wave's next block, so only the first block of a call starts cold.
    const int nb = (jch + UNR - 1) / UNR;           // batches per column block; the last has jch - (nb - 1) UNR rows
    cgdp lbase = uniform_ptr(lsim);
    const unsigned rowbytes = (unsigned)ld * 8u;
    const int kstep = uniform_int(nwaves()) * 16;
    int k0 = uniform_int(wave_id()) * 16;
    if (k0 < N) {
        int k = k0 + i;
        int kc = k < N ? k : N - 1;                 // lanes past the last column stay active: they are operand sources
        unsigned offA = ((unsigned)jA0 * (unsigned)ld + (unsigned)kc) * 8u, offB = ((unsigned)jB0 * (unsigned)ld + (unsigned)kc) * 8u;
        double aA[UNR], aB[UNR], nA_[UNR], nB_[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int rn = u < jch ? u : jch - 1;
            cgdp rowp = (cgdp)((const VPK_GLOBAL char*)lbase + (size_t)rn * rowbytes);
            aA[u] = load_at(rowp, offA);
            aB[u] = load_at(rowp, offB);
        }
        for (;;) {
            const double lwk = lweight[kc];         // requested now, consumed after the row loop
            double dn = den[kc];
            const int k0n = k0 + kstep;
            const bool has_next = k0n < N;
            const int kn = k0n + i;
            const int kcn = has_next ? (kn < N ? kn : N - 1) : kc;
            const unsigned offAn = ((unsigned)jA0 * (unsigned)ld + (unsigned)kcn) * 8u, offBn = ((unsigned)jB0 * (unsigned)ld + (unsigned)kcn) * 8u;
            double accA[W], accB[W];
#pragma unroll
            for (int t = 0; t < W; ++t) { accA[t] = 0.0; accB[t] = 0.0; }
            double cA0 = oA[0], cA1 = W >= 24 ? oA[16] : 0.0, cB0 = oB[0], cB1 = W >= 24 ? oB[16] : 0.0;   // operands of row 0
            // one batch: request the rows of the following batch into (nxA, nxB), then the FMAs of this batch's rows
            // out of (cuA, cuB).  The two register sets swap roles from batch to batch (no copies: a copy would wait
            // for the loads it moves).
            auto batch = [&](int b, double (&cuA)[UNR], double (&cuB)[UNR], double (&nxA)[UNR], double (&nxB)[UNR])
                             __attribute__((always_inline)) {
                const int r = b * UNR;
                const bool lastb = b + 1 == nb;
                const int nrow = lastb ? jch - r : UNR;
                const int rnext = lastb ? 0 : r + UNR;          // first row of the batch requested now
                const unsigned oa = lastb ? offAn : offA, ob = lastb ? offBn : offB;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    int rn = rnext + u;
                    rn = rn < jch ? rn : jch - 1;
                    cgdp rowp = (cgdp)((const VPK_GLOBAL char*)lbase + (size_t)rn * rowbytes);
                    nxA[u] = load_at(rowp, oa);
                    nxB[u] = load_at(rowp, ob);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (u < nrow) {                 // wave-uniform
                        int rq = r + u + 1;         // the next row's operands are requested before this row's FMAs
                        rq = rq < jch ? rq : 0;     // (after the block's last row: row 0 again, for the next block)
                        const double* qA = oA + (size_t)rq * W;
                        const double* qB = oB + (size_t)rq * W;
                        const double xA0 = qA[0], xA1 = W >= 24 ? qA[16] : 0.0, xB0 = qB[0], xB1 = W >= 24 ? qB[16] : 0.0;
                        fmac8_row_bcast<0>(accA, cA0, cuA[u]);
                        if (W >= 16) fmac8_row_bcast<8>(accA + (W >= 16 ? 8 : 0), cA0, cuA[u]);
                        if (W >= 24) fmac8_row_bcast<0>(accA + (W >= 24 ? 16 : 0), cA1, cuA[u]);
                        if (W >= 32) fmac8_row_bcast<8>(accA + (W >= 32 ? 24 : 0), cA1, cuA[u]);
                        fmac8_row_bcast<0>(accB, cB0, cuB[u]);
                        if (W >= 16) fmac8_row_bcast<8>(accB + (W >= 16 ? 8 : 0), cB0, cuB[u]);
                        if (W >= 24) fmac8_row_bcast<0>(accB + (W >= 24 ? 16 : 0), cB1, cuB[u]);
                        if (W >= 32) fmac8_row_bcast<8>(accB + (W >= 32 ? 24 : 0), cB1, cuB[u]);
                        cA0 = xA0; cA1 = xA1; cB0 = xB0; cB1 = xB1;
                    }
                }
            };
            int b = 0;
            for (; b + 1 < nb; b += 2) {
                batch(b, aA, aB, nA_, nB_);
                batch(b + 1, nA_, nB_, aA, aB);
            }
            const bool odd = b < nb;
            if (odd) batch(b, aA, aB, nA_, nB_);    // the next block's first rows are in (nA_, nB_): moved after the rounds
        // the eight partials of every (VP, column) of this wave, summed in slice order: RS_TT VPs per round through the
        // wave's scratch [vp][column][slice]; lane (d, i) writes its slices d and d + 4 and finishes VP t0 + d of column i
        if (tid() == 0) sh.dbuf[9] += lap(tq_);     // row loops (wave 0)
        const double blw = bias * lwk;
        asm volatile("" : "+v"(dn));                // dn has arrived before the rounds: no wait inside them (a wait there
                                                    //   would also wait for the previous round's store)
        const double* wk = wt + rs_row(kc, jch, S, W);       // w_[kc][.]
        double* rw = red + (size_t)i * 9 + d;
        const double* rr_ = red + ((size_t)d * 16 + i) * 9;
#pragma unroll
        for (int t0 = 0; t0 < W; t0 += RS_TT) {
            if (t0 < M) {
#pragma unroll
                for (int u = 0; u < RS_TT; ++u) {
                    rw[(size_t)u * 16 * 9] = accA[t0 + u];
                    rw[(size_t)u * 16 * 9 + 4] = accB[t0 + u];
                }
                wave_lds_order();
                double sum = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) sum += rr_[q];                                      // fixed order
                const int t = t0 + d;
                if (t < M && k < N) wout[(size_t)(m0 + t) * ldn + k] = (wk[t] + blw * sum) / dn;
                wave_lds_order();
            }
        }
        if (tid() == 0) sh.dbuf[10] += lap(tq_);    // reduction rounds (wave 0)
            if (!has_next) break;
            if (odd) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) { aA[u] = nA_[u]; aB[u] = nB_[u]; }
            }
            k0 = k0n; k = kn; kc = kcn; offA = offAn; offB = offBn;
        }
    }
    block_sync();
    if (tid() == 0) sh.dbuf[10] += lap(tq_);        // + waiting for the other waves
}

// ---------------------------------------------------------------------------------------------
// Sparse smoother (round 4, NOT the default: vpk_em_set_smoother(h, 2)): the same sums in the same order as smooth_rows /
// smooth_full, without the zero terms.  Built to test the lead "82-85 % of the operands are zeros" and measured SLOWER than
// the dense row-sliced kernel on the bench's batch (smoothing 114 ms of workgroup time per YUD batch against 84 ms; 58-62
// against 55 us per call at N = 364, M = 24): a wave issues at most one instruction every four cycles and the workgroup has
// two waves per SIMD, so what counts is instructions per wave, and the sparse kernel spends ~45 (mostly scalar: next set
// bit, slice boundary test, a test and a branch per VP, two v_readlane per weight) per staged row step where the dense
// kernel spends its W = 24 v_fmac_f64_dpp and almost nothing else -- six times fewer FMAs bought with more than six times
// the control instructions.  The LDS traffic (21 us estimated) and the staging (1 % of the time waiting for the DMA) are
// not what bounds it; the barrier per block costs 28 % (the waves own VPs, and VPs have unequal numbers of lines).  Kept
// as an option with its bit-equality test; DESIGN.md section 8.
//
// 82-85 % of the operands w_[line][vp] = p_vl * lweight are exact zeros: a line has a non-zero responsibility for two to
// four of ~20 hypotheses, exp underflows to 0 for the rest (sigma^2 <= 1e-6, :306).  fma(0, x, acc) returns acc bit for
// bit for a finite x (acc is never -0), so leaving those terms out changes nothing -- provided lsim holds no NaN / Inf
// (sh.ibuf[2], set from the row sums: a line of length 0).  The dense kernels cannot skip them: one of their FMA
// instructions covers four lines (slices) at once.  Here
//   * a WAVE owns up to four VPs (t = wave, wave + 8, ...), a LANE owns the columns k = lane, lane + 64, ... (CMAX per lane):
//     the accumulators of a (VP, column) never leave their lane;
//   * lsim is staged through LDS in blocks of SP_R consecutive rows by all threads (every element fetched once per call,
//     16-byte loads one block ahead of the block being used: the traffic of the dense kernels), two buffers, ONE
//     workgroup barrier per block;
//   * the wave's weights sit in registers, lane l holding w_[64 ci + l][t]; per block and VP a ballot gives the rows of
//     the block with a non-zero weight, and for each of them the wave reads the staged row (conflict-free 8-byte reads)
//     and issues ONE fma per column group with the weight as a scalar operand (v_readlane);
//   * the summation order of the dense kernels is kept: rows ascending, a partial per slice of jch = ceil(N / 8) rows,
//     the eight partials added in slice order (an empty slice adds +0) -- hence the same bits in every output
//     (tests/test_gpu_em.py compares the three smoothers with array_equal).
// Applies where smooth_rows applied and N <= 64 CMAX; everything else keeps its kernel.
// ---------------------------------------------------------------------------------------------
constexpr int SP_R = 16;                                // rows per staged block
static_assert((size_t)SP_R + 1 <= vpk::EM_LSIM_PAD_ROWS, "smooth_sparse stages rows up to 16 ceil(N / 16) - 1 plus one piece's overrun: em_layout must pad lsim for them");
constexpr int SP_CMAX = 7;                              // column groups of 64 per lane: N <= 448
VPK_DEV int sp_cgroups(int N) { return (N + WAVE - 1) / WAVE; }
VPK_DEV int sp_ldw(int C) { return ((C + 1) / 2) * 2 * WAVE; }   // staged row: whole 1 KB DMA pieces (128 doubles)
VPK_DEV int sp_ring(int C) { return C <= 6 ? 3 : 2; }           // staged blocks in LDS (one in use, the others in flight)
VPK_DEV bool sparse_smoother_fits(const EmCtx& c) {
    const int C = sp_cgroups(c.N);
    return WAVE == 64 && nwaves() == 8 && c.smoother == 2 && c.N > 0 && C <= SP_CMAX &&
           sp_ring(C) * SP_R * sp_ldw(C) <= c.wt_doubles;
}
template <int C>                                        // C = column groups of 64 in use: ceil(N / 64)
VPK_DEVFN void smooth_sparse(EmCtx& c, int m0) {
    Shared& sh = SH();
    constexpr int R = SP_R, NB = C <= 6 ? 3 : 2, VPW = 4;   // rows per block, ring size (sp_ring), VPs per wave
    constexpr int AHEAD = NB - 1;                   // blocks in flight ahead of the one in use
    constexpr int LDW = ((C + 1) / 2) * 2 * (WAVE >= 2 ? WAVE : 2);   // row stride of a staged row (doubles)
    constexpr int DPR = LDW / 128 > 0 ? LDW / 128 : 1;                 // DMA pieces per row
    constexpr int DPB = R * DPR / 8;                // DMA pieces per wave and block (8 waves: two rows' worth)
    constexpr int BPG = (WAVE >= R ? WAVE : R) / R; // blocks per group of 64 rows
    const int N = uniform_int(c.N);
    m0 = uniform_int(m0);
    const int M = uniform_int(sh.M) - m0 < 32 ? uniform_int(sh.M) - m0 : 32;   // VPs of this pass: [m0, m0 + M)
    const int jch = rs_jchunk(N);
    const int nblk = (N + R - 1) / R;
    const size_t ld = (size_t)uniform_int(c.ld), ldn = (size_t)uniform_int(c.ldn);
    const double bias = c.prm.wbias;
    cgdp lsim = uniform_ptr(c.lsim);
    cgdp lweight = c.lweight, den = c.den, pvl = c.pvl;
    gdp wout = c.w;
    double* buf = WT();                             // [NB][R][LDW]: the ring
    const unsigned buf_lds = lds_addr_of(buf);
    long long tq_ = clock_ticks();
    const int wv = uniform_int(wave_id()), ln = lane();
    // ---- this wave's weights: wreg[q][ci] = w_[64 ci + lane][m0 + wv + 8 q] = p_vl * lweight (weight_matrix :519).  All of
    //      them up front: the main loop then has no vector-memory operation but its DMA, whose completion it counts ----
    double wreg[VPW][C];
    {
        // unconditional loads (indices clamped into the arrays) so that they are issued together, selected afterwards
        double lwv[C], raw[VPW][C];
#pragma unroll
        for (int ci = 0; ci < C; ++ci) {
            const int j = ci * WAVE + ln;
            lwv[ci] = lweight[j < N ? j : N - 1];
        }
#pragma unroll
        for (int q = 0; q < VPW; ++q) {
            const int t = wv + 8 * q;
            cgdp row = pvl + (size_t)(m0 + (t < M ? t : M - 1)) * ldn;
#pragma unroll
            for (int ci = 0; ci < C; ++ci) {
                const int j = ci * WAVE + ln;
                raw[q][ci] = row[j < N ? j : N - 1];
            }
        }
#pragma unroll
        for (int q = 0; q < VPW; ++q)
#pragma unroll
            for (int ci = 0; ci < C; ++ci) {
                const double prod = raw[q][ci] * lwv[ci];
                wreg[q][ci] = (ci * WAVE + ln < N && wv + 8 * q < M) ? prod : 0.0;
            }
    }
    double part[VPW][C], tot[VPW][C];
#pragma unroll
    for (int q = 0; q < VPW; ++q)
#pragma unroll
        for (int cc = 0; cc < C; ++cc) { part[q][cc] = 0.0; tot[q][cc] = 0.0; }
    int bound[VPW];                                 // first row of the slice after the one part[q] belongs to
#pragma unroll
    for (int q = 0; q < VPW; ++q) bound[q] = jch;
    double blwk[C], dnk[C];                         // the results' per-column constants (:522), fetched now for the same reason
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int k = cc * WAVE + ln;
        const int kc = k < N ? k : N - 1;
        blwk[cc] = bias * lweight[kc];
        dnk[cc] = den[kc];
        pin1(blwk[cc]); pin1(dnk[cc]);
    }
    // every weight has arrived before the first DMA is issued: from here on the compiler has no vector-memory operation of
    // its own in flight and puts no s_waitcnt vmcnt into the main loop (one there would wait for the whole ring)
#pragma unroll
    for (int q = 0; q < VPW; ++q)
#pragma unroll
        for (int ci = 0; ci < C; ++ci) pin1(wreg[q][ci]);
    wait_vm<0>();
    // ---- staging by LDS-DMA: piece p of a block = (row p / DPR, 128 doubles p % DPR); wave w issues the pieces w, w + 8, ..
    //      Rows up to 8 ceil(N / 8) - 1 are zeros (zero_tail_rows), rows up to N + EM_LSIM_PAD_ROWS - 1 belong to lsim
    //      (em_layout): a block's last rows and a piece that runs past its row's ld doubles into the next row stay inside
    //      the matrix; what they hold meets zero operand bits / columns no lane owns a result for ----
    auto issue = [&](int blk) __attribute__((always_inline)) {
        const unsigned dst = buf_lds + (unsigned)((blk % NB) * R * LDW * 8);
#pragma unroll
        for (int u = 0; u < DPB; ++u) {
            const int p = wv + 8 * u;
            const int r = p / DPR, x = p - r * DPR;
            cgdp src = lsim + ((size_t)(blk * R + r) * ld + (size_t)x * 128);
            lds_dma16((unsigned)ln * 16u, (const void*)uniform_ptr(src), (unsigned)uniform_int((int)(dst + (unsigned)((r * LDW + x * 128) * 8))));
        }
    };
    issue(0);
    if (AHEAD > 1 && nblk > 1) issue(1);
    if (AHEAD > 2 && nblk > 2) issue(2);
    if (tid() == 0) sh.dbuf[8] += lap(tq_);
    // ---- the blocks: group ci of 64 rows = BPG blocks; (ci, q) static so that the accumulators stay in registers ----
#pragma unroll
    for (int ci = 0; ci < C; ++ci) {
        if (ci * BPG >= nblk) break;                // uniform
        unsigned long long nz[VPW];
#pragma unroll
        for (int q = 0; q < VPW; ++q) nz[q] = wave_ballot(wreg[q][ci] != 0.0);
        for (int b8 = 0; b8 < BPG; ++b8) {
            const int blk = ci * BPG + b8;
            if (blk >= nblk) break;                 // uniform
            // this wave's pieces of block blk have landed (the pieces of the AHEAD - 1 later blocks may still be in flight) ...
            const int later = nblk - 1 - blk;
            if (AHEAD >= 2 && later >= AHEAD - 1) wait_vm<(AHEAD - 1) * DPB>(); else wait_vm<0>();
            raw_barrier();                          // ... and every wave's; everybody is done with block blk - 1
            if (blk + AHEAD < nblk) issue(blk + AHEAD);   // into the buffer block blk - 1 used
            const double* rows = buf + (size_t)(blk % NB) * R * LDW + ln;
            unsigned mq[VPW], any = 0;              // per VP: the rows of this block with a non-zero weight; their union
#pragma unroll
            for (int q = 0; q < VPW; ++q) { mq[q] = (unsigned)(nz[q] >> (b8 * R)) & ((1u << R) - 1u); any |= mq[q]; }
            if (any == 0) continue;                 // uniform
            // One staged row serves all of the wave's VPs that have a weight for it; the row of the NEXT step is requested
            // before the FMAs of the current one (two register sets that swap roles).
            auto read_row = [&](int bit, double (&v)[C]) __attribute__((always_inline)) {
                const double* rp = rows + (size_t)bit * LDW;
#pragma unroll
                for (int cc = 0; cc < C; ++cc) v[cc] = rp[cc * WAVE];
            };
            int bit = __builtin_ctz(any);
            any &= any - 1;
            double va[C], vb[C];
            read_row(bit, va);
            auto step = [&](double (&cur)[C], double (&nxt)[C]) __attribute__((always_inline)) {
                const int cb = bit;
                const bool more = any != 0;
                bit = more ? __builtin_ctz(any) : cb;   // (after the last row: the same row once more -- the reads are issued
                any &= any - 1;                         //  unconditionally so that the compiler can count them: a conditional
                read_row(bit, nxt);                     //  request makes it wait for ALL outstanding reads before the FMAs)
                const int j = ci * WAVE + b8 * R + cb;
#pragma unroll
                for (int q = 0; q < VPW; ++q) {
                    if (!((mq[q] >> cb) & 1u)) continue;    // uniform
                    while (j >= bound[q]) {         // the row opens a later slice: close the current partial
#pragma unroll
                        for (int cc = 0; cc < C; ++cc) { tot[q][cc] += part[q][cc]; part[q][cc] = 0.0; }
                        bound[q] += jch;
                    }
                    const double wj = readlane_f64(wreg[q][ci], b8 * R + cb);
#pragma unroll
                    for (int cc = 0; cc < C; ++cc) part[q][cc] = fma(wj, cur[cc], part[q][cc]);
                }
                return more;
            };
            for (;;) {
                if (!step(va, vb)) break;
                if (!step(vb, va)) break;
            }
        }
    }
    if (tid() == 0) sh.dbuf[9] += lap(tq_);
    // ---- results: w[m][k] = (w_[k][m] + bias lweight[k] sum) / den[k]  (:522) ----
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int k = cc * WAVE + ln;
        if (k < N) {
#pragma unroll
            for (int q = 0; q < VPW; ++q) {
                const int t = wv + 8 * q;
                if (t < M) wout[(size_t)(m0 + t) * ldn + k] = (wreg[q][cc] + blwk[cc] * (tot[q][cc] + part[q][cc])) / dnk[cc];
            }
        }
    }
    block_sync();                                   // (also: nobody reads the ring any more -- the panel region is free)
    if (tid() == 0) sh.dbuf[10] += lap(tq_);
}
VPK_DEVFN void smooth_sparse_any(EmCtx& c, int m0) {
    switch (sp_cgroups(c.N)) {
        case 1: smooth_sparse<1>(c, m0); break;
        case 2: smooth_sparse<2>(c, m0); break;
        case 3: smooth_sparse<3>(c, m0); break;
        case 4: smooth_sparse<4>(c, m0); break;
        case 5: smooth_sparse<5>(c, m0); break;
        case 6: smooth_sparse<6>(c, m0); break;
        default: smooth_sparse<7>(c, m0); break;
    }
}

// (smooth and smooth_dispatch are inlined into their callers: as functions of their own they cost two more levels of callee-saved
//  register saves and restores -- scratch memory, i.e. HBM round trips at the stress shape -- per E-step for a chain of ifs)
VPK_DEV void smooth_dispatch(EmCtx& c);
VPK_DEV void smooth(EmCtx& c) {
    smooth_dispatch(c);
    if (tid() == 0) SH().ibuf[5] = 0;               // the E-step's panel is valid for one smoothing only
    block_sync();
}
VPK_DEV void smooth_dispatch(EmCtx& c) {
    Shared& sh = SH();
    const int M = sh.M, N = c.N;
    if (!c.prm.use_weights) {   // lsim == 0 and lweight == 1 (:180,:235): w = p_vl
        for (int m = 0; m < M; ++m)
            for (int k = tid(); k < N; k += nthreads()) c.w[(size_t)m * c.ldn + k] = c.wsrc[(size_t)k * c.mcap + m];
        block_sync();
        return;
    }
    if (M == 0) return;
    const int plan = smooth_plan(c, M);
    if (WAVE == 64 && (sh.ibuf[5] >= RS_PANEL_FLAG || plan == 2 || plan == 3)) {   // (an E-step's panel decides; none: the plan)
        if (sparse_smoother_fits(c) && sh.ibuf[2] == 0) {   // the zero terms left out (same sums, same order, same bits)
            for (int m0 = 0; m0 < M; m0 += 32) smooth_sparse_any(c, m0);
            return;
        }
        const int wpass = plan == 3 ? rs_wfit(c) : 32;      // VPs per pass
        for (int m0 = 0; m0 < M; m0 += wpass) {
            const int mm = (M - m0) < wpass ? (M - m0) : wpass;
            if (mm <= 8) smooth_rows<1>(c, m0);
            else if (mm <= 16) smooth_rows<2>(c, m0);
            else if (mm <= 24) smooth_rows<3>(c, m0);
            else smooth_rows<4>(c, m0);
        }
        return;
    }
    // single-pass kernel on as many VPs as the LDS panel holds (N x wfit doubles, wfit a multiple of the VP
    // tile, at most 32 accumulator sets per lane); more VPs than that take further passes over lsim
    int wfit = (int)((c.wt_doubles / N) / MT) * MT;
    if (wfit > 32) wfit = 32;
    if (wfit >= MT) {
        for (int m0 = 0; m0 < M; m0 += wfit) {
            const int mm = (M - m0) < wfit ? (M - m0) : wfit;
            if (N > WAVE) {
                if (mm <= 8) smooth_full<1, 2>(c, m0);
                else if (mm <= 16) smooth_full<2, 2>(c, m0);
                else if (mm <= 24) smooth_full<3, 2>(c, m0);
                else smooth_full<4, 2>(c, m0);
            } else {
                if (mm <= 8) smooth_full<1, 1>(c, m0);
                else if (mm <= 16) smooth_full<2, 1>(c, m0);
                else if (mm <= 24) smooth_full<3, 1>(c, m0);
                else smooth_full<4, 1>(c, m0);
            }
        }
        return;
    }
    if (N > WAVE) smooth_blocks<2, 8>(c);
    else smooth_blocks<1, 4>(c);
}

// ---------------------------------------------------------------------------------------------
// line -> VP association and counts: calc_vp_line_counts (vp_localisation.py:482-512)
// ---------------------------------------------------------------------------------------------
// np.argmax over VPs (first maximum; a NaN counts as the maximum).  hard = apply the outlier test.
VPK_DEVFN void assign_lines(EmCtx& c, bool hard) {
    Shared& sh = SH();
    const int M = sh.M, N = c.N;
    for (int n = tid(); n < N; n += nthreads()) {
        int best = 0;
        double bv = c.w[n];
        for (int m = 1; m < M; ++m) {
            double v = c.w[(size_t)m * c.ldn + n];
            if (!is_nan(bv) && (v > bv || is_nan(v))) { bv = v; best = m; }
        }
        if (hard && M > 0) {
            double dist = c.lvsq[(size_t)best * c.ldn + n];   // == calc_lvsq_single on the same VP slice
            if (dist > c.prm.outlier_thresh * sqrt(sh.s[best]))
                best = -1;                                    // :504
            else if (c.lweight[n] == 0)
                best = -1;                                    // :506
        }
        c.assoc[n] = best;
    }
    block_sync();
}
VPK_DEVFN void count_lines(EmCtx& c) {
    Shared& sh = SH();
    const int M = sh.M, N = c.N;
    for (int m = wave_id(); m < M; m += nwaves()) {
        int cnt = 0;
        double cw = 0.0;
        for (int n = lane(); n < N; n += WAVE)
            if (c.assoc[n] == m) { ++cnt; cw += c.lweight[n]; }
        cnt = wave_sum_int(cnt);
        cw = wave_sum(cw);
        if (lane() == 0) { sh.cnt[m] = (double)cnt; sh.cntw[m] = cw; }
    }
    block_sync();
}

// remove the VPs flagged in sh.removed from cur / nxt / s (np.delete along the VP axis)
VPK_DEV void compact_vps(EmCtx& c) {
    Shared& sh = SH();
    static_assert(MAXM <= 64, "compact_vps: one lane per hypothesis");
    if (WAVE == 64 && c.smoother != 1) {
        // MAXM = 64 hypotheses = the lanes of one wave: lane m keeps its VP's values in registers, a ballot of the survivors gives
        // every survivor its new index (popcount of the survivors below it), and the common case -- nothing removed, every
        // iteration of a settled image -- writes nothing at all.  (One thread walking the list cost ~2 us per call.)
        if (wave_id() == 0) {
            const int M = sh.M, m = lane();
            const bool keep = m < M && !sh.removed[m];
            const unsigned long long km = wave_ballot(keep);
            const int kept = popcount64(km);
            if (kept != M) {
                double v[7];
                if (keep) {
                    for (int d = 0; d < 3; ++d) { v[d] = sh.cur[3 * m + d]; v[3 + d] = sh.nxt[3 * m + d]; }
                    v[6] = sh.s[m];
                }
                wave_lds_order();
                const int k = popcount64(km & lanes_below());
                if (keep && k != m) {
                    for (int d = 0; d < 3; ++d) { sh.cur[3 * k + d] = v[d]; sh.nxt[3 * k + d] = v[3 + d]; }
                    sh.s[k] = v[6];
                }
                if (m == 0) sh.M = kept;
            }
        }
        block_sync();
        return;
    }
    if (tid() == 0) {
        int k = 0;
        for (int m = 0; m < sh.M; ++m) {
            if (sh.removed[m]) continue;
            if (k != m) {
                for (int d = 0; d < 3; ++d) {
                    sh.cur[3 * k + d] = sh.cur[3 * m + d];
                    sh.nxt[3 * k + d] = sh.nxt[3 * m + d];
                }
                sh.s[k] = sh.s[m];
            }
            ++k;
        }
        sh.M = k;
    }
    block_sync();
}

// ---------------------------------------------------------------------------------------------
// M-step: calc_new_vanishing_point (vp_localisation.py:453-479) + variance (:301-307)
// ---------------------------------------------------------------------------------------------
// One group of VPG lanes per VP (four VPs per wave: the serial 3x3 eigen-solves of four VPs then run in
// the lanes of one wave instead of four waves' worth of rounds).  mode 0: soft (all lines, weights w[m]); mode 1: hard (lines with
// assoc == m, :353-392).  On return sh.removed[] / sh.err[] are set; nxt and s updated.
// LB = lines whose loads are in flight per lane (group_null_vector).  Four at the sizes whose arrays live in L2 (measured: eight is slower
// there); large images (N >= 512: ECD / HLW / the stress shape) walk N / 16 >= 32 lines per lane through arrays that come from HBM beside
// 255 other workgroups' lsim streams -- there the walk is a chain of memory round trips (73 us per M-step at the stress shape, 42 alone)
// and twice the loads in flight halve it.  Same lines in the same order per lane: same bits.
// (Round 6 also measured a whole WAVE per hypothesis for large images with at most eight hypotheses -- another summation order, so other
//  bits; every golden and all four config tables stayed green --: the stress shape's M-step 61 -> 39 us per call, and the launch 60.6 ->
//  59.0 ms: the time moves into the smoother, whose stream then shares the HBM with more workgroups.  Not worth new bits.)
template <int LB>
VPK_DEVFN void mstep_lb(EmCtx& c, int mode, double max_stdd) {
    Shared& sh = SH();
    const int M = sh.M, N = c.N;
    constexpr int G = VPG;
    const int gl = lane() % G;
    const int per_round = nwaves() * (WAVE / G);
    for (int m = wave_id() * (WAVE / G) + lane() / G; m < M; m += per_round) {
        cgdp wm = c.w + (size_t)m * c.ldn;
        double wmax = -1e300;
        int nsel = 0, selidx = -1;
        double sv = 0, sp = 0;
        cgdp lvs = c.lvsq + (size_t)m * c.ldn;
        cgdp pvl = c.pvl + (size_t)m * c.ldn;
        for (int n0 = gl; n0 < N; n0 += LB * G) {
            double pq[LB], lq[LB], wq[LB];
            int aq[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int n = n0 + u * G;
                const int nc = n < N ? n : 0;
                pq[u] = pvl[nc]; lq[u] = lvs[nc]; wq[u] = wm[nc];
                aq[u] = mode == 1 ? c.assoc[nc] : m;
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int n = n0 + u * G;
                if (n >= N) break;
                sv += lq[u] * pq[u];                          // :303 (all lines, also in hard mode :374)
                sp += pq[u];
                if (mode == 1 && aq[u] != m) continue;
                wmax = nanmax(wmax, wq[u]);
                ++nsel;
                selidx = n;
            }
        }
        wmax = group_max<G>(wmax);
        nsel = group_sum_int<G>(nsel);
        selidx = group_max_int<G>(selidx);
        sv = group_sum<G>(sv);
        sp = group_sum<G>(sp);
        if (mode == 1 && nsel == 0) {                         // :355-356 `continue`
            if (gl == 0) { sh.removed[m] = 0; sh.err[m] = -1.0; }
            continue;
        }
        bool valid = nsel > 0 && (wmax > 0 || wmax < 0);      // :456-460; NaN -> LinAlgError -> None
        double vp[3] = {0, 0, 0};
        if (valid && nsel > 1) {
            const VPK_GLOBAL int* assoc = c.assoc;
            // row weight w / max w (:462; hard mode: :358 then / 1 at :462)
            group_null_vector<G, LB>(c.l, N, [=](int n) { return (mode == 1 && assoc[n] != m) ? 0.0 : wm[n] / wmax; }, vp);
        }
        if (gl == 0) {
            int rem = 0;
            double err = -1.0;
            if (!valid) {
                rem = 1;                                      // newVP is None (:294-296)
            } else {
                if (nsel == 1) {                              // one row: LAPACK's reflector decides
                    cgdp ln = c.l + 3 * (size_t)selidx;
                    lapack_null_1row(ln[0], ln[1], ln[2], vp);    // the row is (w/max w) * l = 1 * l
                    double nr = norm3(vp[0], vp[1], vp[2]);
                    vp[0] /= nr; vp[1] /= nr; vp[2] /= nr;    // :472
                }
                double sg = sign_np(vp[2]);                   // :474
                vp[0] *= sg; vp[1] *= sg; vp[2] *= sg;
                sh.nxt[3 * m] = vp[0]; sh.nxt[3 * m + 1] = vp[1]; sh.nxt[3 * m + 2] = vp[2];
                double sm = exp(log(sv) - log(sp));           // :303-304
                sm = (sm < max_stdd || is_nan(sm)) ? sm : max_stdd;          // :306 np.minimum
                if (mode == 0)
                    sm = (sm > c.prm.s_thresh || is_nan(sm)) ? sm : c.prm.s_thresh;   // :307
                sh.s[m] = sm;
                if (is_nan(sm) || (mode == 1 && sm < c.prm.s_thresh)) {
                    rem = 1;                                  // :309-310 / :379-380
                } else {
                    double d = fabs(dot3(sh.cur[3 * m], sh.cur[3 * m + 1], sh.cur[3 * m + 2], vp[0], vp[1], vp[2]));
                    err = acos(d < 1.0 ? d : 1.0);            // :312
                    if (err > 1.5) rem = 1;                   // :316-317
                }
            }
            sh.removed[m] = rem;
            sh.err[m] = err;
        }
    }
    block_sync();
}

VPK_DEV void mstep(EmCtx& c, int mode, double max_stdd) {
    if (c.N >= 512 && c.smoother != 1) mstep_lb<8>(c, mode, max_stdd); else mstep_lb<4>(c, mode, max_stdd);
}

// max over the per-VP errors with np.maximum semantics (NaN sticks); VPs without an error are -1
VPK_DEV double max_err_of(const Shared& sh, int M) {
    double mx = 0.0;
    for (int m = 0; m < M; ++m) {
        double e = sh.err[m];
        if (e == -1.0) continue;
        mx = (is_nan(mx) || is_nan(e)) ? (is_nan(mx) ? mx : e) : (e > mx ? e : mx);
    }
    return mx;
}

// ---------------------------------------------------------------------------------------------
// merge_vps (vp_localisation.py:633-697)
// ---------------------------------------------------------------------------------------------
VPK_DEVFN void merge_vps(EmCtx& c, bool use_next, double thresh) {
    Shared& sh = SH();
    const int N = c.N;
    for (int guard = 0; guard < 4 * MAXM; ++guard) {
        const int M = sh.M;
        if (M <= 1) break;
        double* X = use_next ? sh.nxt : sh.cur;
        double bv = 1e300;
        int bi = 0x7fffffff;
        for (int p = tid(); p < M * M; p += nthreads()) {
            int j = p / M, k = p % M;
            double d = X[3 * j] * X[3 * k] + X[3 * j + 1] * X[3 * k + 1] + X[3 * j + 2] * X[3 * k + 2];
            double ang = (j == k) ? PI_D : fabs(acos(clip(fabs(clip(d, -1.0, 1.0)), -1.0, 1.0)));  // :691-696
            if (ang < bv || (ang == bv && p < bi)) { bv = ang; bi = p; }
        }
        block_argmin(sh, bv, bi);                             // first row-major minimum (:650)
        if (!(bv < thresh)) break;                            // :655,:679-680
        const int j = bi / M, k = bi % M;
        estep(c, X);                                          // :658 (at the caller's index)
        smooth(c);
        if (wave_id() == 0) {                                 // newVP from w[j] + w[k] (:661)
            cgdp wj = c.w + (size_t)j * c.ldn;
            cgdp wk = c.w + (size_t)k * c.ldn;
            double wmax = -1e300;
            for (int n = lane(); n < N; n += WAVE) wmax = nanmax(wmax, wj[n] + wk[n]);
            wmax = wave_max(wmax);
            bool valid = N > 0 && (wmax > 0 || wmax < 0);
            double sv = 0, sp = 0;
            cgdp lj = c.lvsq + (size_t)j * c.ldn;
            cgdp lk = c.lvsq + (size_t)k * c.ldn;
            cgdp pj = c.pvl + (size_t)j * c.ldn;
            cgdp pk = c.pvl + (size_t)k * c.ldn;
            for (int n = lane(); n < N; n += WAVE) {
                double pq = pk[n] + pj[n];
                sv += 0.5 * (lj[n] + lk[n]) * pq;             // :664
                sp += pq;                                     // :663
            }
            sv = wave_sum(sv);
            sp = wave_sum(sp);
            double vp[3] = {0, 0, 0};
            if (valid) wave_null_vector(c.l, N, [=](int n) { return (wj[n] + wk[n]) / wmax; }, vp);
            if (lane() == 0) {
                double sk = exp(log(sv) - log(sp));
                sh.s[k] = sk;                                 // :666 written BEFORE the abort test
                int ok = valid && !(sk > 0.01);               // :668 (max_stdd = 0.01)
                if (ok) {
                    double sg = sign_np(vp[2]);
                    X[3 * k] = vp[0] * sg; X[3 * k + 1] = vp[1] * sg; X[3 * k + 2] = vp[2] * sg;   // :672
                    for (int m = 0; m < M; ++m) sh.removed[m] = (m == j);                            // :674-675
                }
                sh.ibuf[0] = ok;
            }
        }
        block_sync();
        if (!sh.ibuf[0]) break;
        compact_vps(c);
    }
    block_sync();
}

// ---------------------------------------------------------------------------------------------
// 2-cluster average-linkage agglomeration == sklearn AgglomerativeClustering(linkage='average',
// connectivity=D, n_clusters=2, metric='precomputed') as called at vp_localisation.py:574-578.
// sklearn 0.18..1.7 behaviour restated: edges are the non-zero entries of D + D^T; repeatedly
// merge the closest connected pair; a neighbour shared by both gets (n_a d_a + n_b d_b)/(n_a+n_b),
// a neighbour of only one keeps its distance; the full tree is built and cut at the root, the
// cluster formed LAST (node 2n-3) gets label 0 (_hc_cut pops the larger node id first).
// Exact ties between candidate merges are resolved by Python heap order in sklearn; here by the
// smallest matrix position, and VPK_EM_FLAG_SPLIT_TIE is raised.
// D: n x n working copy in global memory (destroyed); member: n ints; labels -> member (0/1).
// ---------------------------------------------------------------------------------------------
VPK_DEVFN void cluster2(Shared&, int n, gdp D, gip member, gip csize) {
    Shared& sh = SH();
    for (int p = tid(); p < n * n; p += nthreads()) {
        int a = p / n, b = p % n;
        double v = D[p];
        if (a == b || !(v + D[(size_t)b * n + a] != 0.0)) D[p] = -1.0;   // no edge
    }
    for (int a = tid(); a < n; a += nthreads()) { member[a] = a; csize[a] = 1; }
    block_sync();
    int last_slot = -1;
    for (int t = 0; t < n - 2; ++t) {
        double bv = 1e300;
        int bi = 0x7fffffff;
        int ties = 0;
        for (int p = tid(); p < n * n; p += nthreads()) {
            int a = p / n, b = p % n;
            if (a <= b || csize[a] == 0 || csize[b] == 0) continue;
            double v = D[p];
            if (v < 0) continue;
            if (v < bv) { bv = v; bi = p; ties = 0; }
            else if (v == bv) { ties = 1; }
        }
        const double myv = bv;
        block_argmin(sh, bv, bi);
        if (bi == 0x7fffffff) {                               // graph exhausted: disconnected
            if (tid() == 0) sh.flags |= VPK_EM_FLAG_SPLIT_DISCONNECTED;
            break;
        }
        // tie detection: the winning value occurs at more than one candidate position
        if (tid() == 0) sh.ibuf[1] = 0;
        block_sync();
        if (myv == bv) atomic_add_int(&sh.ibuf[1], 1 + ties);
        block_sync();
        const int a = bi / n, b = bi % n;                     // a > b; the merged cluster lives in slot a
        const int na = csize[a], nb = csize[b];
        block_sync();
        for (int cidx = tid(); cidx < n; cidx += nthreads()) {
            if (cidx == a || cidx == b || csize[cidx] == 0) continue;
            double da = D[(size_t)a * n + cidx], db = D[(size_t)b * n + cidx];
            double nv;
            if (da >= 0 && db >= 0)
                nv = (na * da + nb * db) / (double)(na + nb);  // average_merge
            else
                nv = da >= 0 ? da : db;                        // only one side connected (or none: -1)
            D[(size_t)a * n + cidx] = nv;
            D[(size_t)cidx * n + a] = nv;
        }
        for (int q = tid(); q < n; q += nthreads())
            if (member[q] == b) member[q] = a;
        block_sync();
        if (tid() == 0) {
            csize[a] = na + nb;
            csize[b] = 0;
            if (sh.ibuf[1] >= 2) sh.flags |= VPK_EM_FLAG_SPLIT_TIE;
        }
        last_slot = a;
        block_sync();
    }
    for (int q = tid(); q < n; q += nthreads()) member[q] = (member[q] == last_slot) ? 0 : 1;
    block_sync();
}

// Same algorithm for small sets (the usual case: the lines of one VP; <= 72 lines in the YUD-shape bench), run by ONE
// wave out of LDS so that a merge costs no workgroup barrier.  D is an n x ld matrix in LDS (ld odd, -1 = no edge; a
// merged-away slot's row and column are set to -1, so the search needs no activity test per entry).  Per merge the
// wave walks the active rows a with lanes over the columns b < a (consecutive LDS words, no index decoding), every
// lane keeps its own best (distance, position), and ONE cross-lane arg-min ends the search -- a cross-lane
// reduction of a double + index costs ~1000 cycles on this part (scripts/ubench/wave_reduce.hip), as much as walking
// 30 rows, so the design minimises reductions, not LDS reads.  (Round 1 decoded a triangular pair index per entry:
// ~10 us per merge; a per-row nearest-neighbour cache with a reduction per rescanned row was no faster.)
// The matrix is the head of the LDS panel (WT()); behind it: member / csize [n] ints each.
constexpr int CLUSTER_LDS_MAX = 128;
VPK_DEV long long cluster_lds_doubles(int n) { return (long long)n * (n | 1) + (long long)n + 4; }
VPK_DEV int* cluster_lds_labels(double* D, int n) {
    return reinterpret_cast<int*>(D + (size_t)n * (n | 1));
}
VPK_DEVFN void cluster2_lds(int n) {
    Shared& sh = SH();
    // the matrix sits at the start of the LDS panel; deriving the pointer from the LDS symbol HERE (not taking it as
    // an argument of this non-inlined function) is what makes the accesses ds_read / ds_write instead of flat_*
    double* D = WT();
    const int ld = n | 1;
    int* member = cluster_lds_labels(D, n);
    int* csize = member + n;
    for (int a = tid(); a < n; a += nthreads()) { member[a] = a; csize[a] = 1; }
    block_sync();
    if (wave_id() == 0) {
        unsigned long long act[2];
        act[0] = n >= 64 ? ~0ull : ((1ull << n) - 1);
        act[1] = n > 64 ? (n >= 128 ? ~0ull : ((1ull << (n - 64)) - 1)) : 0ull;
        int last_slot = -1;
        bool tie_seen = false, disconnected = false;
        for (int t = 0; t < n - 2; ++t) {
            // Branch-free search, four rows per trip (their LDS reads are in flight together).  Distances are >= 0, so
            // their bit patterns order like unsigned integers, and "no edge" (-1.0: sign bit set) is larger than every
            // distance: one 64-bit integer compare per entry, no validity test.
            typedef unsigned long long u64;
            const u64 NONE = 0x7fe0000000000000ull;              // above every finite distance, below -1.0's pattern
            u64 bk = NONE;
            int bi = 0x7fffffff;
            int ties = 0;
            for (int c0 = 0; c0 < n; c0 += WAVE) {
                const int bq = c0 + lane();
                for (int a0 = (c0 > 0 ? c0 : 1); a0 < n; a0 += 4) {
                    u64 k[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int a = a0 + u;
                        const bool in = a < n && bq < a;
                        k[u] = in ? __double_as_longlong(D[(in ? a : 0) * ld + (in ? bq : 0)]) : ~0ull;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int pos = (a0 + u) * ld + bq;
                        const bool eq = k[u] == bk && k[u] < NONE;
                        const bool lt = k[u] < bk;
                        ties = lt ? 0 : (eq ? 1 : ties);
                        bi = (lt || (eq && pos < bi)) ? pos : bi;     // equal distances: the smallest position
                        bk = lt ? k[u] : bk;
                    }
                }
            }
            double bv = bk < NONE ? __longlong_as_double((long long)bk) : 1e300;
            if (!(bk < NONE)) bi = 0x7fffffff;
            const double myv = bv;
            wave_argmin(bv, bi);
            if (bi == 0x7fffffff) { disconnected = true; break; }
            if (wave_sum_int(myv == bv ? 1 + ties : 0) >= 2) tie_seen = true;
            const int ma = bi / ld, mb = bi - ma * ld;         // ma > mb; the merged cluster lives in slot ma
            const int na = csize[ma], nb = csize[mb];
            for (int cidx = lane(); cidx < n; cidx += WAVE) {
                if (cidx == ma || cidx == mb) continue;
                const double da = D[ma * ld + cidx], db = D[mb * ld + cidx];
                double nv;
                if (da >= 0 && db >= 0)
                    nv = (na * da + nb * db) / (double)(na + nb);  // average_merge
                else
                    nv = da >= 0 ? da : db;                        // only one side connected (or none: -1)
                D[ma * ld + cidx] = nv;                            // (dead slots hold -1 in every row: they stay -1)
                D[cidx * ld + ma] = nv;
                D[mb * ld + cidx] = -1.0;                          // slot mb leaves the search
                D[cidx * ld + mb] = -1.0;
            }
            if (lane() == 0) { D[ma * ld + mb] = -1.0; D[mb * ld + ma] = -1.0; }
            for (int q = lane(); q < n; q += WAVE)
                if (member[q] == mb) member[q] = ma;
            wave_sync();
            if (lane() == 0) { csize[ma] = na + nb; csize[mb] = 0; }
            wave_sync();
            act[mb >> 6] &= ~(1ull << (mb & 63));
            last_slot = ma;
        }
        for (int q = lane(); q < n; q += WAVE) member[q] = (member[q] == last_slot) ? 0 : 1;
        if (lane() == 0) {
            if (tie_seen) sh.flags |= VPK_EM_FLAG_SPLIT_TIE;
            if (disconnected) sh.flags |= VPK_EM_FLAG_SPLIT_DISCONNECTED;
        }
    }
    block_sync();
}

// ---------------------------------------------------------------------------------------------
// split_best_vp (vp_localisation.py:527-630).  Expects w = weight matrix of sh.cur.
// ---------------------------------------------------------------------------------------------
VPK_DEVFN void split_vp(EmCtx& c) {
    Shared& sh = SH();
    const int M = sh.M, N = c.N;
    if (M == 0 || c.cl == nullptr) return;
    long long tq_ = clock_ticks();
    assign_lines(c, false);                                   // weightIndices (:536) == vpAssoc (:551)
    double wmx = -1e300;
    for (int m = 0; m < M; ++m)
        for (int n = tid(); n < N; n += nthreads()) wmx = nanmax(wmx, c.w[(size_t)m * c.ldn + n]);
    wmx = block_max(sh, wmx);                                 // weightMatrix.max() (:539)
    // per VP: std of the folded line angle over lines with greedy weight > 0 (:541-544)
    for (int m = wave_id(); m < M; m += nwaves()) {
        int cnt = 0, call = 0;
        double sum = 0.0;
        for (int n = lane(); n < N; n += WAVE) {
            if (c.assoc[n] != m) continue;
            ++call;
            if (c.w[(size_t)m * c.ldn + n] / wmx > 0) { ++cnt; sum += c.langle[n]; }
        }
        cnt = wave_sum_int(cnt);
        call = wave_sum_int(call);
        sum = wave_sum(sum);
        double mean = sum / cnt;
        double sq = 0.0;
        for (int n = lane(); n < N; n += WAVE)
            if (c.assoc[n] == m && c.w[(size_t)m * c.ldn + n] / wmx > 0) {
                double d = c.langle[n] - mean;
                sq += d * d;
            }
        sq = wave_sum(sq);
        if (lane() == 0) {
            sh.err[m] = cnt > 0 ? sqrt(sq / cnt) : __builtin_nan("");   // np.std of an empty set is NaN
            sh.icnt[m] = call;
        }
    }
    block_sync();
    if (tid() == 0) {
        // worstVPs = argsort(stdd)[::-1] (:546-547): ascending with NaN last, reversed
        int order[MAXM];
        for (int m = 0; m < M; ++m) order[m] = m;
        for (int i = 1; i < M; ++i) {                         // stable insertion sort
            int key = order[i];
            double kv = sh.err[key];
            int j = i - 1;
            while (j >= 0) {
                double jv = sh.err[order[j]];
                bool greater = (is_nan(jv) && !is_nan(kv)) || (jv > kv);
                if (!greater) break;
                order[j + 1] = order[j];
                --j;
            }
            order[j + 1] = key;
        }
        int worst = -1;
        for (int m = 0; m < M; ++m) {
            int cand = order[M - 1 - m];
            double px = sh.cur[3 * m] / sh.cur[3 * m + 2];    // :557 tests VP m, not worstVPs[m]
            double py = sh.cur[3 * m + 1] / sh.cur[3 * m + 2];
            if (sh.icnt[cand] > 8 && (px > -1 && py > -1 && px < 1 && py < 1)) { worst = cand; break; }
        }
        sh.ibuf[3] = worst;
    }
    block_sync();
    if (wave_id() == 0) {                                     // assocLines, ascending (:552): ordered compaction
        const int worst = sh.ibuf[3];
        int nw = 0;
        if (worst >= 0)
            for (int n0 = 0; n0 < N; n0 += WAVE) {
                const int n = n0 + lane();
                const bool hit = n < N && c.assoc[n] == worst;
                const unsigned long long mask = wave_ballot(hit);
                if (hit) c.idx[nw + popcount64(mask & lanes_below())] = n;
                nw += popcount64(mask);
            }
        if (lane() == 0) sh.ibuf[4] = nw;
    }
    block_sync();
    const int worst = sh.ibuf[3], nw = sh.ibuf[4];
    if (tid() == 0) sh.dbuf[11] += lap(tq_);
    if (worst < 0) return;
    const double stdd = sh.s[worst] / 2;                      // :566
    gip member = c.idx + N;          // idx has room for 3N ints
    gip csize = c.idx + 2 * N;
    const int ld = nw | 1;
    const bool in_lds = nw <= CLUSTER_LDS_MAX && cluster_lds_doubles(nw) + 3 * nw <= c.wt_doubles;
    double* DL = WT();
    // Ldist (:568-572): 1 - cos(clip(2 acos |cos angle|, -pi/2, pi/2)) for every pair of the set's lines.  The lines'
    // direction vectors and norms are staged in LDS once (not two dependent global loads per pair), and for 2 phi <
    // pi/2 the value is 1 - (2 c^2 - 1) = 2 (1 - c)(1 + c) without acos / cos (as cos9_of_cos does for the similarity:
    // within 2e-16 of the library chain); the clipped branch is numpy's 1 - cos(pi/2) = 1 - 6.123e-17.
    // Staged [nw][vx, vy, norm]: behind the LDS matrix, alone in LDS, or -- a set of more lines than a third of the LDS
    // panel has doubles (3 nw > wt_doubles: thousands of lines on one VP) -- in the slot's p_vl rows in HBM (mcap x ldn >=
    // 8 N doubles; the E-step that follows every split rewrites them before anything reads them).  Same values, same
    // expressions, wherever they are staged.
    const bool dirs_lds = in_lds || 3 * (long long)nw <= c.wt_doubles;
    double* dirs = in_lds ? DL + cluster_lds_doubles(nw) : DL;
    gdp dirs_g = c.pvl;
    for (int a = tid(); a < nw; a += nthreads()) {
        cgdp q = c.lp + 4 * (size_t)c.idx[a];
        const double vx = q[0] - q[2], vy = q[1] - q[3];      // lines_points_cosangle :716-719
        const double nv = norm2(vx, vy);
        if (dirs_lds) { dirs[3 * a] = vx; dirs[3 * a + 1] = vy; dirs[3 * a + 2] = nv; }
        else { dirs_g[3 * (size_t)a] = vx; dirs_g[3 * (size_t)a + 1] = vy; dirs_g[3 * (size_t)a + 2] = nv; }
    }
    block_sync();
    for (long long p = tid(); p < (long long)nw * nw; p += nthreads()) {
        const int a = (int)(p / nw), b = (int)(p - (long long)a * nw);
        double v = 0.0;
        if (a != b) {
            double ax, ay, an, bx, by, bn;
            if (dirs_lds) { ax = dirs[3 * a]; ay = dirs[3 * a + 1]; an = dirs[3 * a + 2]; bx = dirs[3 * b]; by = dirs[3 * b + 1]; bn = dirs[3 * b + 2]; }
            else {
                ax = dirs_g[3 * (size_t)a]; ay = dirs_g[3 * (size_t)a + 1]; an = dirs_g[3 * (size_t)a + 2];
                bx = dirs_g[3 * (size_t)b]; by = dirs_g[3 * (size_t)b + 1]; bn = dirs_g[3 * (size_t)b + 2];
            }
            const double cc = clip(fabs(dot2(ax, ay, bx, by) / (an * bn)), -1.0, 1.0);
            const double COS_PI_4 = 0.70710678118654757;      // cos(pi/4): 2 phi >= pi/2 below it
            if (cc != cc) v = cc;
            else if (!(cc > COS_PI_4)) v = 1 - 6.123233995736766e-17;
            else v = 2 * ((1.0 - cc) * (1.0 + cc));
        }
        // (Ldist is bitwise symmetric, so sklearn's edge test D + D^T != 0 is v + v != 0)
        if (in_lds) DL[a * ld + b] = (a == b || !(v + v != 0.0)) ? -1.0 : v;
        else c.cl[p] = v;
    }
    block_sync();
    if (in_lds) {
        cluster2_lds(nw);
        const int* lmember = cluster_lds_labels(DL, nw);
        for (int q = tid(); q < nw; q += nthreads()) member[q] = lmember[q];
        block_sync();
    } else {
        cluster2(sh, nw, c.cl, member, csize);
    }
    if (tid() == 0) sh.dbuf[12] += lap(tq_);
    // per cluster: smallest right singular vector of the lweight-scaled lines (:580-602)
    // cluster label per line (-1 = not in the set), in the assoc scratch (recomputed before next use)
    gip lab = c.assoc;
    for (int n = tid(); n < N; n += nthreads()) lab[n] = -1;
    block_sync();
    for (int q = tid(); q < nw; q += nthreads()) lab[c.idx[q]] = member[q];
    block_sync();
    for (int cidx = wave_id(); cidx < 2; cidx += nwaves()) {
        int cnt = 0;
        for (int q = lane(); q < nw; q += WAVE) cnt += (member[q] == cidx);
        cnt = wave_sum_int(cnt);
        double vp[3] = {0, 0, 0};
        if (cnt >= 3) {                                       // :592-593
            // rows = lweight * l over the lines of this cluster (:580-595); evaluated over all N lines
            // with weight 0 outside the cluster, so the gather order does not matter
            cgdp lwt = c.lweight;
            wave_null_vector(c.l, N, [=](int n) { return lab[n] == cidx ? lwt[n] : 0.0; }, vp);
        }
        if (lane() == 0) {
            double* o = sh.dbuf + 4 * cidx;
            o[3] = 0.0;
            if (cnt >= 3) {
                if (vp[2] < 0) { vp[0] = -vp[0]; vp[1] = -vp[1]; vp[2] = -vp[2]; }   // :599-600
                o[0] = vp[0]; o[1] = vp[1]; o[2] = vp[2]; o[3] = 1.0;
            }
        }
    }
    block_sync();
    if (tid() == 0) {
        double* v0 = sh.dbuf;
        double* v1 = sh.dbuf + 4;
        bool too_similar = true;                              // :604-615
        if (v0[3] != 0.0 && v1[3] != 0.0) {
            double cphi = clip(dot3(v0[0], v0[1], v0[2], v1[0], v1[1], v1[2]), -1.0, 1.0);
            double ang = fabs(acos(clip(fabs(cphi), -1.0, 1.0)));
            if (ang > c.prm.merge_thresh) too_similar = false;
        }
        if (!too_similar) {                                   // :617-628 (both clusters valid here)
            sh.cur[3 * worst] = v0[0]; sh.cur[3 * worst + 1] = v0[1]; sh.cur[3 * worst + 2] = v0[2];
            sh.s[worst] = stdd;
            if (sh.M < MAXM && sh.M < c.mcap) {               // the [vp][line] scratch has mcap rows
                int m = sh.M;
                sh.cur[3 * m] = v1[0]; sh.cur[3 * m + 1] = v1[1]; sh.cur[3 * m + 2] = v1[2];
                sh.nxt[3 * m] = 0; sh.nxt[3 * m + 1] = 0; sh.nxt[3 * m + 2] = 0;
                sh.s[m] = stdd;
                sh.M = m + 1;
            } else {
                sh.flags |= VPK_EM_FLAG_VP_OVERFLOW;
            }
        }
    }
    if (tid() == 0) sh.dbuf[13] += lap(tq_);
    block_sync();
}

// ---------------------------------------------------------------------------------------------
// outputs
// ---------------------------------------------------------------------------------------------
struct EmOut {
    double* vp;       // max_vp x 3
    double* sigma;    // max_vp
    double* counts;   // max_vp
    double* counts_w; // max_vp
    int* num_vp;
    long long* assoc; // N
    int* iterations;
    int* status;
    unsigned* flags;
    double* metric;   // N x max_vp or null
    double* trace;    // (num_iter + 1) x TRACE_COLS or null
    int max_vp;
    double* dbg = nullptr;   // test hook: per iteration [M, s[0..MAXM), cur[0..3 MAXM)] before the E-step
    // EM_result['distribution'] (vpk_em_set_distribution_out), all null or all set
    double* d_pv = nullptr;      // max_vp
    double* d_angles = nullptr;  // max_vp x 2
    double* d_pl = nullptr;      // N
    double* d_plv = nullptr;     // N x max_vp
    double* d_pvl = nullptr;     // N x max_vp
    double* d_lvsq = nullptr;    // N x max_vp
};

VPK_DEVFN void write_result(EmCtx& c, EmOut& o, int status, int iterations) {
    Shared& sh = SH();
    const int N = c.N;
    int M = status == VPK_EM_OK ? sh.M : 0;
    if (M > o.max_vp) {
        M = o.max_vp;
        if (tid() == 0) sh.flags |= VPK_EM_FLAG_VP_OVERFLOW;
    }
    for (int m = tid(); m < o.max_vp; m += nthreads()) {
        bool ok = m < M;
        for (int d = 0; d < 3; ++d) o.vp[3 * m + d] = ok ? sh.nxt[3 * m + d] : 0.0;
        o.sigma[m] = ok ? sh.s[m] : 0.0;
        o.counts[m] = ok ? sh.cnt[m] : 0.0;
        o.counts_w[m] = ok ? sh.cntw[m] : 0.0;
    }
    for (int n = tid(); n < N; n += nthreads()) {
        int a = status == VPK_EM_OK ? c.assoc[n] : -1;
        o.assoc[n] = (a >= M) ? -1 : a;
        if (o.metric)
            for (int m = 0; m < o.max_vp; ++m)
                o.metric[(size_t)n * o.max_vp + m] = m < M ? c.w[(size_t)m * c.ldn + n] : 0.0;
    }
    if (o.d_pv) {
        // The PDF of the last calc_probabilities call (vp_localisation.py:415/:430 -> :441): lvsq and p_vl are where the last
        // E-step left them, p_lv and p_l are re-evaluated from lvsq with the E-step's expressions (it keeps their
        // product with p_v only), the angles from the VPs with the prior's (probability_functions.py:252-259).
        for (int m = tid(); m < o.max_vp; m += nthreads()) {
            const bool ok = m < M;
            double alpha = 0.0, beta = 0.0;
            if (ok) {
                const double x0 = sh.nxt[3 * m], x1 = sh.nxt[3 * m + 1];
                beta = asin(x1);
                double inner = x0 / cos(beta);
                inner = inner < 1 ? inner : (is_nan(inner) ? inner : 1.0);
                inner = inner > -1 ? inner : (is_nan(inner) ? inner : -1.0);
                alpha = asin(inner);
            }
            o.d_pv[m] = ok ? sh.pv[m] : 0.0;
            o.d_angles[2 * m] = alpha;
            o.d_angles[2 * m + 1] = beta;
        }
        for (int n = tid(); n < N; n += nthreads()) {
            double pl = 0.0;
            for (int m = 0; m < o.max_vp; ++m) {
                const bool ok = m < M;
                const double lv = ok ? c.lvsq[(size_t)m * c.ldn + n] : 0.0;
                const double plv = ok ? exp_underflow(-(lv / (2 * sh.s[m]))) * sh.k2[m] : 0.0;   // calc_plv :137-145
                if (ok) pl += plv * sh.pv[m];
                o.d_lvsq[(size_t)n * o.max_vp + m] = lv;
                o.d_plv[(size_t)n * o.max_vp + m] = plv;
                o.d_pvl[(size_t)n * o.max_vp + m] = ok ? c.pvl[(size_t)m * c.ldn + n] : 0.0;
            }
            o.d_pl[n] = (pl > 1e-12 || is_nan(pl)) ? pl : 1e-12;                                   // :116-117
        }
    }
    block_sync();
    if (tid() == 0) {
        *o.num_vp = M;
        *o.iterations = iterations;
        *o.status = status;
        *o.flags = sh.flags;
    }
    block_sync();
}

VPK_DEV void trace_put(EmOut& o, int i, int slot, double v) {
    if (o.trace && tid() == 0) o.trace[TRACE_COLS * i + slot] = v;
}
VPK_DEV void trace_add(EmOut& o, int i, int slot, double v) {
    if (o.trace && tid() == 0) o.trace[TRACE_COLS * i + slot] += v;
}


// ---------------------------------------------------------------------------------------------
// the driver: expectation_maximisation (vp_localisation.py:168-450)
// ---------------------------------------------------------------------------------------------
// Time slicing.  A launch may carry a deadline: an image that is still iterating when it passes is SUSPENDED
// at the top of its next iteration -- the only state that lives outside the slot's HBM scratch at that point
// is the Shared block in LDS, which is copied into the slot -- and resumed by a later launch (any workgroup)
// at exactly that point.  The arithmetic does not depend on where an image was suspended: results are
// bit-identical to an uninterrupted run.  Why: the EM of a never-converging image takes 99 iterations (~20 ms)
// against ~5 ms for the average one, and a launch that must run every image to completion holds its CUs for
// the slowest image.
constexpr int EM_DONE = 0, EM_SUSPENDED = 1;
constexpr long long EM_NO_DEADLINE = 0x7fffffffffffffffll;
struct EmSlice {
    long long deadline;   // clock_ticks() value; EM_NO_DEADLINE = run to completion
    int start_iter;       // in: -1 = fresh image, i >= 0 = resume at the top of iteration i; out: where it was suspended
};

VPK_DEVFN void save_state(EmCtx& c) {
    typedef VPK_GLOBAL unsigned long long* gup;
    gup dst = (gup)c.state;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&SH());
    for (int q = tid(); q < (int)(sizeof(Shared) / 8); q += nthreads()) dst[q] = src[q];
    block_sync();
}
VPK_DEVFN void restore_state(EmCtx& c) {
    typedef const VPK_GLOBAL unsigned long long* cgup;
    cgup src = (cgup)c.state;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&SH());
    block_sync();
    for (int q = tid(); q < (int)(sizeof(Shared) / 8); q += nthreads()) dst[q] = src[q];
    block_sync();
}

VPK_DEVFN int em_run(EmCtx& c, EmOut& o, EmSlice& sl) {
    Shared& sh = SH();
    const vpk_em_params& P = c.prm;
    const double max_stdd = 1e-6;                             // :196-198 ("angle")
    const double merge_thresh_final = P.merge_thresh * 10;    // :190
    const int split_merge_it = 100;                           // :193
    long long tk = clock_ticks();
    const long long t_begin = tk;
    int first = 0;
    if (sl.start_iter >= 0) {
        restore_state(c);
        first = sl.start_iter;
    } else {
    if (tid() == 0) { sh.flags = 0; sh.M = 0; sh.ncomp = 0; sh.ibuf[5] = 0; sh.ibuf[2] = 0; sh.active_us = 0; for (int q = 8; q < 16; ++q) sh.dbuf[q] = 0; }
    block_sync();
    if (o.trace)
        for (int q = tid(); q < TRACE_COLS * (P.num_iter + 1); q += nthreads()) o.trace[q] = 0.0;
    if (c.N <= 0) { write_result(c, o, VPK_EM_NO_VP, 0); return EM_DONE; }

    if (P.use_weights) { pairwise_setup(c, true); zero_tail_rows(c); }   // :177-178 (+ :230 kNN score)
    else pairwise_setup(c, false);                            // only lines_angles is needed
    trace_put(o, P.num_iter, 0, lap(tk));                     // last trace row: setup timings
    normalise_lines(c);                                       // :185-186, :226 (the caller's array, in place)
    for (int q = tid(); q < 3 * c.N; q += nthreads()) c.lcopy[q] = c.l[q];
    for (int q = tid(); q < 4 * c.N; q += nthreads()) c.lpcopy[q] = c.lp[q];
    block_sync();
    c.l = c.lcopy;                                            // from here on the image lives in its slot only
    c.lp = c.lpcopy;
    initial_vps(c);                                           // :208
    const int m_found = sh.M;
    prior_setup(c);                                           // :210
    if (m_found == 0) { write_result(c, o, VPK_EM_NO_INITIAL_VP, 0); return EM_DONE; }   // ValueError at :165
    if (c.init_vp) {                                          // :212-215
        if (tid() == 0) {
            int m = c.n_init < MAXM ? c.n_init : MAXM;
            for (int k = 0; k < m; ++k) {
                cgdp q = c.init_vp + 3 * (size_t)k;
                double nr = norm3(q[0], q[1], q[2]);
                sh.cur[3 * k] = q[0] / nr; sh.cur[3 * k + 1] = q[1] / nr; sh.cur[3 * k + 2] = q[2] / nr;
            }
            sh.M = m;
        }
        block_sync();
    }
    weights_setup(c);                                         // :227-235
    line_geometry_setup(c);
    for (int m = tid(); m < MAXM; m += nthreads()) {
        sh.s[m] = 1.0 * (sh.sigma_prior * 1e-6);              // :219,:239
        sh.nxt[3 * m] = 0; sh.nxt[3 * m + 1] = 0; sh.nxt[3 * m + 2] = 0;
    }
    block_sync();

    estep(c, sh.cur);                                         // :245
    smooth(c);                                                // :246
    assign_lines(c, true);                                    // :247
    count_lines(c);
    for (int m = tid(); m < sh.M; m += nthreads()) sh.removed[m] = sh.cnt[m] < 3;   // :250-251
    block_sync();
    compact_vps(c);
    trace_put(o, P.num_iter, 1, lap(tk));
    }

    for (int i = first; i < P.num_iter; ++i) {
        if (sl.deadline != EM_NO_DEADLINE && i > sl.start_iter) {   // checkpoint (at least one iteration per slice)
            if (tid() == 0) sh.ibuf[6] = clock_ticks() >= sl.deadline;
            block_sync();
            if (sh.ibuf[6]) {
                if (tid() == 0) sh.active_us += (double)(clock_ticks() - t_begin) * CLOCK_US;
                block_sync();
                save_state(c);
                sl.start_iter = i;
                return EM_SUSPENDED;
            }
        }
        tk = clock_ticks();
        const long long t_iter = tk;
        if (sh.M == 0) { write_result(c, o, VPK_EM_NO_VP, 0); return EM_DONE; }     // :258-260
        double events = 0;
        if (i % P.split_merge_freq == 0 && i > 0 && i < split_merge_it && P.do_split) {   // :262-269
            int mb = sh.M;
            if (tid() == 0) { sh.dbuf[11] = 0; sh.dbuf[12] = 0; sh.dbuf[13] = 0; }
            estep(c, sh.cur);
            smooth(c);
            split_vp(c);
            if (sh.M != mb) events += 1;
            trace_put(o, i, 8, sh.dbuf[11]);
            trace_put(o, i, 9, sh.dbuf[12]);
            trace_put(o, i, 10, sh.dbuf[13]);
        }
        if (o.dbg && tid() == 0) {
            double* q = o.dbg + (size_t)i * (1 + 4 * MAXM);
            q[0] = sh.M;
            for (int m = 0; m < MAXM; ++m) q[1 + m] = sh.s[m];
            for (int m = 0; m < 3 * MAXM; ++m) q[1 + MAXM + m] = sh.cur[m];
        }
        lap(tk);
        estep(c, sh.cur);                                     // :273
        trace_put(o, i, 4, lap(tk));
        smooth(c);                                            // :282
        trace_put(o, i, 5, lap(tk));
        double max_err = 0.0;
        if (P.do_iterations) {
            mstep(c, 0, max_stdd);                            // :284-322
            trace_put(o, i, 6, lap(tk));
            max_err = max_err_of(sh, sh.M);
            block_sync();
            compact_vps(c);                                   // :329-331
        } else {
            for (int q = tid(); q < 3 * sh.M; q += nthreads()) sh.nxt[q] = sh.cur[q];   // :324-325
            block_sync();
        }
        trace_put(o, i, 0, (double)sh.M);
        trace_put(o, i, 1, max_err);
        // (:332 recomputes and discards an E-step; its only side effect, the floor of s at
        //  1e-200, cannot change s after the clamp at :307)

        if (max_err < P.final_convergence || i == P.num_iter - 1 || !P.do_iterations) {   // :335
            if (P.do_merge) merge_vps(c, true, merge_thresh_final);                       // :339
            trace_put(o, P.num_iter, 3, (double)sh.M);        // finalisation audit trail: M after merge
            if (sh.M == 0) { write_result(c, o, VPK_EM_NO_VP, i); return EM_DONE; }   // reference: argmax of empty (:349)
            estep(c, sh.cur);                                 // :344 (stale index i)
            smooth(c);                                        // :346
            assign_lines(c, false);                           // :349
            mstep(c, 1, max_stdd);                            // :353-392
            compact_vps(c);                                   // :394-396
            trace_put(o, P.num_iter, 4, (double)sh.M);        // ... after the hard-assignment M-step
            estep(c, sh.cur);                                 // :398 (still index i)
            smooth(c);                                        // :400
            if (sh.M == 0) { write_result(c, o, VPK_EM_NO_VP, 0); return EM_DONE; }       // :402-404
            assign_lines(c, false);                           // :406
            for (int m = tid(); m < sh.M; m += nthreads()) sh.icnt[m] = 0;
            block_sync();
            for (int n = tid(); n < c.N; n += nthreads()) sh.icnt[c.assoc[n]] = 1;        // np.unique (:408)
            block_sync();
            for (int m = tid(); m < sh.M; m += nthreads()) sh.removed[m] = !sh.icnt[m];
            block_sync();
            compact_vps(c);                                   // :412-413
            trace_put(o, P.num_iter, 5, (double)sh.M);        // ... after keeping the VPs that win a line
            estep(c, sh.nxt);                                 // :415 (index i+1 at last)
            smooth(c);                                        // :417
            assign_lines(c, true);                            // :418
            count_lines(c);
            // :423-437.  The reference's scan does NOT start over after a removal: `vidx` stays where it is (the next VP has
            // moved into that index), so a VP in front of it whose count drops below num_min_lines through the re-assignment
            // that follows a removal is never looked at again and survives with fewer lines (configs[3] image 558: a VP with
            // 2 lines in the reference's result)
            int vscan = 0;
            for (int guard = 0; guard < MAXM + 1; ++guard) {
                int vidx = -1;
                for (int m = vscan; m < sh.M; ++m)
                    if (sh.cnt[m] < P.num_min_lines) { vidx = m; break; }
                block_sync();
                if (vidx < 0) break;
                vscan = vidx;
                for (int m = tid(); m < sh.M; m += nthreads()) sh.removed[m] = (m == vidx);
                block_sync();
                compact_vps(c);
                estep(c, sh.nxt);
                smooth(c);
                assign_lines(c, true);
                count_lines(c);
            }
            trace_put(o, i, 2, (double)sh.M);
            trace_put(o, i, 3, events + 2);
            trace_put(o, i, 7, (double)(clock_ticks() - t_iter) * CLOCK_US);
            trace_put(o, P.num_iter, 2, sh.active_us + (double)(clock_ticks() - t_begin) * CLOCK_US);
            trace_put(o, P.num_iter, 6, sh.dbuf[8] + sh.dbuf[10]);   // smoother: operand staging + partial reduction
            trace_put(o, P.num_iter, 7, sh.dbuf[9]);                  // smoother: main loop (wave 0)
            trace_put(o, P.num_iter, 8, sh.dbuf[14]);                 // E-step: prior part
            trace_put(o, P.num_iter, 9, sh.dbuf[15]);                 // E-step: line part
            write_result(c, o, VPK_EM_OK, i);                 // :439-442
            return EM_DONE;
        }
        if (i % P.split_merge_freq == 0 && i > 0 && i <= split_merge_it + P.split_merge_freq && P.do_merge) {
            int mb = sh.M;
            lap(tk);
            merge_vps(c, true, P.merge_thresh);               // :444-448
            trace_put(o, i, 11, lap(tk));
            if (sh.M != mb) events += 4;
        }
        trace_put(o, i, 2, (double)sh.M);
        trace_put(o, i, 3, events);
        trace_put(o, i, 7, (double)(clock_ticks() - t_iter) * CLOCK_US);
        for (int q = tid(); q < 3 * MAXM; q += nthreads()) {  // v[i+1] becomes v[i]; v[i+2] is zeros
            sh.cur[q] = sh.nxt[q];
            sh.nxt[q] = 0.0;
        }
        block_sync();
    }
    write_result(c, o, VPK_EM_NO_VP, 0);                      // :450
    return EM_DONE;
}

}  // namespace vpk
#endif
