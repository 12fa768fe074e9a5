// vpk_lsd.cpp -- line segment detector of the front end (host code; C-ABI entry vpk_lsd_detect).
//
// replaces: lsdpython.lsd.detect_line_segments(image) as called from detect_lsd_lines (evaluation.py:227-251).
// The reference's detector lives in an un-vendored submodule (.gitmodules:1-3 -> github.com/fkluger/lsd-python, a
// Cython wrapper around R. Grompone von Gioi's LSD 1.6; the directory is empty in /root/reference), so there is no
// source to follow and no output to pin against: PARITY UNPINNED.  This file restates the PUBLISHED algorithm --
// Grompone von Gioi, Jakubowicz, Morel, Randall, "LSD: a Line Segment Detector", Image Processing On Line 2 (2012),
// with that paper's default parameters (scale 0.8, sigma_scale 0.6, quant 2.0, ang_th 22.5 deg, log_eps 0,
// density_th 0.7, n_bins 1024) -- step by step:
//   1. Gaussian sub-sampling to 80 %,                      2. gradient with a 2 x 2 mask, level-line angle, magnitude,
//   3. pixels pseudo-ordered by magnitude (1024 bins),     4. region growing with the region's running angle,
//   5. rectangle from the region's weighted inertia,       6. density check / refinement (angle tolerance, radius),
//   7. a-contrario validation: NFA = N_tests * binomial tail, kept if -log10(NFA) > 0, with the rectangle variations
//      of the paper's "rect_improve".
// Output rows: x1, y1, x2, y2, width, p, -log10(NFA) in pixel coordinates of the input image (7 doubles), the layout
// detect_lsd_lines consumes (columns 0..3 and 6).  The detector is sequential by construction (region growing follows
// the gradient order), exactly like the reference's C implementation; it is not on the hot path of any BASELINE
// config (all of them take LSD lines as given).  tests/test_frontend.py checks the contract the paper states:
// synthetic segments are recovered to sub-pixel accuracy and white noise yields (almost) no detection.
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "../../include/vpk.h"

namespace {

constexpr double NOTDEF = -1024.0;
constexpr double PI_L = 3.14159265358979323846;
constexpr double M_3_2_PI_L = 4.71238898038;
constexpr double M_2__PI_L = 6.28318530718;
constexpr double LN10_L = 2.30258509299404568402;

struct Pt { int x, y; };
struct Rect {
    double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p;
};

double dist(double x1, double y1, double x2, double y2) { return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }

bool double_equal(double a, double b) {
    if (a == b) return true;
    const double diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
    double mx = aa > bb ? aa : bb;
    if (mx < DBL_MIN) mx = DBL_MIN;
    return diff / mx <= 100.0 * DBL_EPSILON;
}

// ---- 1. Gaussian sub-sampling ---------------------------------------------------------------------------------
void gaussian_kernel(std::vector<double>& k, double sigma, double mean) {
    double sum = 0.0;
    for (size_t i = 0; i < k.size(); ++i) {
        const double v = ((double)i - mean) / sigma;
        k[i] = exp(-0.5 * v * v);
        sum += k[i];
    }
    if (sum >= 0.0)
        for (double& v : k) v /= sum;
}

void gaussian_sampler(const double* in, int xs, int ys, double scale, double sigma_scale, std::vector<double>& out, int& N,
                      int& M) {
    N = (int)ceil(xs * scale);
    M = (int)ceil(ys * scale);
    std::vector<double> aux((size_t)N * ys);
    out.assign((size_t)N * M, 0.0);
    const double sigma = scale < 1.0 ? sigma_scale / scale : sigma_scale;
    const int h = (int)ceil(sigma * sqrt(2.0 * 3.0 * log(10.0)));
    const int n = 1 + 2 * h;
    std::vector<double> kernel(n);
    const int dxs = 2 * xs, dys = 2 * ys;
    for (int x = 0; x < N; ++x) {
        const double xx = (double)x / scale;
        const int xc = (int)floor(xx + 0.5);
        gaussian_kernel(kernel, sigma, (double)h + xx - (double)xc);
        for (int y = 0; y < ys; ++y) {
            double sum = 0.0;
            for (int i = 0; i < n; ++i) {
                int j = xc - h + i;
                while (j < 0) j += dxs;
                while (j >= dxs) j -= dxs;
                if (j >= xs) j = dxs - 1 - j;                  // symmetric boundary
                sum += in[(size_t)y * xs + j] * kernel[i];
            }
            aux[(size_t)y * N + x] = sum;
        }
    }
    for (int y = 0; y < M; ++y) {
        const double yy = (double)y / scale;
        const int yc = (int)floor(yy + 0.5);
        gaussian_kernel(kernel, sigma, (double)h + yy - (double)yc);
        for (int x = 0; x < N; ++x) {
            double sum = 0.0;
            for (int i = 0; i < n; ++i) {
                int j = yc - h + i;
                while (j < 0) j += dys;
                while (j >= dys) j -= dys;
                if (j >= ys) j = dys - 1 - j;
                sum += aux[(size_t)j * N + x] * kernel[i];
            }
            out[(size_t)y * N + x] = sum;
        }
    }
}

// ---- 2./3. gradient, level-line angle, pseudo-ordering -----------------------------------------------------------
void ll_angle(const std::vector<double>& img, int p, int n, double threshold, int n_bins, std::vector<double>& angles,
              std::vector<double>& modgrad, std::vector<Pt>& order) {
    angles.assign((size_t)p * n, NOTDEF);
    modgrad.assign((size_t)p * n, 0.0);
    double max_grad = 0.0;
    for (int x = 0; x < p - 1; ++x)
        for (int y = 0; y < n - 1; ++y) {
            const size_t adr = (size_t)y * p + x;
            const double com1 = img[adr + p + 1] - img[adr];
            const double com2 = img[adr + 1] - img[adr + p];
            const double gx = com1 + com2, gy = com1 - com2;
            const double norm = sqrt((gx * gx + gy * gy) / 4.0);
            modgrad[adr] = norm;
            if (norm <= threshold) {
                angles[adr] = NOTDEF;
            } else {
                angles[adr] = atan2(gx, -gy);                  // level-line angle
                if (norm > max_grad) max_grad = norm;
            }
        }
    std::vector<std::vector<Pt>> bins(n_bins);
    if (max_grad > 0.0)
        for (int x = 0; x < p - 1; ++x)
            for (int y = 0; y < n - 1; ++y) {
                const double norm = modgrad[(size_t)y * p + x];
                int i = (int)(norm * (double)n_bins / max_grad);
                if (i >= n_bins) i = n_bins - 1;
                bins[i].push_back(Pt{x, y});
            }
    order.clear();
    for (int i = n_bins - 1; i >= 0; --i) order.insert(order.end(), bins[i].begin(), bins[i].end());
}

bool isaligned(int x, int y, const std::vector<double>& angles, int xs, double theta, double prec) {
    const double a = angles[(size_t)y * xs + x];
    if (a == NOTDEF) return false;
    theta -= a;
    if (theta < 0.0) theta = -theta;
    if (theta > M_3_2_PI_L) {
        theta -= M_2__PI_L;
        if (theta < 0.0) theta = -theta;
    }
    return theta <= prec;
}

double angle_diff_signed(double a, double b) {
    a -= b;
    while (a <= -PI_L) a += M_2__PI_L;
    while (a > PI_L) a -= M_2__PI_L;
    return a;
}
double angle_diff(double a, double b) { return fabs(angle_diff_signed(a, b)); }

// ---- 7. NFA ---------------------------------------------------------------------------------------------------------
double log_gamma_lanczos(double x) {
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424,
                                2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
    double b = 0.0;
    for (int n = 0; n < 7; ++n) {
        a -= log(x + (double)n);
        b += q[n] * pow(x, (double)n);
    }
    return a + log(b);
}
double log_gamma_windschitl(double x) {
    return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
}
double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

double nfa(int n, int k, double p, double logNT) {
    const double tolerance = 0.1;
    if (n == 0 || k == 0) return -logNT;
    if (n == k) return -logNT - (double)n * log10(p);
    const double p_term = p / (1.0 - p);
    const double log1term = log_gamma((double)n + 1.0) - log_gamma((double)k + 1.0) - log_gamma((double)(n - k) + 1.0) +
                            (double)k * log(p) + (double)(n - k) * log(1.0 - p);
    double term = exp(log1term);
    if (double_equal(term, 0.0)) {
        if ((double)k > (double)n * p) return -log1term / LN10_L - logNT;
        return -logNT;
    }
    double bin_tail = term;
    for (int i = k + 1; i <= n; ++i) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1.0) {
            const double err = term * ((1.0 - pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
            if (err < tolerance * fabs(-log10(bin_tail) - logNT) * bin_tail) break;
        }
    }
    return -log10(bin_tail) - logNT;
}

// rectangle pixel iterator (column by column between the lower and upper edges)
struct RectIter {
    double vx[4], vy[4], ys, ye;
    int x, y;
};
double inter_low(double x, double x1, double y1, double x2, double y2) {
    if (double_equal(x1, x2) && y1 < y2) return y1;
    if (double_equal(x1, x2) && y1 > y2) return y2;
    return y1 + (x - x1) * (y2 - y1) / (x2 - x1);
}
double inter_hi(double x, double x1, double y1, double x2, double y2) {
    if (double_equal(x1, x2) && y1 < y2) return y2;
    if (double_equal(x1, x2) && y1 > y2) return y1;
    return y1 + (x - x1) * (y2 - y1) / (x2 - x1);
}
bool ri_end(const RectIter& i) { return (double)i.x > i.vx[2]; }
void ri_inc(RectIter& i) {
    if (!ri_end(i)) i.y++;
    while ((double)i.y > i.ye && !ri_end(i)) {
        i.x++;
        if (ri_end(i)) return;
        if ((double)i.x < i.vx[3]) i.ys = inter_low((double)i.x, i.vx[0], i.vy[0], i.vx[3], i.vy[3]);
        else i.ys = inter_low((double)i.x, i.vx[3], i.vy[3], i.vx[2], i.vy[2]);
        if ((double)i.x < i.vx[1]) i.ye = inter_hi((double)i.x, i.vx[0], i.vy[0], i.vx[1], i.vy[1]);
        else i.ye = inter_hi((double)i.x, i.vx[1], i.vy[1], i.vx[2], i.vy[2]);
        i.y = (int)ceil(i.ys);
    }
}
RectIter ri_ini(const Rect& r) {
    double vx[4], vy[4];
    vx[0] = r.x1 - r.dy * r.width / 2.0; vy[0] = r.y1 + r.dx * r.width / 2.0;
    vx[1] = r.x2 - r.dy * r.width / 2.0; vy[1] = r.y2 + r.dx * r.width / 2.0;
    vx[2] = r.x2 + r.dy * r.width / 2.0; vy[2] = r.y2 - r.dx * r.width / 2.0;
    vx[3] = r.x1 + r.dy * r.width / 2.0; vy[3] = r.y1 - r.dx * r.width / 2.0;
    int offset;
    if (r.x1 < r.x2 && r.y1 <= r.y2) offset = 0;
    else if (r.x1 >= r.x2 && r.y1 < r.y2) offset = 1;
    else if (r.x1 > r.x2 && r.y1 >= r.y2) offset = 2;
    else offset = 3;
    RectIter i;
    for (int n = 0; n < 4; ++n) {
        i.vx[n] = vx[(offset + n) % 4];
        i.vy[n] = vy[(offset + n) % 4];
    }
    i.x = (int)ceil(i.vx[0]) - 1;
    i.y = (int)ceil(i.vy[0]);
    i.ys = i.ye = -DBL_MAX;
    ri_inc(i);
    return i;
}

double rect_nfa(const Rect& rec, const std::vector<double>& angles, int xs, int ys, double logNT) {
    int pts = 0, alg = 0;
    for (RectIter i = ri_ini(rec); !ri_end(i); ri_inc(i))
        if (i.x >= 0 && i.y >= 0 && i.x < xs && i.y < ys) {
            ++pts;
            if (isaligned(i.x, i.y, angles, xs, rec.theta, rec.prec)) ++alg;
        }
    return nfa(pts, alg, rec.p, logNT);
}

// ---- 4. region growing ------------------------------------------------------------------------------------------------
void region_grow(int x, int y, const std::vector<double>& angles, int xs, int ys, std::vector<Pt>& reg, int& reg_size,
                 double& reg_angle, std::vector<unsigned char>& used, double prec) {
    reg_size = 1;
    reg[0] = Pt{x, y};
    reg_angle = angles[(size_t)y * xs + x];
    double sumdx = cos(reg_angle), sumdy = sin(reg_angle);
    used[(size_t)y * xs + x] = 1;
    for (int i = 0; i < reg_size; ++i)
        for (int xx = reg[i].x - 1; xx <= reg[i].x + 1; ++xx)
            for (int yy = reg[i].y - 1; yy <= reg[i].y + 1; ++yy)
                if (xx >= 0 && yy >= 0 && xx < xs && yy < ys && used[(size_t)yy * xs + xx] != 1 &&
                    isaligned(xx, yy, angles, xs, reg_angle, prec)) {
                    used[(size_t)yy * xs + xx] = 1;
                    reg[reg_size] = Pt{xx, yy};
                    ++reg_size;
                    const double a = angles[(size_t)yy * xs + xx];
                    sumdx += cos(a);
                    sumdy += sin(a);
                    reg_angle = atan2(sumdy, sumdx);
                }
}

// ---- 5. rectangle approximation -----------------------------------------------------------------------------------
double get_theta(const std::vector<Pt>& reg, int reg_size, double x, double y, const std::vector<double>& modgrad, int xs,
                 double reg_angle, double prec) {
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (int i = 0; i < reg_size; ++i) {
        const double w = modgrad[(size_t)reg[i].y * xs + reg[i].x];
        Ixx += ((double)reg[i].y - y) * ((double)reg[i].y - y) * w;
        Iyy += ((double)reg[i].x - x) * ((double)reg[i].x - x) * w;
        Ixy -= ((double)reg[i].x - x) * ((double)reg[i].y - y) * w;
    }
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = fabs(Ixx) > fabs(Iyy) ? atan2(lambda - Ixx, Ixy) : atan2(Ixy, lambda - Iyy);
    if (angle_diff(theta, reg_angle) > prec) theta += PI_L;
    return theta;
}

void region2rect(const std::vector<Pt>& reg, int reg_size, const std::vector<double>& modgrad, int xs, double reg_angle,
                 double prec, double p, Rect& rec) {
    double x = 0.0, y = 0.0, sum = 0.0;
    for (int i = 0; i < reg_size; ++i) {
        const double w = modgrad[(size_t)reg[i].y * xs + reg[i].x];
        x += (double)reg[i].x * w;
        y += (double)reg[i].y * w;
        sum += w;
    }
    x /= sum;
    y /= sum;
    const double theta = get_theta(reg, reg_size, x, y, modgrad, xs, reg_angle, prec);
    const double dx = cos(theta), dy = sin(theta);
    double l_min = 0.0, l_max = 0.0, w_min = 0.0, w_max = 0.0;
    for (int i = 0; i < reg_size; ++i) {
        const double l = ((double)reg[i].x - x) * dx + ((double)reg[i].y - y) * dy;
        const double w = -((double)reg[i].x - x) * dy + ((double)reg[i].y - y) * dx;
        if (l > l_max) l_max = l;
        if (l < l_min) l_min = l;
        if (w > w_max) w_max = w;
        if (w < w_min) w_min = w;
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}

// ---- 6. refinement ----------------------------------------------------------------------------------------------------
struct Ctx {
    const std::vector<double>& angles;
    const std::vector<double>& modgrad;
    std::vector<unsigned char>& used;
    int xs, ys;
};

bool reduce_region_radius(Ctx& c, std::vector<Pt>& reg, int& reg_size, double reg_angle, double prec, double p, Rect& rec,
                          double density_th) {
    double density = (double)reg_size / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    const double xc = (double)reg[0].x, yc = (double)reg[0].y;
    const double rad1 = dist(xc, yc, rec.x1, rec.y1), rad2 = dist(xc, yc, rec.x2, rec.y2);
    double rad = rad1 > rad2 ? rad1 : rad2;
    while (density < density_th) {
        rad *= 0.75;
        for (int i = 0; i < reg_size; ++i)
            if (dist(xc, yc, (double)reg[i].x, (double)reg[i].y) > rad) {
                c.used[(size_t)reg[i].y * c.xs + reg[i].x] = 0;
                reg[i] = reg[reg_size - 1];
                --reg_size;
                --i;
            }
        if (reg_size < 2) return false;
        region2rect(reg, reg_size, c.modgrad, c.xs, reg_angle, prec, p, rec);
        density = (double)reg_size / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
}

bool refine(Ctx& c, std::vector<Pt>& reg, int& reg_size, double reg_angle, double prec, double p, Rect& rec,
            double density_th) {
    double density = (double)reg_size / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    const double xc = (double)reg[0].x, yc = (double)reg[0].y;
    const double ang_c = c.angles[(size_t)reg[0].y * c.xs + reg[0].x];
    double sum = 0.0, s_sum = 0.0;
    int n = 0;
    for (int i = 0; i < reg_size; ++i) {
        c.used[(size_t)reg[i].y * c.xs + reg[i].x] = 0;
        if (dist(xc, yc, (double)reg[i].x, (double)reg[i].y) < rec.width) {
            const double ang_d = angle_diff_signed(c.angles[(size_t)reg[i].y * c.xs + reg[i].x], ang_c);
            sum += ang_d;
            s_sum += ang_d * ang_d;
            ++n;
        }
    }
    const double mean_angle = sum / (double)n;
    const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
    region_grow(reg[0].x, reg[0].y, c.angles, c.xs, c.ys, reg, reg_size, reg_angle, c.used, tau);
    if (reg_size < 2) return false;
    region2rect(reg, reg_size, c.modgrad, c.xs, reg_angle, prec, p, rec);
    density = (double)reg_size / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < density_th) return reduce_region_radius(c, reg, reg_size, reg_angle, prec, p, rec, density_th);
    return true;
}

double rect_improve(Rect& rec, const std::vector<double>& angles, int xs, int ys, double logNT, double log_eps) {
    const double delta = 0.5, delta_2 = delta / 2.0;
    double log_nfa = rect_nfa(rec, angles, xs, ys, logNT);
    if (log_nfa > log_eps) return log_nfa;
    Rect r = rec;                                              // finer precisions
    for (int n = 0; n < 5; ++n) {
        r.p /= 2.0;
        r.prec = r.p * PI_L;
        const double v = rect_nfa(r, angles, xs, ys, logNT);
        if (v > log_nfa) { log_nfa = v; rec = r; }
    }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;                                                   // narrower
    for (int n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.width -= delta;
            const double v = rect_nfa(r, angles, xs, ys, logNT);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;                                                   // one side of the rectangle
    for (int n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
            r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
            r.width -= delta;
            const double v = rect_nfa(r, angles, xs, ys, logNT);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;                                                   // the other side
    for (int n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
            r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
            r.width -= delta;
            const double v = rect_nfa(r, angles, xs, ys, logNT);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    if (log_nfa > log_eps) return log_nfa;
    r = rec;                                                   // even finer precisions
    for (int n = 0; n < 5; ++n) {
        r.p /= 2.0;
        r.prec = r.p * PI_L;
        const double v = rect_nfa(r, angles, xs, ys, logNT);
        if (v > log_nfa) { log_nfa = v; rec = r; }
    }
    return log_nfa;
}

}  // namespace

extern "C" int vpk_lsd_detect(const double* image, int width, int height, double scale, double* out, int max_segments,
                              int* n_out) {
    if (!image || width < 8 || height < 8 || !n_out || max_segments < 0 || (max_segments > 0 && !out) || !(scale > 0.0))
        return VPK_ERR_ARG;
    const double sigma_scale = 0.6, quant = 2.0, ang_th = 22.5, log_eps = 0.0, density_th = 0.7;
    const int n_bins = 1024;
    const double prec = PI_L * ang_th / 180.0, p = ang_th / 180.0, rho = quant / sin(prec);
    std::vector<double> scaled;
    int xs = width, ys = height;
    if (scale != 1.0) gaussian_sampler(image, width, height, scale, sigma_scale, scaled, xs, ys);
    else scaled.assign(image, image + (size_t)width * height);
    std::vector<double> angles, modgrad;
    std::vector<Pt> order;
    ll_angle(scaled, xs, ys, rho, n_bins, angles, modgrad, order);
    const double logNT = 5.0 * (log10((double)xs) + log10((double)ys)) / 2.0 + log10(11.0);
    const int min_reg_size = (int)(-logNT / log10(p));
    std::vector<unsigned char> used((size_t)xs * ys, 0);
    std::vector<Pt> reg((size_t)xs * ys);
    Ctx ctx{angles, modgrad, used, xs, ys};
    int count = 0;
    for (const Pt& s : order) {
        if (used[(size_t)s.y * xs + s.x] != 0 || angles[(size_t)s.y * xs + s.x] == NOTDEF) continue;
        int reg_size = 0;
        double reg_angle = 0.0;
        region_grow(s.x, s.y, angles, xs, ys, reg, reg_size, reg_angle, used, prec);
        if (reg_size < min_reg_size) continue;
        Rect rec;
        region2rect(reg, reg_size, modgrad, xs, reg_angle, prec, p, rec);
        if (!refine(ctx, reg, reg_size, reg_angle, prec, p, rec, density_th)) continue;
        const double log_nfa = rect_improve(rec, angles, xs, ys, logNT, log_eps);
        if (log_nfa <= log_eps) continue;
        rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;   // the gradient sits between the pixels of its mask
        if (scale != 1.0) {
            rec.x1 /= scale; rec.y1 /= scale; rec.x2 /= scale; rec.y2 /= scale;
            rec.width /= scale;
        }
        if (count < max_segments) {
            double* o = out + 7 * (size_t)count;
            o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width; o[5] = rec.p; o[6] = log_nfa;
        }
        ++count;
    }
    *n_out = count;                                            // may exceed max_segments: call again with a larger buffer
    return VPK_OK;
}
