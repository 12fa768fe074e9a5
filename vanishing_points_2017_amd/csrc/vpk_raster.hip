// vpk_raster.hip -- the reference's sphere rasteriser (sphere_mapping.py:36-72) for gfx950, faithful to what the reference
// actually executes: matplotlib's Agg backend.
//
// The reference plots, for every line (a, b, c), beta(alpha) = atan((-a sin alpha - c cos alpha) / b) at 10 000 alpha in
// [-pi/2, pi/2] (:40,:61-63) with ax.plot(..., c=[1, 1, 1, alpha]) on black axes that fill a size x size canvas, reads the
// canvas back and averages R, G, B (:65-68).  What that does to a pixel is defined by third-party code -- matplotlib's
// RendererAgg::draw_path and the Anti-Grain Geometry library it embeds -- whose stages for a solid anti-aliased Line2D
// are restated here one by one (the CPU restatement, pinned bit for bit against matplotlib itself and against the
// rasters stored in tests/golden, is oracle/agg_raster.py; it cites the sources stage by stage):
//
//   curve -> pixels           x = (alpha + pi/2) / pi * W,  y = H - (beta + pi/2) / pi * H
//   PathSimplifier            runs of segments that stay within 1/9 px of the run's first segment's line are merged
//   agg::conv_stroke          width 100/72 px, projecting caps, round joins (mitred when almost straight), inner miter
//   rasterizer_scanline_aa    24.8 fixed-point cells (cover, area) with the canvas as clip box, non-zero winding
//   fixed_blender_rgba_plain  white, alpha8 = uround(255 alpha); per pixel a = round(alpha8 cover / 255);
//                             p' = ((65280 - 255 p) a + 65280 p) / (65280 + a), truncated; ONE LINE AFTER THE OTHER
//   the axes' four spines     0.8 pt black strokes snapped to the pixel centres of the border, drawn over the lines
//
// Decomposition.  One persistent workgroup per image (images come from a queue).  Phase A: one THREAD per line evaluates
// the 10 000 samples, simplifies them on the fly (the simplifier is a sequential state machine) and strokes the result:
// a closed outline polygon of ~100-200 vertices in the workgroup's HBM scratch.  Phase B, line by line in input order
// (the blend is not commutative in 8-bit arithmetic): the polygon's edges are dealt to the threads and walked twice -- first
// to find every row's cell range, then, after a prefix sum has packed the ranges into a 96 KB LDS pool, to add the
// cells (cover, area) there with LDS atomics (sums are order independent: same cells as Agg's sorted list) -- and one
// thread per touched row sweeps its cells left to right: running cover, alpha, blend into the uint8 image.  Four
// workgroup barriers per line.
#include "vpk_internal.hpp"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int RT = 512;                 // threads per workgroup
constexpr int MAXS = 384;               // simplified points kept per line (typical: 30-100)
constexpr int MAXV = 1024;              // outline vertices per line (typical: 60-200)
constexpr int MAXSUB = 4;               // sub-paths per line (a NaN sample breaks the path)
constexpr int SUB = 256, SHIFT = 8;     // agg::poly_subpixel_scale / _shift
constexpr double PI_D = 3.14159265358979323846;
constexpr unsigned FLAG_OVERFLOW = 1u;  // a line produced more points / vertices / sub-paths than the buffers hold

struct V2 { double x, y; };

__device__ __forceinline__ int iround(double v) { return (int)(v < 0.0 ? v - 0.5 : v + 0.5); }

// ---------------------------------------------------------------------------------------------------------------
// PathSimplifier (matplotlib src/path_converters.h) as a push machine: feed() the vertices, it emit()s the kept ones
// ---------------------------------------------------------------------------------------------------------------
struct Simplifier {
    V2* out; int n, cap; unsigned* flags;
    double thr2;
    double lastx, lasty, origdx, origdy, orig_norm2, fwd_max, bwd_max, nextx, nexty, nbx, nby, startx, starty;
    bool last_fwd, last_bwd, clipped, have;
    __device__ void init(V2* o, int capacity, unsigned* fl) {
        out = o; n = 0; cap = capacity; flags = fl;
        thr2 = (1.0 / 9.0) * (1.0 / 9.0);
        have = false;
    }
    __device__ void emit(double x, double y) {
        if (n < cap) { out[n].x = x; out[n].y = y; ++n; } else { *flags |= FLAG_OVERFLOW; }
    }
    __device__ void begin(double x, double y) {      // move_to
        lastx = x; lasty = y; orig_norm2 = 0.0; bwd_max = 0.0; clipped = true; have = true;
        origdx = origdy = fwd_max = nextx = nexty = nbx = nby = startx = starty = 0.0;
        last_fwd = last_bwd = false;
    }
    __device__ void feed(double x, double y) {       // line_to
        if (orig_norm2 == 0.0) {
            if (clipped) { emit(lastx, lasty); clipped = false; }
            origdx = x - lastx; origdy = y - lasty;
            orig_norm2 = origdx * origdx + origdy * origdy;
            fwd_max = orig_norm2; bwd_max = 0.0; last_fwd = true; last_bwd = false;
            startx = lastx; starty = lasty;
            nextx = lastx = x; nexty = lasty = y;
            return;
        }
        const double totdx = x - startx, totdy = y - starty;
        const double totdot = origdx * totdx + origdy * totdy;
        const double paradx = totdot * origdx / orig_norm2, parady = totdot * origdy / orig_norm2;
        const double perpdx = totdx - paradx, perpdy = totdy - parady;
        const double perp2 = perpdx * perpdx + perpdy * perpdy;
        if (perp2 < thr2) {
            const double para2 = paradx * paradx + parady * parady;
            last_fwd = last_bwd = false;
            if (totdot > 0.0) {
                if (para2 > fwd_max) { last_fwd = true; fwd_max = para2; nextx = x; nexty = y; }
            } else {
                if (para2 > bwd_max) { last_bwd = true; bwd_max = para2; nbx = x; nby = y; }
            }
            lastx = x; lasty = y;
            return;
        }
        // _push: the run ends here
        double ex, ey;                                  // the last point written
        if (bwd_max > 0.0) {
            if (last_fwd) { emit(nbx, nby); emit(nextx, nexty); ex = nextx; ey = nexty; }
            else { emit(nextx, nexty); emit(nbx, nby); ex = nbx; ey = nby; }
        } else { emit(nextx, nexty); ex = nextx; ey = nexty; }
        if (clipped || (!last_fwd && !last_bwd)) { emit(lastx, lasty); ex = lastx; ey = lasty; }
        origdx = x - lastx; origdy = y - lasty;
        orig_norm2 = origdx * origdx + origdy * origdy;
        fwd_max = orig_norm2; last_fwd = true;
        startx = ex; starty = ey;
        lastx = nextx = x; lasty = nexty = y;
        bwd_max = 0.0; last_bwd = false; clipped = false;
    }
    __device__ void end() {                           // path_cmd_stop
        if (!have) return;
        if (orig_norm2 != 0.0) {
            emit(nextx, nexty);
            if (bwd_max > 0.0) emit(nbx, nby);
        }
        emit(lastx, lasty);
        have = false;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// agg::conv_stroke (agg_vcgen_stroke.cpp + agg_math_stroke.h): square caps, round joins, inner miter, scale 1
// ---------------------------------------------------------------------------------------------------------------
struct Outline {
    V2* v; int n, cap; unsigned* flags;
    __device__ void add(double x, double y) {
        if (n < cap) { v[n].x = x; v[n].y = y; ++n; } else { *flags |= FLAG_OVERFLOW; }
    }
};

__device__ bool calc_intersection(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy,
                                  double* x, double* y) {
    const double num = (ay - cy) * (dx - cx) - (ax - cx) * (dy - cy);
    const double den = (bx - ax) * (dy - cy) - (by - ay) * (dx - cx);
    if (fabs(den) < 1.0e-30) return false;
    const double r = num / den;
    *x = ax + r * (bx - ax);
    *y = ay + r * (by - ay);
    return true;
}
__device__ __forceinline__ double cross3(double x1, double y1, double x2, double y2, double x, double y) {
    return (x - x2) * (y2 - y1) - (y - y2) * (x2 - x1);
}

struct Stroker {
    double w, w_abs, w_eps;
    __device__ void init(double width) { w = width * 0.5; w_abs = fabs(w); w_eps = w / 1024.0; }
    __device__ void cap(Outline& o, const V2& v0, const V2& v1, double len) const {
        double dx1 = (v1.y - v0.y) / len, dy1 = (v1.x - v0.x) / len;
        dx1 *= w; dy1 *= w;
        const double dx2 = dy1, dy2 = dx1;               // square cap
        o.add(v0.x - dx1 - dx2, v0.y + dy1 - dy2);
        o.add(v0.x + dx1 - dx2, v0.y - dy1 - dy2);
    }
    __device__ void miter(Outline& o, const V2& v0, const V2& v1, const V2& v2, double dx1, double dy1, double dx2,
                          double dy2, double mlimit) const {
        const double lim = w_abs * mlimit;
        bool exceeded = true;
        double xi, yi;
        if (calc_intersection(v0.x + dx1, v0.y - dy1, v1.x + dx1, v1.y - dy1, v1.x + dx2, v1.y - dy2, v2.x + dx2,
                              v2.y - dy2, &xi, &yi)) {
            const double di = sqrt((xi - v1.x) * (xi - v1.x) + (yi - v1.y) * (yi - v1.y));
            if (di <= lim) { o.add(xi, yi); exceeded = false; }
        } else {
            const double x2 = v1.x + dx1, y2 = v1.y - dy1;
            if ((cross3(v0.x, v0.y, v1.x, v1.y, x2, y2) < 0.0) == (cross3(v1.x, v1.y, v2.x, v2.y, x2, y2) < 0.0)) {
                o.add(v1.x + dx1, v1.y - dy1);
                exceeded = false;
            }
        }
        if (exceeded) {                                  // miter_join_revert
            o.add(v1.x + dx1, v1.y - dy1);
            o.add(v1.x + dx2, v1.y - dy2);
        }
    }
    __device__ void arc(Outline& o, double x, double y, double dx1, double dy1, double dx2, double dy2) const {
        double a1 = atan2(dy1, dx1), a2 = atan2(dy2, dx2);
        double da = acos(w_abs / (w_abs + 0.125)) * 2;
        o.add(x + dx1, y + dy1);
        if (a1 > a2) a2 += 2 * PI_D;
        const int n = (int)((a2 - a1) / da);
        da = (a2 - a1) / (n + 1);
        a1 += da;
        for (int i = 0; i < n; ++i) {
            o.add(x + cos(a1) * w, y + sin(a1) * w);
            a1 += da;
        }
        o.add(x + dx2, y + dy2);
    }
    __device__ void join(Outline& o, const V2& v0, const V2& v1, const V2& v2, double len1, double len2) const {
        const double dx1 = w * (v1.y - v0.y) / len1, dy1 = w * (v1.x - v0.x) / len1;
        const double dx2 = w * (v2.y - v1.y) / len2, dy2 = w * (v2.x - v1.x) / len2;
        const double cp = cross3(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
        if (cp != 0 && (cp > 0) == (w > 0)) {            // inner join: inner_miter
            double limit = (len1 < len2 ? len1 : len2) / w_abs;
            if (limit < 1.01) limit = 1.01;
            miter(o, v0, v1, v2, dx1, dy1, dx2, dy2, limit);
            return;
        }
        double dx = (dx1 + dx2) / 2, dy = (dy1 + dy2) / 2;
        const double dbevel = sqrt(dx * dx + dy * dy);
        if ((w_abs - dbevel) < w_eps) {                  // no visible bevel: one point
            if (calc_intersection(v0.x + dx1, v0.y - dy1, v1.x + dx1, v1.y - dy1, v1.x + dx2, v1.y - dy2, v2.x + dx2,
                                  v2.y - dy2, &dx, &dy))
                o.add(dx, dy);
            else
                o.add(v1.x + dx1, v1.y - dy1);
            return;
        }
        arc(o, v1.x, v1.y, dx1, -dy1, dx2, -dy2);        // round join
    }
};

// vcgen_stroke on an open polyline p[0..n): vertex_sequence<vertex_dist> drops a vertex that coincides with its
// predecessor (in place), then cap, joins forward, cap, joins backward.  Returns the number of outline vertices added.
__device__ void stroke_outline(V2* p, int n, double width, Outline& o) {
    int m = 0;                                           // compacted length
    for (int i = 0; i < n; ++i) {
        if (m > 1) {
            const double d = sqrt((p[m - 1].x - p[m - 2].x) * (p[m - 1].x - p[m - 2].x) +
                                  (p[m - 1].y - p[m - 2].y) * (p[m - 1].y - p[m - 2].y));
            if (!(d > 1e-14)) --m;
        }
        p[m++] = p[i];
    }
    while (m > 1) {                                      // close(false): trailing coincident vertices go
        const double d = sqrt((p[m - 1].x - p[m - 2].x) * (p[m - 1].x - p[m - 2].x) +
                              (p[m - 1].y - p[m - 2].y) * (p[m - 1].y - p[m - 2].y));
        if (d > 1e-14) break;
        --m;
    }
    if (m < 2) return;
    Stroker st;
    st.init(width);
    auto dist = [&](int a, int b) {
        return sqrt((p[b].x - p[a].x) * (p[b].x - p[a].x) + (p[b].y - p[a].y) * (p[b].y - p[a].y));
    };
    st.cap(o, p[0], p[1], dist(0, 1));
    for (int i = 1; i < m - 1; ++i) st.join(o, p[i - 1], p[i], p[i + 1], dist(i - 1, i), dist(i, i + 1));
    st.cap(o, p[m - 1], p[m - 2], dist(m - 2, m - 1));
    for (int i = m - 2; i > 0; --i) st.join(o, p[i + 1], p[i], p[i - 1], dist(i, i + 1), dist(i - 1, i));
}

// ---------------------------------------------------------------------------------------------------------------
// rasterizer_cells_aa::line / render_hline: cells of one edge, added with integer atomics
// ---------------------------------------------------------------------------------------------------------------
// Where the cells of the line being drawn live.  Default: an LDS pool -- a stroke touches a few thousand pixels, so the
// polygon's edges are walked twice: pass 1 (BOUNDS) only records every row's first / last cell, a prefix sum over the rows
// packs the rows' cell ranges into the pool, pass 2 (POOL) adds the cells there with LDS atomics.  A polygon whose ranges
// do not fit the pool (GLOBAL) uses image-sized accumulators in HBM / L2 with global atomics instead.
constexpr int POOL = 12288;             // cells (cover, area) of one polygon in LDS: 96 KB
enum SinkMode { BOUNDS = 0, POOLED = 1, GLOBAL = 2 };

struct CellSink {
    int* cover; int* area;              // GLOBAL: [size][size + 2], x shifted by one (cells at x = -1 and x = size exist)
    int* pcover; int* parea;            // POOLED: LDS pool
    int* rowmin; int* rowmax; int* rowoff;   // LDS, per row
    int size;
    template <int MODE> __device__ __forceinline__ void add(int ex, int ey, int c, int a) const {
        if ((c | a) == 0) return;
        if (ey < 0 || ey >= size || ex < -1 || ex > size) return;
        if (MODE == BOUNDS) {
            atomicMin(rowmin + ey, ex + 1);
            atomicMax(rowmax + ey, ex + 1);
        } else if (MODE == POOLED) {
            const int idx = rowoff[ey] + (ex + 1 - rowmin[ey]);
            if (c) atomicAdd(pcover + idx, c);
            if (a) atomicAdd(parea + idx, a);
        } else {
            const size_t idx = (size_t)ey * (size + 2) + ex + 1;
            if (c) atomicAdd(cover + idx, c);
            if (a) atomicAdd(area + idx, a);
        }
    }
    template <int MODE> __device__ void hline(int ey, int x1, int y1, int x2, int y2) const {
        int ex1 = x1 >> SHIFT;
        const int ex2 = x2 >> SHIFT, fx1 = x1 & (SUB - 1), fx2 = x2 & (SUB - 1);
        if (y1 == y2) return;
        if (ex1 == ex2) {
            const int delta = y2 - y1;
            add<MODE>(ex1, ey, delta, (fx1 + fx2) * delta);
            return;
        }
        int p = (SUB - fx1) * (y2 - y1), first = SUB, incr = 1, dx = x2 - x1;
        if (dx < 0) { p = fx1 * (y2 - y1); first = 0; incr = -1; dx = -dx; }
        int delta = p / dx, mod = p % dx;
        if (mod < 0) { --delta; mod += dx; }
        add<MODE>(ex1, ey, delta, (fx1 + first) * delta);
        ex1 += incr;
        y1 += delta;
        if (ex1 != ex2) {
            p = SUB * (y2 - y1 + delta);
            int lift = p / dx, rem = p % dx;
            if (rem < 0) { --lift; rem += dx; }
            mod -= dx;
            while (ex1 != ex2) {
                delta = lift;
                mod += rem;
                if (mod >= 0) { mod -= dx; ++delta; }
                add<MODE>(ex1, ey, delta, SUB * delta);
                y1 += delta;
                ex1 += incr;
            }
        }
        delta = y2 - y1;
        add<MODE>(ex1, ey, delta, (fx2 + SUB - first) * delta);
    }
    template <int MODE> __device__ void line(int x1, int y1, int x2, int y2) const {
        const int dx = x2 - x1;
        int dy = y2 - y1;
        int ey1 = y1 >> SHIFT;
        const int ey2 = y2 >> SHIFT, fy1 = y1 & (SUB - 1), fy2 = y2 & (SUB - 1);
        if (ey1 == ey2) { hline<MODE>(ey1, x1, fy1, x2, fy2); return; }
        int incr = 1;
        if (dx == 0) {
            const int ex = x1 >> SHIFT;
            const int two_fx = (x1 - (ex << SHIFT)) << 1;
            int first = SUB;
            if (dy < 0) { first = 0; incr = -1; }
            int delta = first - fy1;
            add<MODE>(ex, ey1, delta, two_fx * delta);
            ey1 += incr;
            delta = first + first - SUB;
            const int a = two_fx * delta;
            while (ey1 != ey2) { add<MODE>(ex, ey1, delta, a); ey1 += incr; }
            delta = fy2 - SUB + first;
            add<MODE>(ex, ey1, delta, two_fx * delta);
            return;
        }
        long long p = (long long)(SUB - fy1) * dx;
        int first = SUB;
        if (dy < 0) { p = (long long)fy1 * dx; first = 0; incr = -1; dy = -dy; }
        int delta = (int)(p / dy), mod = (int)(p % dy);
        if (mod < 0) { --delta; mod += dy; }
        int x_from = x1 + delta;
        hline<MODE>(ey1, x1, fy1, x_from, first);
        ey1 += incr;
        if (ey1 != ey2) {
            p = (long long)SUB * dx;
            int lift = (int)(p / dy), rem = (int)(p % dy);
            if (rem < 0) { --lift; rem += dy; }
            mod -= dy;
            while (ey1 != ey2) {
                delta = lift;
                mod += rem;
                if (mod >= 0) { mod -= dy; ++delta; }
                const int x_to = x_from + delta;
                hline<MODE>(ey1, x_from, SUB - first, x_to, first);
                x_from = x_to;
                ey1 += incr;
            }
        }
        hline<MODE>(ey1, x_from, SUB - first, x2, fy2);
    }
};

// rasterizer_sl_clip<ras_conv_dbl>::line_to for ONE edge (the clipper's only state is the previous vertex)
struct EdgeClip {
    double bx1, by1, bx2, by2;
    const CellSink* c;
    __device__ __forceinline__ unsigned flags(double x, double y) const {
        return (unsigned)(x > bx2) | ((unsigned)(y > by2) << 1) | ((unsigned)(x < bx1) << 2) | ((unsigned)(y < by1) << 3);
    }
    __device__ __forceinline__ unsigned flags_y(double y) const { return ((unsigned)(y > by2) << 1) | ((unsigned)(y < by1) << 3); }
    template <int MODE> __device__ void clip_y(double x1, double y1, double x2, double y2, unsigned f1, unsigned f2) const {
        f1 &= 10; f2 &= 10;
        if ((f1 | f2) == 0) { c->line<MODE>(iround(x1 * SUB), iround(y1 * SUB), iround(x2 * SUB), iround(y2 * SUB)); return; }
        if (f1 == f2) return;
        double tx1 = x1, ty1 = y1, tx2 = x2, ty2 = y2;
        if (f1 & 8) { tx1 = x1 + (by1 - y1) * (x2 - x1) / (y2 - y1); ty1 = by1; }
        if (f1 & 2) { tx1 = x1 + (by2 - y1) * (x2 - x1) / (y2 - y1); ty1 = by2; }
        if (f2 & 8) { tx2 = x1 + (by1 - y1) * (x2 - x1) / (y2 - y1); ty2 = by1; }
        if (f2 & 2) { tx2 = x1 + (by2 - y1) * (x2 - x1) / (y2 - y1); ty2 = by2; }
        c->line<MODE>(iround(tx1 * SUB), iround(ty1 * SUB), iround(tx2 * SUB), iround(ty2 * SUB));
    }
    template <int MODE> __device__ void edge(double x1, double y1, double x2, double y2) const {
        const unsigned f1 = flags(x1, y1), f2 = flags(x2, y2);
        if ((f1 & 10) == (f2 & 10) && (f1 & 10) != 0) return;      // invisible by y
        double y3, y4;
        unsigned f3, f4;
        switch (((f1 & 5) << 1) | (f2 & 5)) {
        case 0: clip_y<MODE>(x1, y1, x2, y2, f1, f2); break;
        case 1:
            y3 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1); f3 = flags_y(y3);
            clip_y<MODE>(x1, y1, bx2, y3, f1, f3); clip_y<MODE>(bx2, y3, bx2, y2, f3, f2); break;
        case 2:
            y3 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1); f3 = flags_y(y3);
            clip_y<MODE>(bx2, y1, bx2, y3, f1, f3); clip_y<MODE>(bx2, y3, x2, y2, f3, f2); break;
        case 3: clip_y<MODE>(bx2, y1, bx2, y2, f1, f2); break;
        case 4:
            y3 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1); f3 = flags_y(y3);
            clip_y<MODE>(x1, y1, bx1, y3, f1, f3); clip_y<MODE>(bx1, y3, bx1, y2, f3, f2); break;
        case 6:
            y3 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1); y4 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1);
            f3 = flags_y(y3); f4 = flags_y(y4);
            clip_y<MODE>(bx2, y1, bx2, y3, f1, f3); clip_y<MODE>(bx2, y3, bx1, y4, f3, f4); clip_y<MODE>(bx1, y4, bx1, y2, f4, f2); break;
        case 8:
            y3 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1); f3 = flags_y(y3);
            clip_y<MODE>(bx1, y1, bx1, y3, f1, f3); clip_y<MODE>(bx1, y3, x2, y2, f3, f2); break;
        case 9:
            y3 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1); y4 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1);
            f3 = flags_y(y3); f4 = flags_y(y4);
            clip_y<MODE>(bx1, y1, bx1, y3, f1, f3); clip_y<MODE>(bx1, y3, bx2, y4, f3, f4); clip_y<MODE>(bx2, y4, bx2, y2, f4, f2); break;
        case 12: clip_y<MODE>(bx1, y1, bx1, y2, f1, f2); break;
        default: break;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// blend: fixed_blender_rgba_plain on an opaque grey pixel (R = G = B, A = 255); rgba8::multiply for the cover
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned calc_alpha(int a) {
    int cover = a >> (SHIFT * 2 + 1 - 8);
    if (cover < 0) cover = -cover;
    return cover > 255 ? 255u : (unsigned)cover;
}
__device__ __forceinline__ unsigned blend(unsigned p, unsigned grey, unsigned a8, unsigned cover) {
    if (a8 == 255 && cover == 255) return grey;          // opaque colour at full coverage: the pixel is copied
    const unsigned t = a8 * cover + 128;
    const unsigned alpha = ((t >> 8) + t) >> 8;
    if (alpha == 0) return p;
    const unsigned r = p * 255u;
    const unsigned a = ((alpha + 255u) << 8) - alpha * 255u;
    return (unsigned)((((int)(grey << 8) - (int)r) * (int)alpha + (int)(r << 8)) / (int)a);
}

struct PolyRef { int first, count; };   // vertices [first, first + count) of the line's vertex buffer: one closed polygon

// One workgroup rasterises one image.  scratch per workgroup: simplified points RT x MAXS, outline vertices RT x MAXV,
// polygon tables, cover / area accumulators.
struct RasterArgs {
    const double* l; const long long* offsets; const double* tab; int batch; int size; int samples; unsigned a8;
    unsigned char* out; int* queue; unsigned* flags;
    V2* simp; V2* verts; int* polys;     // per workgroup: RT * MAXS, RT * MAXV, RT * (1 + 2 * MAXSUB)
    int* cover; int* area;               // per workgroup: size * (size + 2) each, zero between lines
};

// sweep_scanline + render_scanline_aa_solid of one row: cells cell(0 .. len) (cover, area), first cell at pixel lo - 1
template <class GetCell>
__device__ __forceinline__ void sweep_row(GetCell cell, int lo, int hi, unsigned char* prow, int size, unsigned grey, unsigned a8) {
    int cover = 0, span_from = 0;
    bool span = false;
    for (int xi = lo; xi <= hi; ++xi) {
        int c, a;
        cell(xi - lo, c, a);
        if ((c | a) == 0) continue;
        const int cx = xi - 1;                            // pixel x of cell index xi
        if (span && cx > span_from) {                     // the run of whole pixels between two cells
            const unsigned al = calc_alpha(cover << (SHIFT + 1));
            if (al)
                for (int xx = span_from < 0 ? 0 : span_from; xx < cx && xx < size; ++xx)
                    prow[xx] = (unsigned char)blend(prow[xx], grey, a8, al);
        }
        cover += c;
        int x = cx;
        if (a) {
            const unsigned al = calc_alpha((cover << (SHIFT + 1)) - a);
            if (al && x >= 0 && x < size) prow[x] = (unsigned char)blend(prow[x], grey, a8, al);
            ++x;
        }
        span = true;
        span_from = x;
    }
}

// One closed polygon: cells, then one thread per touched row sweeps and blends.  img: the image in HBM; every row is
// read and written by one thread only (the same thread for every polygon), lines are ordered by the barriers.
__device__ void raster_polygon(const V2* v, int n, unsigned grey, unsigned a8, const CellSink& sink, unsigned char* img,
                               int size, int* s_total) {
    EdgeClip ec;
    ec.bx1 = 0.0; ec.by1 = 0.0; ec.bx2 = (double)size; ec.by2 = (double)size; ec.c = &sink;
    // pass 1: the rows' cell ranges
    for (int k = threadIdx.x; k < n; k += RT) {
        const V2 a = v[k], b = v[k + 1 < n ? k + 1 : 0];
        ec.edge<BOUNDS>(a.x, a.y, b.x, b.y);
    }
    __syncthreads();
    // exclusive prefix sum of the ranges' lengths over the rows (one wave, 16 rows per lane at size <= 1024)
    if (threadIdx.x < 64) {
        const int per = (size + 63) / 64;
        const int y0 = threadIdx.x * per;
        int sum = 0;
        for (int y = y0; y < y0 + per && y < size; ++y) {
            const int len = sink.rowmax[y] >= sink.rowmin[y] ? sink.rowmax[y] - sink.rowmin[y] + 1 : 0;
            sum += len;
        }
        int incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += up;
        }
        int off = incl - sum;
        for (int y = y0; y < y0 + per && y < size; ++y) {
            sink.rowoff[y] = off;
            off += sink.rowmax[y] >= sink.rowmin[y] ? sink.rowmax[y] - sink.rowmin[y] + 1 : 0;
        }
        if (threadIdx.x == 63) *s_total = incl;
    }
    __syncthreads();
    const bool pooled = *s_total <= POOL;
    // pass 2: the cells
    if (pooled) {
        for (int k = threadIdx.x; k < n; k += RT) {
            const V2 a = v[k], b = v[k + 1 < n ? k + 1 : 0];
            ec.edge<POOLED>(a.x, a.y, b.x, b.y);
        }
    } else {
        for (int k = threadIdx.x; k < n; k += RT) {
            const V2 a = v[k], b = v[k + 1 < n ? k + 1 : 0];
            ec.edge<GLOBAL>(a.x, a.y, b.x, b.y);
        }
    }
    __syncthreads();
    const int ldc = size + 2;
    for (int y = threadIdx.x; y < size; y += RT) {
        const int lo = sink.rowmin[y], hi = sink.rowmax[y];
        if (hi < lo) continue;
        unsigned char* prow = img + (size_t)y * size;
        if (pooled) {
            int* pc = sink.pcover + sink.rowoff[y];
            int* pa = sink.parea + sink.rowoff[y];
            sweep_row([&](int q, int& c, int& a) { c = pc[q]; a = pa[q]; pc[q] = 0; pa[q] = 0; }, lo, hi, prow, size, grey, a8);
        } else {
            int* crow = sink.cover + (size_t)y * ldc + lo;
            int* arow = sink.area + (size_t)y * ldc + lo;
            sweep_row([&](int q, int& c, int& a) { c = atomicExch(crow + q, 0); a = atomicExch(arow + q, 0); }, lo, hi, prow,
                      size, grey, a8);
        }
        sink.rowmin[y] = 0x7fffffff;
        sink.rowmax[y] = -1;
    }
    __syncthreads();
}

// sin / cos of the sample angles: the same 10 000 values for every line of every image
__global__ void raster_table_kernel(int ns, double* tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const double lo_a = -PI_D / 2, hi_a = PI_D / 2;
    const double step = (hi_a - lo_a) / (ns - 1);
    const double al = (i == ns - 1) ? hi_a : lo_a + i * step;       // numpy.linspace
    tab[3 * i] = al;
    tab[3 * i + 1] = sin(al);
    tab[3 * i + 2] = cos(al);
}

__global__ __launch_bounds__(RT) void raster_kernel(RasterArgs A) {
    __shared__ int s_rowmin[1024], s_rowmax[1024], s_rowoff[1024];
    __shared__ int s_pcover[POOL], s_parea[POOL];
    __shared__ int s_img, s_total;
    const int size = A.size;
    const int wg = blockIdx.x;
    V2* simp = A.simp + (size_t)wg * RT * MAXS;
    V2* verts = A.verts + (size_t)wg * RT * MAXV;
    int* polys = A.polys + (size_t)wg * RT * (1 + 2 * MAXSUB);
    CellSink sink;
    sink.cover = A.cover + (size_t)wg * size * (size + 2);
    sink.area = A.area + (size_t)wg * size * (size + 2);
    sink.rowmin = s_rowmin; sink.rowmax = s_rowmax; sink.rowoff = s_rowoff; sink.size = size;
    sink.pcover = s_pcover; sink.parea = s_parea;
    for (int y = threadIdx.x; y < 1024; y += RT) { s_rowmin[y] = 0x7fffffff; s_rowmax[y] = -1; s_rowoff[y] = 0; }
    for (int q = threadIdx.x; q < POOL; q += RT) { s_pcover[q] = 0; s_parea[q] = 0; }   // the sweeps leave the pool zero again
    const double width_px = 100.0 / 72.0;                 // 1 pt at 100 dpi (matplotlib 1.5.1's default line width)
    for (;;) {
        if (threadIdx.x == 0) s_img = atomicAdd(A.queue, 1);
        __syncthreads();
        const int img_i = s_img;
        __syncthreads();
        if (img_i >= A.batch) break;
        unsigned char* img = A.out + (size_t)img_i * size * size;
        for (int p = threadIdx.x; p < size * size; p += RT) img[p] = 0;
        const long long lo = A.offsets[img_i], hi = A.offsets[img_i + 1];
        unsigned* fl = A.flags + img_i;
        if (threadIdx.x == 0) *fl = 0;
        __syncthreads();
        for (long long base = lo; base < hi; base += RT) {
            const int nb = (int)((hi - base) < RT ? (hi - base) : RT);
            const long long t_a = wall_clock64();
            // ---- phase A: thread t -> outline of line base + t ----
            if ((int)threadIdx.x < nb) {
                const int t = threadIdx.x;
                const double la = A.l[3 * (base + t)], lb = A.l[3 * (base + t) + 1], lc = A.l[3 * (base + t) + 2];
                V2* sp = simp + (size_t)t * MAXS;
                Outline o;
                o.v = verts + (size_t)t * MAXV; o.n = 0; o.cap = MAXV; o.flags = fl;
                int* pt = polys + t * (1 + 2 * MAXSUB);
                int npoly = 0;
                Simplifier s;
                s.init(sp, MAXS, fl);
                const int ns = A.samples;
                const double lo_a = -PI_D / 2, hi_a = PI_D / 2;
                auto flush = [&]() {                      // end of a sub-path: stroke what the simplifier kept
                    s.end();
                    if (s.n >= 2) {
                        const int first = o.n;
                        stroke_outline(sp, s.n, width_px, o);
                        if (o.n - first >= 3) {
                            if (npoly < MAXSUB) { pt[1 + 2 * npoly] = first; pt[2 + 2 * npoly] = o.n - first; ++npoly; }
                            else atomicOr(fl, FLAG_OVERFLOW);
                        }
                    }
                    s.n = 0;
                };
                for (int i = 0; i < ns; ++i) {
                    const double al = A.tab[3 * i], sa = A.tab[3 * i + 1], ca = A.tab[3 * i + 2];   // (wave-uniform loads)
                    double be = -atan((-la * sa - lc * ca) / lb);                     // sphere_mapping.py:63
                    be *= -1;                                                         // :65
                    const double x = (al - lo_a) / (hi_a - lo_a) * size;
                    const double y = size - (be - lo_a) / (hi_a - lo_a) * size;
                    if (!(x == x) || !(y == y) || isinf(x) || isinf(y)) {             // PathNanRemover: breaks the path
                        if (s.have) flush();
                        continue;
                    }
                    if (!s.have) s.begin(x, y); else s.feed(x, y);
                }
                if (s.have) flush();
                pt[0] = npoly;
            }
            __syncthreads();
            const long long t_b = wall_clock64();
            // ---- phase B: the lines of this batch in order ----
            for (int t = 0; t < nb; ++t) {
                const int* pt = polys + t * (1 + 2 * MAXSUB);
                const int npoly = pt[0];
                for (int q = 0; q < npoly; ++q)
                    raster_polygon(verts + (size_t)t * MAXV + pt[1 + 2 * q], pt[2 + 2 * q], 255u, A.a8, sink, img, size, &s_total);
            }
            if (threadIdx.x == 0 && blockIdx.x == 0) {     // device time of the two phases (100 MHz ticks), workgroup 0
                atomicAdd(A.queue + 2, (int)(t_b - t_a));
                atomicAdd(A.queue + 3, (int)(wall_clock64() - t_b));
            }
        }
        // ---- the axes' spines over the lines: left, right, bottom, top (matplotlib's drawing order) ----
        {
            const double s = (double)size, w_spine = 0.8 * 100.0 / 72.0;
            V2* sp = simp;                                // the scratch of thread 0's slot serves
            V2* sv = verts;
            for (int side = 0; side < 4; ++side) {
                if (threadIdx.x == 0) {
                    // two-vertex rectilinear paths, snapped to pixel centres (PathSnapper: floor(v + 0.5) + 0.5 for a stroke
                    // whose width rounds to an odd number of pixels)
                    const double x0 = (side == 1) ? s : 0.0, y0 = (side == 3) ? 0.0 : s;
                    const double x1 = (side == 0) ? 0.0 : s, y1 = (side == 2) ? s : 0.0;
                    sp[0].x = floor(x0 + 0.5) + 0.5; sp[0].y = floor(y0 + 0.5) + 0.5;
                    sp[1].x = floor(x1 + 0.5) + 0.5; sp[1].y = floor(y1 + 0.5) + 0.5;
                    Outline o;
                    o.v = sv; o.n = 0; o.cap = MAXV; o.flags = fl;
                    stroke_outline(sp, 2, w_spine, o);
                    polys[0] = o.n;
                }
                __syncthreads();
                raster_polygon(sv, polys[0], 0u, 255u, sink, img, size, &s_total);
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int vpk_sphere_raster(vpk_handle* h, const double* l, const int64_t* offsets, int batch, int size, double alpha,
                      uint8_t* out) {
    if (!h || !l || !offsets || !out || batch < 1 || size < 8 || size > 1024 || !(alpha >= 0.0 && alpha <= 1.0))
        return vpk_fail(h, VPK_ERR_ARG, "vpk_sphere_raster: bad argument (size must be 8..1024, alpha 0..1)");
    VPK_HIP(h, hipSetDevice(h->device));
    int wgs = h->num_cu < batch ? h->num_cu : batch;
    // workspace: [offsets | queue + per-image flags | per workgroup: simplified points, outline vertices, polygon tables,
    //             cover, area]
    const size_t ob = vpk::em_align((size_t)(batch + 1) * 8, 256);
    const size_t fb = vpk::em_align(256 + (size_t)batch * 4, 256);
    const int samples = 10000;                            // sphere_mapping.py:40
    const size_t tb = vpk::em_align((size_t)samples * 3 * 8, 256);
    const size_t simp_b = (size_t)RT * MAXS * sizeof(V2), vert_b = (size_t)RT * MAXV * sizeof(V2);
    const size_t poly_b = vpk::em_align((size_t)RT * (1 + 2 * MAXSUB) * 4, 256);
    const size_t acc_b = vpk::em_align((size_t)size * (size + 2) * 4, 256);
    const size_t per_wg = simp_b + vert_b + poly_b + 2 * acc_b;
    while (wgs > 1 && ob + fb + tb + (size_t)wgs * per_wg > h->total_mem / 4) wgs /= 2;
    int rc = vpk_reserve(h, &h->raster_hdr, &h->raster_hdr_bytes, ob + fb + tb + (size_t)wgs * per_wg, "hipMalloc(raster workspace)");
    if (rc) return rc;
    char* base = (char*)h->raster_hdr;
    // offsets [host] -> device: caller-owned pageable memory, so the copy is waited for
    VPK_HIP(h, hipStreamSynchronize(h->stream));
    VPK_HIP(h, hipMemcpyAsync(base, offsets, (size_t)(batch + 1) * 8, hipMemcpyHostToDevice, h->stream));
    VPK_HIP(h, hipStreamSynchronize(h->stream));
    VPK_HIP(h, hipMemsetAsync(base + ob, 0, fb, h->stream));
    RasterArgs A;
    hipLaunchKernelGGL(raster_table_kernel, dim3((samples + 255) / 256), dim3(256), 0, h->stream, samples, (double*)(base + ob + fb));
    A.l = l; A.offsets = (const long long*)base; A.tab = (const double*)(base + ob + fb); A.batch = batch; A.size = size;
    A.samples = samples;
    A.a8 = (unsigned)(alpha * 255.0 + 0.5);               // agg::rgba8(rgba): uround
    A.out = out; A.queue = (int*)(base + ob); A.flags = (unsigned*)(base + ob + 256);
    char* p = base + ob + fb + tb;
    A.simp = (V2*)p; p += (size_t)wgs * simp_b;
    A.verts = (V2*)p; p += (size_t)wgs * vert_b;
    A.polys = (int*)p; p += (size_t)wgs * poly_b;
    A.cover = (int*)p; p += (size_t)wgs * acc_b;
    A.area = (int*)p;
    VPK_HIP(h, hipMemsetAsync(A.cover, 0, 2 * (size_t)wgs * acc_b, h->stream));    // the sweeps leave them zero again
    hipLaunchKernelGGL(raster_kernel, dim3(wgs), dim3(RT), 0, h->stream, A);
    VPK_HIP(h, hipGetLastError());
    if (getenv("VPK_RASTER_TIMES")) {                     // development: where workgroup 0 spent its time
        int q[8];
        VPK_HIP(h, hipStreamSynchronize(h->stream));
        VPK_HIP(h, hipMemcpy(q, A.queue, sizeof(q), hipMemcpyDeviceToHost));
        fprintf(stderr, "vpk_sphere_raster: workgroup 0: outlines %.2f ms, cells + blending %.2f ms (%d images on %d workgroups)\n",
                q[2] * 1e-5, q[3] * 1e-5, batch, wgs);
    }
    return VPK_OK;
}

}  // extern "C"
