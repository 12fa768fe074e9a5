// raster_device.hpp -- the arithmetic of the Agg-faithful sphere rasteriser (see vpk_raster.hip for the pipeline and its
// sources): PathSimplifier, conv_stroke, the cell walker with the clip box, calculate_alpha and the plain blender, as
// plain structs and functions.  Compiled by hipcc into the kernels of vpk_raster.hip and -- unmodified, with RDEV empty and
// the atomics plain -- by g++ into the test-only host build tests/hostsim/sim_raster.cpp, so that the CPU suite runs the
// product's own code against the reference's rasters.
#ifndef VPK_RASTER_DEVICE_HPP_
#define VPK_RASTER_DEVICE_HPP_

#include <math.h>

#ifdef __HIPCC__
#define RDEV __device__
#define RDEV_INLINE __device__ __forceinline__
#define RDEV_NOINLINE __device__ __noinline__
// the per-polygon accumulators live in LDS: pointers typed with the LDS address space make the accesses ds_* instructions
// (through generic pointers -- a sink handed to a noinline function -- they were flat_* instructions: LDS by way of the
// vector-memory address path, a round trip several times longer, with the waves parked on it 80 % of the time)
#define RS_LDS __attribute__((address_space(3)))
#define RS_ATOMIC_ADD(p, v) (void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RS_ATOMIC_MIN(p, v) (void)__hip_atomic_fetch_min((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RS_ATOMIC_MAX(p, v) (void)__hip_atomic_fetch_max((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RS_ATOMIC_ADD_GLOBAL(p, v) atomicAdd((p), (v))
#else
#define RDEV
#define RDEV_INLINE inline
#define RDEV_NOINLINE
#define RS_LDS
#define RS_ATOMIC_ADD(p, v) (*(p) += (v))
#define RS_ATOMIC_MIN(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
#define RS_ATOMIC_MAX(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#define RS_ATOMIC_ADD_GLOBAL(p, v) (*(p) += (v))
#endif

namespace vpk_raster {

constexpr int RT = 256;                 // threads per workgroup of coverage_kernel (one polygon at a time: ~250 work items)
constexpr int MAXS = 384;               // simplified points kept per line (typical: 30-100)
constexpr int MAXV = 1024;              // outline vertices per line (typical: 60-200)
constexpr int MAXSUB = 4;               // sub-paths per line (a NaN sample breaks the path)
constexpr int SUB = 256, SHIFT = 8;     // agg::poly_subpixel_scale / _shift
constexpr double PI_D = 3.14159265358979323846;
constexpr unsigned FLAG_OVERFLOW = 1u;  // a line produced more points / vertices / sub-paths than the buffers hold

struct V2 { double x, y; };

RDEV_INLINE int iround(double v) { return (int)(v < 0.0 ? v - 0.5 : v + 0.5); }

// ---------------------------------------------------------------------------------------------------------------
// PathSimplifier (matplotlib src/path_converters.h) as a push machine: feed() the vertices, it emit()s the kept ones
// ---------------------------------------------------------------------------------------------------------------
struct Simplifier {
    V2* out; int n, cap; unsigned overflow;     // overflow: FLAG_OVERFLOW once a vertex did not fit `out`
    double thr2;
    double lastx, lasty, origdx, origdy, orig_norm2, fwd_max, bwd_max, nextx, nexty, nbx, nby, startx, starty;
    bool last_fwd, last_bwd, clipped, have;
    int epoch;                                  // counts the changes of the run's constants (start, orig*): see feed_group
    RDEV_INLINE void init(V2* o, int capacity) {
        out = o; n = 0; cap = capacity; overflow = 0;
        thr2 = (1.0 / 9.0) * (1.0 / 9.0);
        have = false; epoch = 0;
        lastx = lasty = origdx = origdy = orig_norm2 = fwd_max = bwd_max = nextx = nexty = nbx = nby = startx = starty = 0.0;
        last_fwd = last_bwd = clipped = false;
    }
    RDEV_INLINE void emit(double x, double y) {
        if (n < cap) { out[n].x = x; out[n].y = y; ++n; } else { overflow |= FLAG_OVERFLOW; }
    }
    RDEV_INLINE void begin(double x, double y) {      // move_to
        lastx = x; lasty = y; orig_norm2 = 0.0; bwd_max = 0.0; clipped = true; have = true;
        origdx = origdy = fwd_max = nextx = nexty = nbx = nby = startx = starty = 0.0;
        last_fwd = last_bwd = false;
        ++epoch;
    }
    // The part of line_to that only reads the run's constants (its start and its first segment): where the vertex lies
    // along and across the run.  Within a run these are independent from vertex to vertex -- feed_group evaluates them for
    // a group of vertices side by side and feeds the results to the sequential part below.
    RDEV_INLINE void metrics(double x, double y, double& totdot, double& perp2, double& para2) const {
        const double totdx = x - startx, totdy = y - starty;
        totdot = origdx * totdx + origdy * totdy;
        const double paradx = totdot * origdx / orig_norm2, parady = totdot * origdy / orig_norm2;
        const double perpdx = totdx - paradx, perpdy = totdy - parady;
        perp2 = perpdx * perpdx + perpdy * perpdy;
        para2 = paradx * paradx + parady * parady;
    }
    RDEV_INLINE void first_segment(double x, double y) {
        if (clipped) { emit(lastx, lasty); clipped = false; }
        origdx = x - lastx; origdy = y - lasty;
        orig_norm2 = origdx * origdx + origdy * origdy;
        fwd_max = orig_norm2; bwd_max = 0.0; last_fwd = true; last_bwd = false;
        startx = lastx; starty = lasty;
        nextx = lastx = x; nexty = lasty = y;
        ++epoch;
    }
    RDEV_INLINE void push(double x, double y) {   // _push: the run ends at this vertex
        // (values first, then the choice: with the two orders written as two branches the compiler merges them into a
        //  choice between the ADDRESSES of the fields, which pins next* / nb* to private memory -- a scratch store per
        //  accepted vertex and a scratch round trip per run)
        const double fx = nextx, fy = nexty, gx = nbx, gy = nby;
        const bool both = bwd_max > 0.0, fwd_last = last_fwd;
        const double ax = both && fwd_last ? gx : fx, ay = both && fwd_last ? gy : fy;     // first point written
        const double cx = fwd_last ? fx : gx, cy = fwd_last ? fy : gy;                     // second one (if both)
        emit(ax, ay);
        if (both) emit(cx, cy);
        double ex = both ? cx : ax, ey = both ? cy : ay;   // the last point written
        if (clipped || (!last_fwd && !last_bwd)) { emit(lastx, lasty); ex = lastx; ey = lasty; }
        origdx = x - lastx; origdy = y - lasty;
        orig_norm2 = origdx * origdx + origdy * origdy;
        fwd_max = orig_norm2; last_fwd = true;
        startx = ex; starty = ey;
        lastx = nextx = x; lasty = nexty = y;
        bwd_max = 0.0; last_bwd = false; clipped = false;
        ++epoch;
    }
    // the sequential part of line_to, given metrics() of the vertex under the CURRENT run constants.  The usual case (the
    // vertex stays within the run) is written without branches -- selects on values: 64 lines share a wave, and every
    // divergent `if` costs the wave its exec-mask bookkeeping whether or not a lane takes it; the rare cases (first
    // segment of a path, end of a run) sit behind ONE branch.
    RDEV_INLINE void consume(double x, double y, double totdot, double perp2, double para2) {
        const bool started = orig_norm2 != 0.0;
        if (!(started && perp2 < thr2)) {
            if (!started) first_segment(x, y); else push(x, y);
            return;
        }
        const bool fwd_side = totdot > 0.0;
        const bool nf = fwd_side && para2 > fwd_max, nb = !fwd_side && para2 > bwd_max;
        last_fwd = nf; last_bwd = nb;
        fwd_max = nf ? para2 : fwd_max; nextx = nf ? x : nextx; nexty = nf ? y : nexty;
        bwd_max = nb ? para2 : bwd_max; nbx = nb ? x : nbx; nby = nb ? y : nby;
        lastx = x; lasty = y;
    }
    RDEV_INLINE void feed(double x, double y) {       // line_to
        double totdot = 0.0, perp2 = 0.0, para2 = 0.0;
        if (orig_norm2 != 0.0) metrics(x, y, totdot, perp2, para2);
        consume(x, y, totdot, perp2, para2);
    }
    RDEV_INLINE void end() {                           // path_cmd_stop
        if (!have) return;
        if (orig_norm2 != 0.0) {
            emit(nextx, nexty);
            if (bwd_max > 0.0) emit(nbx, nby);
        }
        emit(lastx, lasty);
        have = false;
    }
};

// G consecutive vertices of a path through PathNanRemover + PathSimplifier.  A thread is alone with the latency of its
// arithmetic (one line per thread, fewer waves than SIMDs), and a vertex costs two divisions before the simplifier can
// decide anything -- but those only depend on the run's constants, which change once in ~100 vertices.  So the metrics
// of all G vertices are evaluated first, as G independent chains, under the constants of the moment; the sequential part
// then takes them in order and re-evaluates a vertex's metrics only if a run ended (epoch changed) inside the group.
// Same operations on the same operands as vertex-by-vertex feeding, hence the same bits.
template <int G, typename Flush>
RDEV_INLINE void feed_group(Simplifier& sm, const double (&xs)[G], const double (&ys)[G], int count, Flush&& flush) {
    double td[G], pp[G], pa[G];
    const int ep = sm.epoch;
#pragma unroll
    for (int u = 0; u < G; ++u) sm.metrics(xs[u], ys[u], td[u], pp[u], pa[u]);
#pragma unroll
    for (int u = 0; u < G; ++u) {
        if (u >= count) break;
        const double x = xs[u], y = ys[u];
        if (!(x == x) || !(y == y) || isinf(x) || isinf(y)) {                 // PathNanRemover: breaks the path
            if (sm.have) flush();
            continue;
        }
        if (!sm.have) { sm.begin(x, y); continue; }
        if (sm.epoch != ep) sm.metrics(x, y, td[u], pp[u], pa[u]);
        sm.consume(x, y, td[u], pp[u], pa[u]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// agg::conv_stroke (agg_vcgen_stroke.cpp + agg_math_stroke.h): square caps, round joins, inner miter, scale 1
// ---------------------------------------------------------------------------------------------------------------
struct Outline {
    V2* v; int n, cap; unsigned* flags;
    RDEV void add(double x, double y) {
        if (n < cap) { v[n].x = x; v[n].y = y; ++n; } else { *flags |= FLAG_OVERFLOW; }
    }
};

RDEV bool calc_intersection(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy,
                                  double* x, double* y) {
    const double num = (ay - cy) * (dx - cx) - (ax - cx) * (dy - cy);
    const double den = (bx - ax) * (dy - cy) - (by - ay) * (dx - cx);
    if (fabs(den) < 1.0e-30) return false;
    const double r = num / den;
    *x = ax + r * (bx - ax);
    *y = ay + r * (by - ay);
    return true;
}
RDEV_INLINE double cross3(double x1, double y1, double x2, double y2, double x, double y) {
    return (x - x2) * (y2 - y1) - (y - y2) * (x2 - x1);
}

struct Stroker {
    double w, w_abs, w_eps;
    RDEV void init(double width) { w = width * 0.5; w_abs = fabs(w); w_eps = w / 1024.0; }
    RDEV void cap(Outline& o, const V2& v0, const V2& v1, double len) const {
        double dx1 = (v1.y - v0.y) / len, dy1 = (v1.x - v0.x) / len;
        dx1 *= w; dy1 *= w;
        const double dx2 = dy1, dy2 = dx1;               // square cap
        o.add(v0.x - dx1 - dx2, v0.y + dy1 - dy2);
        o.add(v0.x + dx1 - dx2, v0.y - dy1 - dy2);
    }
    RDEV void miter(Outline& o, const V2& v0, const V2& v1, const V2& v2, double dx1, double dy1, double dx2,
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
    RDEV void arc(Outline& o, double x, double y, double dx1, double dy1, double dx2, double dy2) const {
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
    RDEV void join(Outline& o, const V2& v0, const V2& v1, const V2& v2, double len1, double len2) const {
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
RDEV void stroke_outline(V2* p, int n, double width, Outline& o) {
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
constexpr int POOL = 7040;              // cells (cover, area) of one polygon in LDS: 55 KB (two workgroups per CU)
enum SinkMode { BOUNDS = 0, POOLED = 1, GLOBAL = 2 };

typedef RS_LDS int* lds_int_ptr;
// A row's cells come in up to TWO ranges.  The curve is the graph of a function with at most one interior extremum, so a row
// beside the apex is crossed twice, far apart, with nothing of the polygon in between: one range from the first to the last
// cell would consist mostly of that gap (measured: 6 100 entries per line, 1 600 of them cells).  The polygon's cells are
// therefore split at the apex column `xs`: cells left of it form the row's L range (rowmin / rowmax), the others its R range
// (rmin / rmax).  A row piece that straddles xs makes the two adjacent; and a gap may only be dropped if nothing is to be
// drawn in it, i.e. if the running cover across it -- the covers of the L cells, `lcov` -- is zero (the flat span at the
// apex itself has cells at its two ends only and IS drawn in between).  After the row scan `lcov` holds, per row, the
// number of L entries, or -1 if the row is one joined range [rowmin, rmax].
struct CellSink {
    int* cover; int* area;              // GLOBAL: [size][size + 2], x shifted by one (cells at x = -1 and x = size exist)
    lds_int_ptr pcover, parea;          // POOLED: LDS pool
    lds_int_ptr rowmin, rowmax, rowoff; // LDS, per row: L range (cell index + 1), first pool entry
    lds_int_ptr rmin, rmax, lcov;       // LDS, per row: R range; cover of the L pieces, then the split (see above)
    int xs;                             // first cell column of the R side (host build: INT_MAX -- one range per row)
    int size;
    int blo, bhi, boff;                 // POOLED: the band of rows [blo, bhi) the pool holds now; pool index = rowoff - boff
};
RDEV_INLINE int row_len(int lo, int hi) { return hi >= lo ? hi - lo + 1 : 0; }
// (the sink travels BY VALUE through the noinline walkers below: a pointer to it would be a pointer into the caller's
//  private memory, and every field access a scratch load)
template <int MODE> RDEV_INLINE void cell_add(const CellSink& s, int ex, int ey, int c, int a, int row_base) {
    if ((c | a) == 0) return;
    if (ex < -1 || ex > s.size) return;
    if (MODE == POOLED) {                     // row_base: the row's first pool entry minus its first cell (cell_hline)
        if (c) RS_ATOMIC_ADD(s.pcover + row_base + ex, c);
        if (a) RS_ATOMIC_ADD(s.parea + row_base + ex, a);
    } else {
        const size_t idx = (size_t)ey * (s.size + 2) + ex + 1;
        if (c) RS_ATOMIC_ADD_GLOBAL(s.cover + idx, c);
        if (a) RS_ATOMIC_ADD_GLOBAL(s.area + idx, a);
    }
}
// render_hline: the cells of one row's piece of an edge.  BOUNDS: only the row's first / last cell are wanted, and a
// superset will do (an entry of the row's range that no cell is added to sweeps to alpha 0): the piece's two end cells.
template <int MODE> RDEV_INLINE void cell_hline(const CellSink& s, int ey, int x1, int y1, int x2, int y2) {
    int ex1 = x1 >> SHIFT;
    const int ex2 = x2 >> SHIFT, fx1 = x1 & (SUB - 1), fx2 = x2 & (SUB - 1);
    if (y1 == y2) return;
    if (MODE == BOUNDS) {
        if (ey < 0 || ey >= s.size) return;
        int lo = ex1 < ex2 ? ex1 : ex2, hi = ex1 < ex2 ? ex2 : ex1;
        if (hi < -1 || lo > s.size) return;
        lo = lo < -1 ? -1 : lo; hi = hi > s.size ? s.size : hi;
        if (hi < s.xs) {                      // a piece of the L side: its cells' covers add up to y2 - y1
            RS_ATOMIC_MIN(s.rowmin + ey, lo + 1);
            RS_ATOMIC_MAX(s.rowmax + ey, hi + 1);
            if (s.lcov) RS_ATOMIC_ADD(s.lcov + ey, y2 - y1);
        } else if (lo >= s.xs) {              // of the R side
            RS_ATOMIC_MIN(s.rmin + ey, lo + 1);
            RS_ATOMIC_MAX(s.rmax + ey, hi + 1);
        } else {                              // across the split: the two ranges touch
            RS_ATOMIC_MIN(s.rowmin + ey, lo + 1);
            RS_ATOMIC_MAX(s.rowmax + ey, s.xs);
            RS_ATOMIC_MIN(s.rmin + ey, s.xs + 1);
            RS_ATOMIC_MAX(s.rmax + ey, hi + 1);
        }
        return;
    }
    if (ey < 0 || ey >= s.size) return;
    int row_base = 0;
    if (MODE == POOLED) {                     // the row's entries in the pool: looked up once per row piece, not per cell
        if (ey < s.blo || ey >= s.bhi) return;
        const int split = s.lcov[ey];         // entries of the L range, or -1: one joined range
        const int lo = ex1 < ex2 ? ex1 : ex2;
        row_base = s.rowoff[ey] - s.boff + 1 - s.rowmin[ey];
        if (split >= 0 && lo >= s.xs) row_base = s.rowoff[ey] - s.boff + split + 1 - s.rmin[ey];
    }
    if (ex1 == ex2) {
        const int delta = y2 - y1;
        cell_add<MODE>(s, ex1, ey, delta, (fx1 + fx2) * delta, row_base);
        return;
    }
    int p = (SUB - fx1) * (y2 - y1), first = SUB, incr = 1, dx = x2 - x1;
    if (dx < 0) { p = fx1 * (y2 - y1); first = 0; incr = -1; dx = -dx; }
    int delta = p / dx, mod = p % dx;
    if (mod < 0) { --delta; mod += dx; }
    cell_add<MODE>(s, ex1, ey, delta, (fx1 + first) * delta, row_base);
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
            cell_add<MODE>(s, ex1, ey, delta, SUB * delta, row_base);
            y1 += delta;
            ex1 += incr;
        }
    }
    delta = y2 - y1;
    cell_add<MODE>(s, ex1, ey, delta, (fx2 + SUB - first) * delta, row_base);
}
// rasterizer_cells_aa::line for the rows [part * nrows / nparts, (part + 1) * nrows / nparts) of the edge only: a long
// edge is shared by several threads.  AGG walks the rows with an integer DDA (x advances by lift, plus one whenever
// the running remainder wraps); after k middle rows the remainder has wrapped floor((mod0 + k rem) / dy) times, so
// any row's (x_from, x_to) follows in closed form and a thread can start in the middle of the edge with exactly the
// state the sequential walk has there.  (AGG's separate code for vertical edges gives the cells the general formulas
// give with dx = 0 -- a row's piece with x_from = x_to is one cell with area 2 fx delta --, so there is one path here.)
// Everything is inlined into ONE call site per pass (the pieces of an edge are looped over, EdgeClip::edge): called
// functions cost this kernel scratch round trips for their saved registers, several per item.
template <int MODE> RDEV_INLINE void cell_line(const CellSink& s, int x1, int y1, int x2, int y2, int part, int nparts) {
    const int dx = x2 - x1;
    int dy = y2 - y1;
    const int ey1 = y1 >> SHIFT, ey2 = y2 >> SHIFT, fy1 = y1 & (SUB - 1), fy2 = y2 & (SUB - 1);
    const int incr = dy < 0 ? -1 : 1;
    const int nrows = (ey2 - ey1) * incr + 1;
    int r0 = (int)((long long)part * nrows / nparts), r1 = (int)((long long)(part + 1) * nrows / nparts);
    if (nrows == 1) { r0 = 0; r1 = part == 0 ? 1 : 0; }
    if (r0 >= r1) return;
    const int first = dy < 0 ? 0 : SUB;
    // (32-bit like AGG: |dx|, dy <= 1024 px x 256, so 256 |dx| and mod0 + k rem stay below 2^29)
    int x_from0 = x2, lift = 0, rem = 0, mod0 = 0;
    if (nrows > 1) {
        int p = dy < 0 ? fy1 * dx : (SUB - fy1) * dx;
        if (dy < 0) dy = -dy;
        int delta0 = p / dy;
        mod0 = p % dy;
        if (mod0 < 0) { --delta0; mod0 += dy; }
        x_from0 = x1 + delta0;
        p = SUB * dx;
        lift = p / dy; rem = p % dy;
        if (rem < 0) { --lift; rem += dy; }
    } else {
        dy = 1;
    }
    int xf = r0 >= 1 ? x_from0 + (r0 - 1) * lift + (mod0 + (r0 - 1) * rem) / dy : x1;
    for (int r = r0; r < r1; ++r) {
        // the row's piece: (row, x and y-fraction where the edge enters it, x and y-fraction where it leaves)
        const bool head = r == 0, tail = r == nrows - 1;
        const int xt = tail ? x2 : (head ? x_from0 : x_from0 + r * lift + (mod0 + r * rem) / dy);
        const int ya = head ? fy1 : SUB - first, yb = tail ? fy2 : first;
        cell_hline<MODE>(s, ey1 + r * incr, xf, ya, xt, yb);
        xf = xt;
    }
}

// rasterizer_sl_clip<ras_conv_dbl>::line_to for ONE edge (the clipper's only state is the previous vertex)
struct EdgeClip {
    double bx1, by1, bx2, by2;
    CellSink c;
    int part, nparts;            // this thread's share of the edge's rows
    RDEV_INLINE unsigned flags(double x, double y) const {
        return (unsigned)(x > bx2) | ((unsigned)(y > by2) << 1) | ((unsigned)(x < bx1) << 2) | ((unsigned)(y < by1) << 3);
    }
    RDEV_INLINE unsigned flags_y(double y) const { return ((unsigned)(y > by2) << 1) | ((unsigned)(y < by1) << 3); }
    // line_clip_y: one piece of an edge (already clipped in x), clipped in y and handed to the cell walker
    template <int MODE> RDEV_INLINE void piece(double ax, double ay, double bx, double by, unsigned fa, unsigned fb) const {
        fa &= 10; fb &= 10;
        double tx1 = ax, ty1 = ay, tx2 = bx, ty2 = by;
        if ((fa | fb) != 0) {
            if (fa == fb) return;                          // invisible by y
            if (fa & 8) { tx1 = ax + (by1 - ay) * (bx - ax) / (by - ay); ty1 = by1; }
            if (fa & 2) { tx1 = ax + (by2 - ay) * (bx - ax) / (by - ay); ty1 = by2; }
            if (fb & 8) { tx2 = ax + (by1 - ay) * (bx - ax) / (by - ay); ty2 = by1; }
            if (fb & 2) { tx2 = ax + (by2 - ay) * (bx - ax) / (by - ay); ty2 = by2; }
        }
        cell_line<MODE>(c, iround(tx1 * SUB), iround(ty1 * SUB), iround(tx2 * SUB), iround(ty2 * SUB), part, nparts);
    }
    // One edge = up to three pieces after clipping in x (the pieces on the clip box's left / right side are kept: they
    // close the winding).  The pieces are listed first and walked by ONE loop (one inlined copy of the walker).
    template <int MODE> RDEV_INLINE void edge(double x1, double y1, double x2, double y2) const {
        const unsigned f1 = flags(x1, y1), f2 = flags(x2, y2);
        if ((f1 & 10) == (f2 & 10) && (f1 & 10) != 0) return;      // invisible by y
        // piece i runs from (px[i], py[i]) to (px[i + 1], py[i + 1]) with y-flags pf[i], pf[i + 1] (held in scalars: an
        // indexed array would live in scratch)
        double px0 = x1, py0 = y1, px1 = x2, py1 = y2, px2 = 0, py2 = 0, px3 = 0, py3 = 0;
        unsigned pf0 = f1, pf1 = f2, pf2 = 0, pf3 = 0;
        int np = 1;
        double y3, y4;
        switch (((f1 & 5) << 1) | (f2 & 5)) {
        case 0: break;
        case 1:
            y3 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1);
            px1 = bx2; py1 = y3; pf1 = flags_y(y3); px2 = bx2; py2 = y2; pf2 = f2; np = 2; break;
        case 2:
            y3 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1);
            px0 = bx2; px1 = bx2; py1 = y3; pf1 = flags_y(y3); px2 = x2; py2 = y2; pf2 = f2; np = 2; break;
        case 3: px0 = bx2; px1 = bx2; break;
        case 4:
            y3 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1);
            px1 = bx1; py1 = y3; pf1 = flags_y(y3); px2 = bx1; py2 = y2; pf2 = f2; np = 2; break;
        case 6:
            y3 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1); y4 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1);
            px0 = bx2; px1 = bx2; py1 = y3; pf1 = flags_y(y3); px2 = bx1; py2 = y4; pf2 = flags_y(y4);
            px3 = bx1; py3 = y2; pf3 = f2; np = 3; break;
        case 8:
            y3 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1);
            px0 = bx1; px1 = bx1; py1 = y3; pf1 = flags_y(y3); px2 = x2; py2 = y2; pf2 = f2; np = 2; break;
        case 9:
            y3 = y1 + (bx1 - x1) * (y2 - y1) / (x2 - x1); y4 = y1 + (bx2 - x1) * (y2 - y1) / (x2 - x1);
            px0 = bx1; px1 = bx1; py1 = y3; pf1 = flags_y(y3); px2 = bx2; py2 = y4; pf2 = flags_y(y4);
            px3 = bx2; py3 = y2; pf3 = f2; np = 3; break;
        case 12: px0 = bx1; px1 = bx1; break;
        default: return;
        }
        for (int i = 0; i < np; ++i) {
            const double ax = i == 0 ? px0 : (i == 1 ? px1 : px2), ay = i == 0 ? py0 : (i == 1 ? py1 : py2);
            const double bx = i == 0 ? px1 : (i == 1 ? px2 : px3), by = i == 0 ? py1 : (i == 1 ? py2 : py3);
            const unsigned fa = i == 0 ? pf0 : (i == 1 ? pf1 : pf2), fb = i == 0 ? pf1 : (i == 1 ? pf2 : pf3);
            piece<MODE>(ax, ay, bx, by, fa, fb);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// blend: fixed_blender_rgba_plain on an opaque grey pixel (R = G = B, A = 255); rgba8::multiply for the cover
// ---------------------------------------------------------------------------------------------------------------
RDEV_INLINE unsigned calc_alpha(int a) {
    int cover = a >> (SHIFT * 2 + 1 - 8);
    if (cover < 0) cover = -cover;
    return cover > 255 ? 255u : (unsigned)cover;
}
RDEV_INLINE unsigned cover_alpha(unsigned a8, unsigned cover) {   // rgba8::multiply(colour alpha, cover)
    const unsigned t = a8 * cover + 128;
    return ((t >> 8) + t) >> 8;
}
RDEV_INLINE unsigned blend_alpha(unsigned p, unsigned grey, unsigned alpha) {
    if (alpha == 0) return p;
    const unsigned r = p * 255u;
    const unsigned a = ((alpha + 255u) << 8) - alpha * 255u;
    return (unsigned)((((int)(grey << 8) - (int)r) * (int)alpha + (int)(r << 8)) / (int)a);
}
RDEV_INLINE unsigned blend(unsigned p, unsigned grey, unsigned a8, unsigned cover) {
    if (a8 == 255 && cover == 255) return grey;          // opaque colour at full coverage: the pixel is copied
    return blend_alpha(p, grey, cover_alpha(a8, cover));
}

}  // namespace vpk_raster
#endif
