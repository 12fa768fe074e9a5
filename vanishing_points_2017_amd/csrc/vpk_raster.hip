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
// Decomposition: three kernels (outlines per line over all CUs, coverage per line over all CUs, ordered blend per image
// row), described where they are defined below.
#include "vpk_internal.hpp"
#include <string.h>

#include "raster_device.hpp"

#include <algorithm>
#include <stdlib.h>
#include <math.h>
#include <stdlib.h>

namespace {

using namespace vpk_raster;

// ---------------------------------------------------------------------------------------------------------------
// Four kernels.
//   simplify_kernel  one WAVE per line, lanes = samples: samples -> PathSimplifier's kept points (see the kernel)
//   outline_kernel   one THREAD per line: conv_stroke on the kept points (and the sequential simplifier for the lines
//                    simplify_kernel leaves to it); the closed outline polygons (up to MAXSUB per line) go to HBM
//   coverage_kernel  one WORKGROUP per line at a time (persistent workgroups over a queue of ALL lines: the lines of an
//                    image need no order here): cells in the LDS pool, sweep -> the line's coverage as one byte per
//                    pixel of every touched row's cell range (the pool's packing) + the dense row table, in HBM
//   blend_kernel     32 image rows x 8 column segments per workgroup, the rows' pixels in LDS: the image's lines IN
//                    INPUT ORDER (the 8-bit blend does not commute), then the four spines; one coalesced store
// ---------------------------------------------------------------------------------------------------------------
// One polygon's coverage of one image row: up to two ranges of pixels (raster_device.hpp: CellSink) whose alpha bytes lie back
// to back in the call's pool -- L: pixels [xminL, xminL + lenL) at `off`, R: [xminR, xminR + lenR) at off + lenL (a length
// of 0: no such range; both 0: the row is not touched).  The table is DENSE -- [line][sub-path][row] -- so that the blend can
// fetch the entries of several lines at once without first reading anything about the lines; bits 12.. of `lenL` in
// sub-path 0 = the line's further sub-paths (a NaN sample breaks the path: rare).
struct RowEnt { unsigned off; short xminL, lenL, xminR, lenR; int pad; };
constexpr int ROW_LEN = 0xfff, ROW_MORE_SHIFT = 12;
constexpr int POLY_INTS = 1 + 3 * MAXSUB;                // per line: sub-paths, then (first vertex, vertices, split column) each

struct RasterArgs {
    const double* l; const long long* offsets; const int* order; const double* tab;
    int batch; int size; int samples; unsigned a8; long long nlines; long long line0;   // this chunk: lines [line0, line0 + nlines)
    unsigned char* out; int* ctr; unsigned* flags;       // ctr[0]: line queue, [1]: unused; 64-bit bump counters follow
    unsigned long long* bump;                            // [0]: alpha bytes used, [1]: row refs used
    V2* simp; V2* verts; int* polys;                     // per line: MAXS, MAXV, POLY_INTS
    int pool, rowcap;                                    // coverage_kernel: pool entries and row-array length in (dynamic) LDS
    int probe;                                           // VPK_RASTER_TIMES: workgroup 0 of coverage_kernel times its phases
    int* seq; int* nsimp; int force_seq;                 // per line: 1 = left to the sequential machine; points kept by simplify_kernel
    unsigned char* alpha; unsigned long long alpha_cap;
    RowEnt* dense;                                       // [nlines + 4 spines][MAXSUB][size]
    int first_image;                                     // images [first_image, first_image + batch) of the caller's batch
    int alt;                                             // 1: the reference's `alternative` curve (sphere_mapping.py:58-59)
};

// beta(alpha) of one sample, operation by operation as NumPy evaluates the reference's expression (sphere_mapping.py:59 /
// :61, then `b *= -1`, :63) -- the unit is built with -ffp-contract=off
__device__ __forceinline__ double curve_beta(double la, double lb, double lc, double sa, double ca, int alt) {
    double be = alt ? -atan(-lc / (ca * la + sa * lb))                     // :59
                    : -atan((-la * sa - lc * ca) / lb);                   // :61
    be *= -1;                                                             // :63
    return be;
}

// per sample, the same for every line: sin / cos of alpha (sphere_mapping.py:61-63) and the pixel x of the sample
__global__ void raster_table_kernel(int ns, int size, double* tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const double lo_a = -PI_D / 2, hi_a = PI_D / 2;
    const double step = (hi_a - lo_a) / (ns - 1);
    const double al = (i == ns - 1) ? hi_a : lo_a + i * step;       // numpy.linspace
    tab[4 * i] = (al - lo_a) / (hi_a - lo_a) * size;                // the axes' transform of alpha (always finite)
    tab[4 * i + 1] = sin(al);
    tab[4 * i + 2] = cos(al);
    tab[4 * i + 3] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// simplify_kernel: samples -> PathSimplifier with one WAVE per line, lanes = 64 consecutive samples.
//
// The simplifier is a sequential machine, but its state only changes shape when a run ends (about 60 times in 10 000
// samples); in between, what a vertex contributes -- its distance across and along the run -- depends on the run's
// constants alone.  So a wave evaluates 64 samples at once (the curve's beta(alpha), the axes' transform, the
// simplifier's metrics), finds the first lane that ends the run (ballot), folds the lanes before it into the state
// with two max-reductions (the farthest vertex forwards / backwards: strict `>` in sequence = the FIRST lane that
// attains the maximum, if that beats the incoming one; the two `last_*` flags = whether the last folded lane is that
// lane), performs the run's end in wave-uniform code, and goes on with the remaining lanes under the new run's constants.
// Same operations on the same operands as the vertex-by-vertex machine (raster_device.hpp: Simplifier), so the same
// points.  A line with a non-finite sample (b = 0 and friends: PathNanRemover breaks the path there) is left to the
// sequential machine in outline_kernel (seq[g] = 1).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_value(double v, int lane) {          // lane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ __launch_bounds__(64) void simplify_kernel(RasterArgs A) {
    const long long g = blockIdx.x;
    const int lane = threadIdx.x;
    const long long gl = A.line0 + g;
    const double la = A.l[3 * gl], lb = A.l[3 * gl + 1], lc = A.l[3 * gl + 2];
    const int size = A.size, ns = A.samples;
    const double lo_a = -PI_D / 2, hi_a = PI_D / 2;
    V2* out = A.simp + (size_t)g * MAXS;
    if (A.force_seq) { if (lane == 0) A.seq[g] = 1; return; }
    // the machine's state (every lane holds the same values)
    const double thr2 = (1.0 / 9.0) * (1.0 / 9.0);
    double lastx = 0, lasty = 0, origdx = 0, origdy = 0, orig_norm2 = 0, fwd_max = 0, bwd_max = 0, nextx = 0, nexty = 0, nbx = 0,
           nby = 0, startx = 0, starty = 0;
    bool last_fwd = false, last_bwd = false, clipped = true, started = false;
    int n = 0, overflow = 0;
    auto emit = [&](double x, double y) {
        if (n < MAXS) { if (lane == 0) { out[n].x = x; out[n].y = y; } ++n; } else overflow = 1;
    };
    for (int c0 = 0; c0 < ns; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < ns;
        const int ic = valid ? i : ns - 1;
        const double sa = A.tab[4 * ic + 1], ca = A.tab[4 * ic + 2], x = A.tab[4 * ic];
        const double be = curve_beta(la, lb, lc, sa, ca, A.alt);
        const double y = size - (be - lo_a) / (hi_a - lo_a) * size;
        const unsigned long long bad = __ballot(valid && (!(y == y) || isinf(y)));
        unsigned long long pending = __ballot(valid);
        if (bad != 0ull) {
            // PathNanRemover's business.  A line whose samples are ALL non-finite (the all-zero line: 0 / 0 everywhere)
            // draws nothing and is settled here; any other mixture goes to the sequential machine.
            if (bad == pending && !started) continue;
            if (lane == 0) A.seq[g] = 1;
            return;
        }
        if (!started) {                                                       // move_to (the first finite sample)
            if (c0 != 0) { if (lane == 0) A.seq[g] = 1; return; }             // (finite samples after non-finite ones: a broken path)
            lastx = lane_value(x, 0); lasty = lane_value(y, 0);
            orig_norm2 = 0.0; bwd_max = 0.0; clipped = true; started = true;
            pending &= ~1ull;
        }
        while (pending != 0ull) {
            if (uniform(orig_norm2 == 0.0)) {                                 // the run's first segment
                const int j = uniform(__builtin_ctzll(pending));
                const double px = lane_value(x, j), py = lane_value(y, j);
                if (clipped) { emit(lastx, lasty); clipped = false; }
                origdx = px - lastx; origdy = py - lasty;
                orig_norm2 = origdx * origdx + origdy * origdy;
                fwd_max = orig_norm2; bwd_max = 0.0; last_fwd = true; last_bwd = false;
                startx = lastx; starty = lasty;
                nextx = lastx = px; nexty = lasty = py;
                pending &= ~(1ull << j);
                continue;
            }
            // metrics of every lane's vertex under the run's constants (Simplifier::metrics)
            const double totdx = x - startx, totdy = y - starty;
            const double totdot = origdx * totdx + origdy * totdy;
            const double paradx = totdot * origdx / orig_norm2, parady = totdot * origdy / orig_norm2;
            const double perpdx = totdx - paradx, perpdy = totdy - parady;
            const double perp2 = perpdx * perpdx + perpdy * perpdy;
            const double para2 = paradx * paradx + parady * parady;
            const bool mine = (pending >> lane) & 1ull;
            const unsigned long long brk = __ballot(mine && !(perp2 < thr2));
            const int jb = brk ? uniform(__builtin_ctzll(brk)) : 64;
            const unsigned long long inrun = jb < 64 ? (pending & ((1ull << jb) - 1ull)) : pending;
            if (inrun != 0ull) {
                const bool in = (inrun >> lane) & 1ull;
                const int jl = uniform(63 - __builtin_clzll(inrun));          // the last vertex folded in
                const bool fwd_el = in && totdot > 0.0, bwd_el = in && !(totdot > 0.0);
                bool lf = false, lbk = false;
                // (a curve usually runs on along its run: the last folded lane is the farthest one forwards, and one ballot
                //  says so -- "no eligible lane reaches its para2" -- without the reduction)
                const double pl = lane_value(para2, jl);
                const bool jl_fwd = (__ballot(fwd_el) >> jl) & 1ull;
                if (jl_fwd && __ballot(fwd_el && para2 >= pl) == (1ull << jl)) {
                    if (uniform(pl > fwd_max)) { fwd_max = pl; nextx = lane_value(x, jl); nexty = lane_value(y, jl); lf = true; }
                } else if (__ballot(fwd_el) != 0ull) {
                    const double m = wave_max(fwd_el ? para2 : -1.0);
                    if (uniform(m > fwd_max)) {
                        const int jf = uniform(__builtin_ctzll(__ballot(fwd_el && para2 == m)));
                        fwd_max = m; nextx = lane_value(x, jf); nexty = lane_value(y, jf);
                        lf = jf == jl;
                    }
                }
                if (__ballot(bwd_el) != 0ull) {
                    const double m = wave_max(bwd_el ? para2 : -1.0);
                    if (uniform(m > bwd_max)) {
                        const int jk = uniform(__builtin_ctzll(__ballot(bwd_el && para2 == m)));
                        bwd_max = m; nbx = lane_value(x, jk); nby = lane_value(y, jk);
                        lbk = jk == jl;
                    }
                }
                last_fwd = lf; last_bwd = lbk;
                lastx = lane_value(x, jl); lasty = lane_value(y, jl);
            }
            if (jb < 64) {                                                    // _push: the run ends at lane jb's vertex
                const double px = lane_value(x, jb), py = lane_value(y, jb);
                const bool both = bwd_max > 0.0;
                const double ax = both && last_fwd ? nbx : nextx, ay = both && last_fwd ? nby : nexty;
                const double cx = last_fwd ? nextx : nbx, cy = last_fwd ? nexty : nby;
                emit(ax, ay);
                if (both) emit(cx, cy);
                double ex = both ? cx : ax, ey = both ? cy : ay;
                if (clipped || (!last_fwd && !last_bwd)) { emit(lastx, lasty); ex = lastx; ey = lasty; }
                origdx = px - lastx; origdy = py - lasty;
                orig_norm2 = origdx * origdx + origdy * origdy;
                fwd_max = orig_norm2; last_fwd = true;
                startx = ex; starty = ey;
                lastx = nextx = px; lasty = nexty = py;
                bwd_max = 0.0; last_bwd = false; clipped = false;
                pending &= ~((2ull << jb) - 1ull);
            } else {
                pending = 0ull;
            }
        }
    }
    // path_cmd_stop
    if (started) {
        if (orig_norm2 != 0.0) {
            emit(nextx, nexty);
            if (bwd_max > 0.0) emit(nbx, nby);
        }
        emit(lastx, lasty);
    }
    if (lane == 0) { A.seq[g] = 0; A.nsimp[g] = n | (overflow ? 0x40000000 : 0); }
}

__global__ __launch_bounds__(64) void outline_kernel(RasterArgs A) {
    const long long g = (long long)blockIdx.x * 64 + threadIdx.x;     // line of this chunk; the 4 spines follow the lines
    if (g >= A.nlines + 4) return;
    V2* sp = A.simp + (size_t)g * MAXS;
    Outline o;
    unsigned dummy = 0;
    unsigned* fl = &dummy;
    o.v = A.verts + (size_t)g * MAXV; o.n = 0; o.cap = MAXV; o.flags = fl;
    int* pt = A.polys + g * POLY_INTS;
    const int size = A.size;
    // the column at which coverage_kernel splits the rows' cells in two ranges: the curve's interior extremum (a row beside
    // it is crossed twice, far apart); any column is correct, this one saves the most.  From the kept points of the path.
    auto split_column = [&](const V2* p, int n) {
        int imin = 0, imax = 0;
        for (int k = 1; k < n; ++k) { if (p[k].y < p[imin].y) imin = k; if (p[k].y > p[imax].y) imax = k; }
        const double xa = p[imin].x, xb = p[imax].x;
        const double da = xa < size - xa ? xa : size - xa, db = xb < size - xb ? xb : size - xb;   // distance from the canvas's sides
        const double x = da >= db ? xa : xb;
        const int c = (int)floor(x);
        return c < 0 ? 0 : (c > size ? size : c);
    };
    if (g >= A.nlines) {
        // the axes' spines: left, right, bottom, top (matplotlib's drawing order): two-vertex rectilinear paths, snapped to
        // pixel centres (PathSnapper: floor(v + 0.5) + 0.5 for a stroke whose width rounds to an odd number of pixels)
        const int side = (int)(g - A.nlines);
        const double s = (double)size, w_spine = 0.8 * 100.0 / 72.0;
        const double x0 = (side == 1) ? s : 0.0, y0 = (side == 3) ? 0.0 : s;
        const double x1 = (side == 0) ? 0.0 : s, y1 = (side == 2) ? s : 0.0;
        sp[0].x = floor(x0 + 0.5) + 0.5; sp[0].y = floor(y0 + 0.5) + 0.5;
        sp[1].x = floor(x1 + 0.5) + 0.5; sp[1].y = floor(y1 + 0.5) + 0.5;
        stroke_outline(sp, 2, w_spine, o);
        pt[0] = 1; pt[1] = 0; pt[2] = o.n; pt[3] = size + 2;        // (a spine: one range per row)
        return;
    }
    const double width_px = 100.0 / 72.0;                 // 1 pt at 100 dpi (matplotlib 1.5.1's default line width)
    const long long gl = A.line0 + g;
    int npoly = 0;
    Simplifier sm;
    sm.init(sp, MAXS);
    const int ns = A.samples;
    auto flush = [&]() {                                  // end of a sub-path: stroke what the simplifier kept
        sm.end();
        if (sm.n >= 2) {
            const int first = o.n;
            const int xsp = split_column(sp, sm.n);
            stroke_outline(sp, sm.n, width_px, o);
            if (o.n - first >= 3) {
                if (npoly < MAXSUB) { pt[1 + 3 * npoly] = first; pt[2 + 3 * npoly] = o.n - first; pt[3 + 3 * npoly] = xsp; ++npoly; }
                else dummy |= FLAG_OVERFLOW;
            }
        }
        sm.n = 0;
    };
    if (A.seq[g]) {
        // the sequential machine (lines with non-finite samples; VPK_RASTER_SEQUENTIAL=1: every line): OG samples are
        // evaluated side by side (independent division / atan chains), then fed to the simplifier as a group
        const double la = A.l[3 * gl], lb = A.l[3 * gl + 1], lc = A.l[3 * gl + 2];
        const double lo_a = -PI_D / 2, hi_a = PI_D / 2;
        constexpr int OG = 4;
        for (int i0 = 0; i0 < ns; i0 += OG) {
            double xs[OG], ys[OG];
#pragma unroll
            for (int u = 0; u < OG; ++u) {
                const int i = i0 + u < ns ? i0 + u : ns - 1;
                const double sa = A.tab[4 * i + 1], ca = A.tab[4 * i + 2];
                const double be = curve_beta(la, lb, lc, sa, ca, A.alt);
                xs[u] = A.tab[4 * i];
                ys[u] = size - (be - lo_a) / (hi_a - lo_a) * size;
            }
            feed_group<OG>(sm, xs, ys, ns - i0, flush);
        }
    } else {
        // simplify_kernel kept the line's points: one finished path
        const int k = A.nsimp[g];
        sm.n = k & 0xffff;
        if (k & 0x40000000) dummy |= FLAG_OVERFLOW;
        sm.have = false;
        if (sm.n >= 2) {
            const int xsp = split_column(sp, sm.n);
            stroke_outline(sp, sm.n, width_px, o);
            if (o.n >= 3) { pt[1] = 0; pt[2] = o.n; pt[3] = xsp; npoly = 1; }
        }
        sm.n = 0;
    }
    if (sm.have) flush();
    pt[0] = npoly;
    dummy |= sm.overflow;
    if (dummy) {                                          // which image the line belongs to: binary search in the offsets
        int lo = 0, hi = A.batch;
        while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (A.offsets[A.first_image + mid] <= gl) lo = mid; else hi = mid; }
        atomicOr(A.flags + A.first_image + lo, dummy);
    }
}

// cells of one polygon -> coverage bytes: bounds pass, row scan, cells into the LDS pool, sweep -> alpha bytes in the pool's
// packing.  A workgroup spends most of a polygon waiting -- on barriers and on memory round trips -- so the round trips
// are kept off the critical path: the vertices are fetched once (LDS), a thread keeps its work item between the two
// passes, the pools' space comes from ONE atomic, the next line's number is requested a polygon ahead.
constexpr int LVERT = 256;                               // vertices of a polygon held in LDS (typical: 60-200)
struct CovShared {
    int* total; long long* base; unsigned short* eoff; V2* vert; int* tail;
};
__device__ bool polygon_coverage(const V2* v, int n, int xsplit, const CellSink& sink, int size, const CovShared& S, const RasterArgs& A,
                                 RowEnt* ent, int more) {
    const bool probe = A.probe && threadIdx.x == 0 && blockIdx.x == 0;
    long long tp = probe ? wall_clock64() : 0;
    auto lap = [&](int slot) { if (probe) { const long long now = wall_clock64(); atomicAdd(A.ctr + 8 + slot, (int)(now - tp)); tp = now; } };
    unsigned short* s_eoff = S.eoff;
    const int POOL = A.pool;
    EdgeClip ec;
    ec.bx1 = 0.0; ec.by1 = 0.0; ec.bx2 = (double)size; ec.by2 = (double)size; ec.c = sink;
    ec.c.xs = xsplit;
    // Work items = (edge, share of its rows).  An outline has ~120 edges of which most cross one to three rows and a few
    // thirty or more; an item costs the clipper's and the walker's set-up (divisions in double and in int) before its first
    // cell, so an edge gets one share per ROWS_PER_ITEM rows it crosses -- not a fixed number of shares (with 4 RT / n
    // shares per edge, 9 in 10 items were empty and the set-up was most of the kernel).  The shares' offsets: one wave's scan.
    constexpr int ROWS_PER_ITEM = 4;
    const bool inlds = n <= LVERT;
    for (int k = threadIdx.x; k < n; k += RT) {
        const V2 a = v[k];
        const double ya = a.y, yb = v[k + 1 < n ? k + 1 : 0].y;
        if (inlds) { S.vert[k] = a; if (k == 0) S.vert[n] = a; }
        double lo = ya < yb ? ya : yb, hi = ya < yb ? yb : ya;
        lo = lo > 0.0 ? lo : 0.0;
        hi = hi < (double)size ? hi : (double)size;
        const int rows = hi >= lo ? (int)hi - (int)lo + 1 : 1;      // (an estimate is enough: any split gives the same cells)
        int ke = (rows + ROWS_PER_ITEM - 1) / ROWS_PER_ITEM;
        s_eoff[k] = (unsigned short)(ke > 32 ? 32 : ke);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int per = (n + 63) / 64, k0 = threadIdx.x * per;
        int sum = 0;
        for (int k = k0; k < k0 + per && k < n; ++k) sum += s_eoff[k];
        int incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += up;
        }
        int off = incl - sum;
        for (int k = k0; k < k0 + per && k < n; ++k) { const int ke = s_eoff[k]; s_eoff[k] = (unsigned short)off; off += ke; }
        if (threadIdx.x == 63) s_eoff[n] = (unsigned short)incl;
    }
    __syncthreads();
    const int items = s_eoff[n];
    struct Work { int part, nparts; V2 a, b; };
    auto work_item = [&](int it) {                        // the edge whose shares contain item `it`: last k with eoff[k] <= it
        int lo = 0, hi = n;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)s_eoff[mid] <= it) lo = mid; else hi = mid; }
        Work w;
        w.part = it - s_eoff[lo];
        w.nparts = s_eoff[lo + 1] - s_eoff[lo];
        if (inlds) { w.a = S.vert[lo]; w.b = S.vert[lo + 1]; }
        else { w.a = v[lo]; w.b = v[lo + 1 < n ? lo + 1 : 0]; }
        return w;
    };
    Work mine = {};                                       // this thread's first item: the same in both passes
    if ((int)threadIdx.x < items) mine = work_item(threadIdx.x);
    for (int it = threadIdx.x; it < items; it += RT) {    // pass 1: the rows' cell ranges (L and R of the split column)
        const Work w = it == (int)threadIdx.x ? mine : work_item(it);
        ec.part = w.part; ec.nparts = w.nparts;
        ec.edge<BOUNDS>(w.a.x, w.a.y, w.b.x, w.b.y);
    }
    __syncthreads();
    lap(0);
    // A row's pool entries: its L range then its R range, or -- if they touch, or if something is to be drawn between them
    // (lcov != 0) -- ONE range from the first L cell to the last R cell.  lcov becomes the number of L entries (-1: joined).
    auto row_entries = [&](int y) {
        const int sp = sink.lcov[y];
        return sp < 0 ? sink.rmax[y] - sink.rowmin[y] + 1 : sp + row_len(sink.rmin[y], sink.rmax[y]);
    };
    // exclusive prefix sum of the rows' entries (one wave), first / last touched row
    if (threadIdx.x < 64) {
        const int per = (size + 63) / 64;
        const int y0 = threadIdx.x * per;
        int sum = 0, ymin = 0x7fffffff, ymax = -1;
        for (int y = y0; y < y0 + per && y < size; ++y) {
            const int ll = row_len(sink.rowmin[y], sink.rowmax[y]), rl = row_len(sink.rmin[y], sink.rmax[y]);
            const bool joined = ll > 0 && rl > 0 && (sink.rowmax[y] + 1 >= sink.rmin[y] || sink.lcov[y] != 0);
            sink.lcov[y] = joined ? -1 : ll;
            const int len = joined ? sink.rmax[y] - sink.rowmin[y] + 1 : ll + rl;
            if (len) { ymin = ymin < y ? ymin : y; ymax = y; }
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
            off += row_entries(y);
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int a = __shfl_xor(ymin, o), b = __shfl_xor(ymax, o);
            ymin = a < ymin ? a : ymin;
            ymax = b > ymax ? b : ymax;
        }
        if (threadIdx.x == 63) {
            S.total[0] = incl; S.total[1] = ymin; S.total[2] = ymax;
            long long ab = -1;
            if (incl > 0) {
                // space in the call's coverage pool: one atomic add; a polygon that does not fit gives its bytes back, so that
                // it alone is dropped (and its image flagged), not every polygon that asks after it.  (A compare-and-swap loop
                // on the one counter, 512 workgroups contending, cost 100 ms per call.)
                const unsigned long long old = atomicAdd(A.bump, (unsigned long long)incl);
                if (old + (unsigned long long)incl > A.alpha_cap) atomicAdd(A.bump, 0ull - (unsigned long long)incl);
                else ab = (long long)old;
            }
            S.base[0] = ab;
        }
    }
    __syncthreads();
    lap(1);
    const int total = S.total[0], ymin = S.total[1], ymax = S.total[2];
    const long long ab = S.base[0];
    const bool ok = ab >= 0;                              // (a polygon past the call's HBM pool is dropped and flagged)
    // The LDS pool holds the entries of a BAND of rows at a time: all touched rows when they fit (the usual case), else as
    // many consecutive rows as fit.
    int ys = ok ? ymin : size;
    while (ys <= ymax && ys < size) {
        int ye = ys + 1;                                  // (uniform: every thread walks the same prefix sums)
        const int boff = sink.rowoff[ys];
        if (total <= POOL) ye = ymax + 1;                  // everything fits: one band
        else while (ye <= ymax && sink.rowoff[ye] + row_entries(ye) - boff <= POOL) ++ye;
        CellSink band = ec.c;
        band.blo = ys; band.bhi = ye; band.boff = boff;
        ec.c = band;
        for (int it = threadIdx.x; it < items; it += RT) {    // pass 2: the cells of the band's rows
            const Work w = it == (int)threadIdx.x ? mine : work_item(it);
            ec.part = w.part; ec.nparts = w.nparts;
            ec.edge<POOLED>(w.a.x, w.a.y, w.b.x, w.b.y);
        }
        __syncthreads();
        lap(2);
        // The sweep (the per-pixel form of sweep_scanline, see blend_kernel) over the band's pool entries, which are the
        // rows' entries back to back: every thread takes an equal run of consecutive entries -- not a row: rows are 1
        // to 500 entries long --, the running cover at the start of its run being the covers of its row before it.  (The
        // running cover simply continues from a row's L entries into its R entries: it is 0 across a dropped gap.)
        const int last = ye - 1;
        const int bend = sink.rowoff[last] + row_entries(last);
        const int E = bend - boff, C = (E + RT - 1) / RT;
        const int q0 = threadIdx.x * C, q1 = q0 + C < E ? q0 + C : E;
        int y = ys, row_lo = 0, row_hi = 0;               // the row of entry q0: pool entries [row_lo, row_hi)
        auto row_span = [&](int yy) { row_lo = sink.rowoff[yy] - boff; row_hi = row_lo + row_entries(yy); };
        if (q0 < q1) {
            int lo = ys, hi = ye;                         // last row whose offset is <= q0 (an empty row shares its successor's)
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sink.rowoff[mid] - boff <= q0) lo = mid; else hi = mid; }
            y = lo;
            row_span(y);
            int tail = 0, started = 0, yy = y, rhi = row_hi;
            for (int q = q0; q < q1; ++q) {
                if (q == rhi) {                           // the next non-empty row starts at this entry
                    do { ++yy; rhi = sink.rowoff[yy] - boff + row_entries(yy); } while (q == rhi);
                    tail = 0; started = 1;
                }
                tail += sink.pcover[q];
            }
            if (q0 == row_lo) started = 1;
            S.tail[2 * threadIdx.x] = tail; S.tail[2 * threadIdx.x + 1] = started;
        }
        __syncthreads();
        if (q0 < q1) {
            int R = 0;
            if (q0 != row_lo)
                for (int u = (int)threadIdx.x - 1; u >= 0; --u) { R += S.tail[2 * u]; if (S.tail[2 * u + 1]) break; }
            for (int q = q0; q < q1; ++q) {
                if (q == row_hi) { do { ++y; row_span(y); } while (q == row_hi); R = 0; }
                const int c = sink.pcover[q], a = sink.parea[q];
                sink.parea[q] = 0;
                R += c;
                unsigned al = 0;
                if (a) al = calc_alpha((R << (SHIFT + 1)) - a);
                else if (q + 1 < row_hi) al = calc_alpha(R << (SHIFT + 1));
                sink.pcover[q] = (int)al;                 // the alphas stay in the pool ...
            }
        }
        for (int yy = ys + threadIdx.x; yy < ye; yy += RT) {
            const int sp = sink.lcov[yy];
            const int ll = sp < 0 ? sink.rmax[yy] - sink.rowmin[yy] + 1 : sp, rl = sp < 0 ? 0 : row_len(sink.rmin[yy], sink.rmax[yy]);
            RowEnt r;
            r.off = (unsigned)(ab + sink.rowoff[yy]);
            r.xminL = (short)(ll ? sink.rowmin[yy] - 1 : 0); r.lenL = (short)(ll | more);
            r.xminR = (short)(rl ? sink.rmin[yy] - 1 : 0); r.lenR = (short)rl;
            r.pad = 0;
            ent[yy] = r;
        }
        __syncthreads();
        {   // ... and leave it in one coalesced copy (the pool is zero again afterwards)
            unsigned char* dst = A.alpha + ab + boff;
            for (int q = threadIdx.x; q < E; q += RT) { dst[q] = (unsigned char)sink.pcover[q]; sink.pcover[q] = 0; }
        }
        __syncthreads();
        lap(3);
        ys = ye;
    }
    for (int y = threadIdx.x; y < size; y += RT) {
        if (!ok || y < ymin || y > ymax) {                // rows the polygon leaves alone
            RowEnt r;
            r.off = 0; r.xminL = 0; r.lenL = (short)more; r.xminR = 0; r.lenR = 0; r.pad = 0;
            ent[y] = r;
        }
        sink.rowmin[y] = 0x7fffffff; sink.rowmax[y] = -1; sink.rmin[y] = 0x7fffffff; sink.rmax[y] = -1; sink.lcov[y] = 0;
    }
    __syncthreads();
    return ok || total == 0;
}

__global__ __launch_bounds__(RT, 3) void coverage_kernel(RasterArgs A) {   // three workgroups (of four waves) per CU
    extern __shared__ int s_dyn[];                       // six row arrays of A.rowcap ints, then the pool (cover, area)
    __shared__ int s_next, s_total[4];
    __shared__ unsigned short s_eoff[MAXV + 2];          // first work item of every edge of the polygon at hand
    __shared__ V2 s_vert[LVERT + 1];                     // its vertices (closed: [n] = [0])
    __shared__ long long s_base[2];
    __shared__ int s_tail[2 * RT];                       // the sweep's per-thread partial sums
    const int size = A.size, rc = A.rowcap;
    int* s_rowmin = s_dyn; int* s_rowmax = s_dyn + rc; int* s_rowoff = s_dyn + 2 * rc;
    int* s_rmin = s_dyn + 3 * rc; int* s_rmax = s_dyn + 4 * rc; int* s_lcov = s_dyn + 5 * rc;
    int* s_pcover = s_dyn + 6 * rc; int* s_parea = s_pcover + A.pool;
    CellSink sink;
    sink.cover = nullptr; sink.area = nullptr;
    sink.rowmin = (lds_int_ptr)s_rowmin; sink.rowmax = (lds_int_ptr)s_rowmax; sink.rowoff = (lds_int_ptr)s_rowoff; sink.size = size;
    sink.rmin = (lds_int_ptr)s_rmin; sink.rmax = (lds_int_ptr)s_rmax; sink.lcov = (lds_int_ptr)s_lcov; sink.xs = 0;
    sink.pcover = (lds_int_ptr)s_pcover; sink.parea = (lds_int_ptr)s_parea;
    sink.blo = 0; sink.bhi = 0; sink.boff = 0;
    CovShared S;
    S.total = s_total; S.base = s_base; S.eoff = s_eoff; S.vert = s_vert; S.tail = s_tail;
    for (int y = threadIdx.x; y < rc; y += RT) {
        s_rowmin[y] = 0x7fffffff; s_rowmax[y] = -1; s_rowoff[y] = 0; s_rmin[y] = 0x7fffffff; s_rmax[y] = -1; s_lcov[y] = 0;
    }
    for (int q = threadIdx.x; q < A.pool; q += RT) { s_pcover[q] = 0; s_parea[q] = 0; }   // the sweeps leave the pool zero again
    if (threadIdx.x == 0) s_next = atomicAdd(A.ctr, 1);
    __syncthreads();
    for (;;) {
        const long long g = __builtin_amdgcn_readfirstlane(s_next);
        __syncthreads();
        if (g >= A.nlines + 4) break;
        int nx = 0;
        if (threadIdx.x == 0) nx = atomicAdd(A.ctr, 1);  // the line after this one: the round trip runs beside the work below
        const int* pt = A.polys + g * POLY_INTS;
        const int npoly = pt[0];
        const int more = (npoly > 1 ? npoly - 1 : 0) << ROW_MORE_SHIFT;
        for (int q = 0; q < (npoly > 0 ? npoly : 1); ++q) {
            RowEnt* ent = A.dense + ((size_t)g * MAXSUB + q) * size;
            if (q < npoly) {
                const bool ok = polygon_coverage(A.verts + (size_t)g * MAXV + pt[1 + 3 * q], pt[2 + 3 * q], pt[3 + 3 * q], sink, size, S,
                                                 A, ent, q == 0 ? more : 0);
                if (!ok && threadIdx.x == 0 && g < A.nlines) {                      // dropped: tell the line's image
                    int lo = 0, hi = A.batch;
                    const long long gl = A.line0 + g;
                    while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (A.offsets[A.first_image + mid] <= gl) lo = mid; else hi = mid; }
                    atomicOr(A.flags + A.first_image + lo, FLAG_OVERFLOW);
                }
            } else {                                      // a line that left no polygon: an empty sub-path 0
                for (int y = threadIdx.x; y < size; y += RT) {
                    RowEnt r;
                    r.off = 0; r.xminL = 0; r.lenL = 0; r.xminR = 0; r.lenR = 0; r.pad = 0;
                    ent[y] = r;
                }
            }
        }
        if (threadIdx.x == 0) s_next = nx;
        __syncthreads();
    }
}

// The blend.  sweep_scanline + render_scanline_aa_solid per pixel: an entry of a row's cell range with area got
// calculate_alpha((R << 9) - area), one without (no cell, or a cell of a vertical edge on the pixel boundary) lies in the
// span that runs to the next cell and got calculate_alpha(R << 9) -- 0 after the row's last cell --, R = running cover;
// coverage_kernel stored those alphas.  Here: fixed_blender_rgba_plain, item after item.
constexpr int BROWS = 32, BSEG = 8;                       // image rows per workgroup x column segments per row: 256 threads
                                                         // (measured, 102 / 95 images: 8 x 8 0.66 / 1.59 ms, 16 x 8 0.77 / 1.47,
                                                         //  32 x 8 0.74 / 1.37)
constexpr unsigned BLEND_LUT_MAX_A8 = 63;                // colour alphas up to this blend through a table (17 KB of LDS at most)
__global__ __launch_bounds__(BROWS * BSEG) void blend_kernel(RasterArgs A) {
    extern __shared__ unsigned char s_px[];              // [BROWS][size rounded to 4], then the tables
    const int size = A.size, ldp = (size + 3) & ~3;
    const int rr = threadIdx.x & (BROWS - 1), seg = threadIdx.x / BROWS;
    const int img_i = A.order[A.first_image + blockIdx.y], y = blockIdx.x * BROWS + rr;   // images with many lines first
    unsigned char* row = s_px + (size_t)rr * ldp;
    // The blend of a white line is a function of (pixel, cover) with few distinct colour alphas (26 * cover / 255 = 0..26
    // at the reference's alpha 0.1): two table reads instead of the blender's multiplications and its division --
    //   amul[cover] = multiply(a8, cover),   lut[alpha][pixel] = fixed_blender_rgba_plain(pixel, white, alpha)
    unsigned char* amul = s_px + (size_t)BROWS * ldp;
    unsigned char* lut = amul + 256;
    const bool tables = A.a8 <= BLEND_LUT_MAX_A8;
    if (tables) {
        for (int q = threadIdx.x; q < 256; q += BROWS * BSEG) amul[q] = (unsigned char)cover_alpha(A.a8, (unsigned)q);
        for (int q = threadIdx.x; q < (int)(A.a8 + 1) * 256; q += BROWS * BSEG)
            lut[q] = (unsigned char)blend_alpha((unsigned)q & 255u, 255u, (unsigned)q >> 8);
    }
    // a thread owns the pixels [x0, x1) of its row: rows with long cell ranges (curves running along the row) are shared
    // by the waves, and a workgroup keeps them all in flight against the latency of the item -> row -> alpha loads
    const int segw = ((size + BSEG - 1) / BSEG + 3) & ~3;
    const int x0 = seg * segw, x1 = x0 + segw < size ? x0 + segw : size;
    for (int x = x0; x < x0 + segw && x < ldp; x += 4) *reinterpret_cast<unsigned*>(row + x) = 0u;
    const long long lo = A.offsets[A.first_image + img_i] - A.line0, hi = A.offsets[A.first_image + img_i + 1] - A.line0;
    __syncthreads();
    if (y < size) {
        const RowEnt* dl = A.dense + y;                   // this row's entries: dl[(line * MAXSUB + sub-path) * size]
        struct Range { unsigned off; int xmin, len; };    // one range of an entry: pixels [xmin, xmin + len), bytes at off
        // eight consecutive pixels.  Through the tables every step is eight independent LDS reads (old pixels, colour alphas,
        // new pixels: alpha 0 maps a pixel to itself), then the writes -- of the pixels with coverage only: a thread must not
        // touch its neighbours' columns, not even to write back what it read.  (Eight `if (cover) pixel = ...` in a row
        // are eight dependent read chains.)
        auto blend8 = [&](const unsigned (&av)[8], int xbase, unsigned grey, unsigned a8, bool use_tables) {
            if (use_tables) {
                unsigned old[8], ca[8], nv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { old[u] = row[xbase + u]; ca[u] = amul[av[u]]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) nv[u] = lut[(ca[u] << 8) | old[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (av[u]) row[xbase + u] = (unsigned char)nv[u];
                return;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = xbase + u;
                if (av[u]) row[x] = (unsigned char)blend(row[x], grey, a8, av[u]);
            }
        };
        auto seg_range = [&](const Range& r, int& qlo, int& qhi) {    // the range's bytes that fall into this thread's columns
            qlo = x0 - r.xmin > 0 ? x0 - r.xmin : 0;
            qhi = x1 - r.xmin < r.len ? x1 - r.xmin : r.len;
        };
        // Eight alpha bytes are ONE unaligned 8-byte load (the hardware takes unaligned global addresses; up to seven bytes
        // past the range's end are read and masked away: the pool has slack behind it).  Loads are issued unconditionally
        // -- an empty range reads the pool's first bytes -- so that the compiler can COUNT the loads in flight: a skipped
        // load makes every later wait a wait for everything.
        typedef unsigned long long __attribute__((aligned(1))) u64u;
        auto fetch8 = [&](const Range& r, int q0, int qhi) {
            return *reinterpret_cast<const u64u*>(A.alpha + (q0 < qhi ? (size_t)r.off + q0 : 0));
        };
        auto unpack8 = [&](unsigned long long w, int q0, int qhi, unsigned (&av)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = q0 + u < qhi ? (unsigned)(w >> (8 * u)) & 255u : 0u;
        };
        auto rest = [&](const Range& r, int qlo, int qhi, unsigned grey, unsigned a8, bool use_tables) {   // bytes past the first eight
            for (int q0 = qlo + 8; q0 < qhi; q0 += 8) {
                unsigned av[8];
                unpack8(fetch8(r, q0, qhi), q0, qhi, av);
                blend8(av, r.xmin + q0, grey, a8, use_tables);
            }
        };
        auto whole = [&](const Range& r, unsigned grey, unsigned a8, bool use_tables) {
            int qlo, qhi;
            seg_range(r, qlo, qhi);
            if (qlo < qhi) {
                unsigned av[8];
                unpack8(fetch8(r, qlo, qhi), qlo, qhi, av);
                blend8(av, r.xmin + qlo, grey, a8, use_tables);
                rest(r, qlo, qhi, grey, a8, use_tables);
            }
        };
        auto range_l = [&](const RowEnt& e) { Range r; r.off = e.off; r.xmin = e.xminL; r.len = e.lenL & ROW_LEN; return r; };
        auto range_r = [&](const RowEnt& e) { Range r; r.off = e.off + (unsigned)(e.lenL & ROW_LEN); r.xmin = e.xminR; r.len = e.lenR; return r; };
        // The image's lines in order -- the blend does not commute -- but FETCHED ahead: the dense table needs no knowledge
        // about a line to address its entry, so while group k (BL lines, two ranges each) is blended (LDS), the first eight
        // alpha bytes of group k + 1's ranges and the entries of group k + 2 are in flight.
        constexpr int BL = 2, NR = 2 * BL;
        const long long last = hi - 1;
        const RowEnt none = {0u, 0, 0, 0, 0, 0};
        auto entries = [&](long long g0, RowEnt (&e)[BL]) {
#pragma unroll
            for (int u = 0; u < BL; ++u) {
                const long long g = g0 + u < hi ? g0 + u : (last > lo ? last : lo);     // (clamped: masked below)
                e[u] = dl[(size_t)g * MAXSUB * size];
            }
        };
        auto settle = [&](long long g0, RowEnt (&e)[BL]) {
#pragma unroll
            for (int u = 0; u < BL; ++u) if (!(g0 + u < hi)) e[u] = none;
        };
        auto ranges_of = [&](const RowEnt (&e)[BL], Range (&r)[NR]) {
#pragma unroll
            for (int u = 0; u < BL; ++u) { r[2 * u] = range_l(e[u]); r[2 * u + 1] = range_r(e[u]); }
        };
        RowEnt e0[BL], e1[BL], e2[BL];
        Range r0[NR], r1[NR];
        unsigned long long w0[NR], w1[NR];
        int ql0[NR], qh0[NR], ql1[NR], qh1[NR];
        if (lo < hi) {
            entries(lo, e0); entries(lo + BL, e1);
            settle(lo, e0);
            ranges_of(e0, r0);
#pragma unroll
            for (int u = 0; u < NR; ++u) { seg_range(r0[u], ql0[u], qh0[u]); w0[u] = fetch8(r0[u], ql0[u], qh0[u]); }
        }
        for (long long g0 = lo; g0 < hi; g0 += BL) {
            entries(g0 + 2 * BL, e2);
            settle(g0 + BL, e1);
            ranges_of(e1, r1);
#pragma unroll
            for (int u = 0; u < NR; ++u) { seg_range(r1[u], ql1[u], qh1[u]); w1[u] = fetch8(r1[u], ql1[u], qh1[u]); }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                if (ql0[u] < qh0[u]) {
                    unsigned av[8];
                    unpack8(w0[u], ql0[u], qh0[u], av);
                    if (tables) { blend8(av, r0[u].xmin + ql0[u], 255u, A.a8, true); rest(r0[u], ql0[u], qh0[u], 255u, A.a8, true); }
                    else { blend8(av, r0[u].xmin + ql0[u], 255u, A.a8, false); rest(r0[u], ql0[u], qh0[u], 255u, A.a8, false); }
                }
                if (u & 1) {                              // after a line's second range: its further sub-paths (rare)
                    const int more = (e0[u >> 1].lenL >> ROW_MORE_SHIFT) & 7;
                    for (int q = 1; q <= more; ++q) {
                        const RowEnt f = dl[((size_t)(g0 + (u >> 1)) * MAXSUB + q) * size];
                        if (tables) { whole(range_l(f), 255u, A.a8, true); whole(range_r(f), 255u, A.a8, true); }
                        else { whole(range_l(f), 255u, A.a8, false); whole(range_r(f), 255u, A.a8, false); }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BL; ++u) { e0[u] = e1[u]; e1[u] = e2[u]; }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                r0[u] = r1[u]; ql0[u] = ql1[u]; qh0[u] = qh1[u]; w0[u] = w1[u];
            }
        }
        for (int side = 0; side < 4; ++side) {
            const RowEnt f = dl[(size_t)(A.nlines + side) * MAXSUB * size];
            whole(range_l(f), 0u, 255u, false);
            whole(range_r(f), 0u, 255u, false);
        }
    }
    __syncthreads();
    // coalesced store of the workgroup's rows
    unsigned char* out = A.out + (size_t)(A.first_image + img_i) * size * size;
    const int row0 = blockIdx.x * BROWS;
    for (int r = 0; r < BROWS && row0 + r < size; ++r)
        for (int x = threadIdx.x; x < size; x += BROWS * BSEG) out[(size_t)(row0 + r) * size + x] = s_px[(size_t)r * ldp + x];
}

}  // namespace

extern "C" {

int vpk_sphere_raster(vpk_handle* h, const double* l, const int64_t* offsets, int batch, int size, double alpha,
                      uint8_t* out) {
    if (!h || !offsets || !out || batch < 1 || size < 8 || size > 1024 || !(alpha >= 0.0 && alpha <= 1.0) ||
        (!l && offsets[batch] != offsets[0]))             // (no lines at all: the frame-only canvas, l may be NULL)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_sphere_raster: bad argument (size must be 8..1024, alpha 0..1)");
    VPK_HIP(h, hipSetDevice(h->device));
    for (int b = 0; b < batch; ++b)
        if (offsets[b + 1] < offsets[b]) return vpk_fail(h, VPK_ERR_ARG, "vpk_sphere_raster: offsets not monotone");
    const int samples = 10000;                            // sphere_mapping.py:40
    // images are processed in chunks of at most ~48k lines (workspace per line: outline scratch + coverage pools)
    // (the `alternative` curve has a pole: a row can be crossed three times -- left branch, the jump, right branch -- and a
    // row range then spans half the canvas, so its lines get a canvas' worth of coverage pool each and smaller chunks)
    const bool alt = h->raster_alternative != 0;
    const long long max_lines = alt ? std::max<long long>(1, std::min<long long>(49152, (3ll << 30) / ((long long)size * (size + 2)))) : 49152;
    const size_t per_line = (size_t)MAXS * sizeof(V2) + (size_t)MAXV * sizeof(V2) + POLY_INTS * 4 +
                            (size_t)MAXSUB * size * sizeof(RowEnt) + 8;
    long long chunk_lines = 0;
    for (int b0 = 0, b1; b0 < batch; b0 = b1) {           // the largest chunk decides the workspace
        b1 = b0 + 1;
        while (b1 < batch && offsets[b1 + 1] - offsets[b0] <= max_lines) ++b1;
        chunk_lines = std::max<long long>(chunk_lines, offsets[b1] - offsets[b0]);
    }
    const size_t ob = vpk::em_align((size_t)(batch + 1) * 8 + (size_t)batch * 4, 256);   // offsets, then the blend's image order
    const size_t fb = vpk::em_align(256 + (size_t)batch * 4, 256);
    const size_t tb = vpk::em_align((size_t)samples * 4 * 8, 256);
    const size_t nl = (size_t)chunk_lines + 4;
    // the coverage pool: 16 KB per line on average (measured: ~6 KB) and never less than eight canvases' worth -- the row
    // ranges of ONE line can span most of the canvas (see polygon_coverage), and a call may consist of one line
    const size_t alpha_bytes = std::max<size_t>(nl * (alt ? (size_t)size * (size + 2) : (size_t)16384), (size_t)8 * size * (size + 2));
    if (alpha_bytes >= (1ull << 32)) {                    // RowEnt::off is 32 bits (a chunk of 49 152 lines needs 0.8 GB)
        // one image is one chunk at least; the pool per line is 16 KB, or a whole canvas in `alternative` mode
        char msg[160];
        snprintf(msg, sizeof msg, "vpk_sphere_raster: one image has more than %lld lines (the 4 GiB coverage pool at %s per line)",
                 (long long)((1ull << 32) / (alt ? (size_t)size * (size + 2) : (size_t)16384)) - 4,
                 alt ? "one canvas (alternative mode)" : "16 KB");
        return vpk_fail(h, VPK_ERR_LIMIT, msg);
    }
    const size_t need = ob + fb + tb + nl * per_line + alpha_bytes + 8192;   // (the slack also covers the blend's 8-byte reads at the pool's end)
    const size_t had_bytes = h->raster_hdr_bytes;
    int rc = vpk_reserve(h, &h->raster_hdr, &h->raster_hdr_bytes, need, "hipMalloc(raster workspace)");
    if (rc) return rc;
    if (h->raster_hdr_bytes != had_bytes) {               // a fresh allocation (possibly at the old address): nothing cached in it
        h->raster_offsets.clear();
        h->raster_table_size = 0;
    }
    h->raster_last_batch = batch;
    char* base = (char*)h->raster_hdr;
    // offsets [host] -> device.  Caller-owned pageable memory, so the copy is waited for -- but a pipeline rasterises the
    // same batch structure again and again: when the device copy already holds these offsets nothing is uploaded and the
    // call does not wait for anything (the sample table is kept the same way).
    const bool same = h->raster_offsets.size() == (size_t)batch + 1 &&
                      memcmp(h->raster_offsets.data(), offsets, (size_t)(batch + 1) * 8) == 0;
    if (!same) {
        VPK_HIP(h, hipStreamSynchronize(h->stream));
        VPK_HIP(h, hipMemcpyAsync(base, offsets, (size_t)(batch + 1) * 8, hipMemcpyHostToDevice, h->stream));
        // the blend's workgroups of an image last as long as the image has lines: within a chunk the images are dealt out
        // longest first, so that the kernel does not end on a 400-line image that started last
        std::vector<int> order((size_t)batch);
        for (int b0 = 0, b1; b0 < batch; b0 = b1) {
            b1 = b0 + 1;
            while (b1 < batch && offsets[b1 + 1] - offsets[b0] <= max_lines) ++b1;
            for (int b = b0; b < b1; ++b) order[b] = b - b0;
            std::stable_sort(order.begin() + b0, order.begin() + b1, [&](int x, int y) {
                return offsets[b0 + x + 1] - offsets[b0 + x] > offsets[b0 + y + 1] - offsets[b0 + y]; });
        }
        VPK_HIP(h, hipMemcpyAsync(base + (size_t)(batch + 1) * 8, order.data(), (size_t)batch * 4, hipMemcpyHostToDevice, h->stream));
        VPK_HIP(h, hipStreamSynchronize(h->stream));
        h->raster_offsets.assign((const long long*)offsets, (const long long*)offsets + batch + 1);
    }
    VPK_HIP(h, hipMemsetAsync(base + ob, 0, fb, h->stream));
    if (!same || h->raster_table_size != size) {
        hipLaunchKernelGGL(raster_table_kernel, dim3((samples + 255) / 256), dim3(256), 0, h->stream, samples, size, (double*)(base + ob + fb));
        h->raster_table_size = size;
    }
    RasterArgs A;
    A.l = l; A.offsets = (const long long*)base; A.order = (const int*)(base + (size_t)(batch + 1) * 8); A.tab = (const double*)(base + ob + fb); A.size = size; A.samples = samples;
    A.a8 = (unsigned)(alpha * 255.0 + 0.5);               // agg::rgba8(rgba): uround
    A.out = out; A.ctr = (int*)(base + ob); A.bump = (unsigned long long*)(base + ob + 64); A.flags = (unsigned*)(base + ob + 256);
    char* p = base + ob + fb + tb;
    A.seq = (int*)p; p += vpk::em_align(nl * 4, 256);
    A.nsimp = (int*)p; p += vpk::em_align(nl * 4, 256);
    A.force_seq = getenv("VPK_RASTER_SEQUENTIAL") != nullptr;   // development / tests: every line through the sequential machine
    A.alt = h->raster_alternative;
    A.simp = (V2*)p; p += nl * MAXS * sizeof(V2);
    A.verts = (V2*)p; p += nl * MAXV * sizeof(V2);
    A.polys = (int*)p; p += vpk::em_align(nl * POLY_INTS * 4, 256);
    A.alpha = (unsigned char*)p; A.alpha_cap = alpha_bytes; p += vpk::em_align(alpha_bytes, 256);
    A.dense = (RowEnt*)p;
    // coverage_kernel's LDS: three workgroups per CU (53 KB each); what the static arrays and the six row arrays leave is the
    // pool.  (Measured with the two-range rows, 102 / 95 images: 512 threads x 2 per CU 1.36 / 2.60 ms, 256 x 4 1.32 / 2.23,
    // 256 x 3 1.06 / 1.98.)
    constexpr int COV_LDS = 53 * 1024, COV_STATIC = 11 * 1024;
    A.rowcap = (size + 63) & ~63;
    A.pool = ((COV_LDS - COV_STATIC - 6 * A.rowcap * 4) / 8) & ~63;
    const size_t cov_dyn = (size_t)(6 * A.rowcap + 2 * A.pool) * 4;
    if (!h->raster_ready) {
        VPK_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(blend_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       BROWS * 1024 + 256 + (BLEND_LUT_MAX_A8 + 1) * 256));
        VPK_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(coverage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       COV_LDS - COV_STATIC));
        h->raster_ready = true;
    }
    const int ldp = (size + 3) & ~3;
    for (int b0 = 0, b1; b0 < batch; b0 = b1) {
        b1 = b0 + 1;
        while (b1 < batch && offsets[b1 + 1] - offsets[b0] <= max_lines) ++b1;
        A.first_image = b0; A.batch = b1 - b0; A.line0 = offsets[b0]; A.nlines = offsets[b1] - offsets[b0];
        VPK_HIP(h, hipMemsetAsync(base + ob, 0, 256, h->stream));          // queue + bump counters of this chunk
        const long long nt = A.nlines + 4;
        const bool times = getenv("VPK_RASTER_TIMES") != nullptr;      // development: per-kernel device time
        A.probe = times ? 1 : 0;
        hipEvent_t ev[5] = {};
        if (times) for (int q = 0; q < 5; ++q) VPK_HIP(h, hipEventCreate(&ev[q]));
        if (times) VPK_HIP(h, hipEventRecord(ev[0], h->stream));
        if (A.nlines > 0)                                 // (a chunk of empty images: only the frame is drawn)
            hipLaunchKernelGGL(simplify_kernel, dim3((unsigned)A.nlines), dim3(64), 0, h->stream, A);
        if (times) VPK_HIP(h, hipEventRecord(ev[4], h->stream));
        hipLaunchKernelGGL(outline_kernel, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, h->stream, A);
        if (times) VPK_HIP(h, hipEventRecord(ev[1], h->stream));
        const int wgs = (int)std::min<long long>(nt, 3ll * h->num_cu);      // three workgroups fit a CU (LDS, registers)
        hipLaunchKernelGGL(coverage_kernel, dim3(wgs), dim3(RT), cov_dyn, h->stream, A);
        if (times) VPK_HIP(h, hipEventRecord(ev[2], h->stream));
        hipLaunchKernelGGL(blend_kernel, dim3((size + BROWS - 1) / BROWS, A.batch), dim3(BROWS * BSEG),
                           (size_t)BROWS * ldp + 256 + (A.a8 <= BLEND_LUT_MAX_A8 ? (size_t)(A.a8 + 1) * 256 : 0), h->stream, A);
        if (times) VPK_HIP(h, hipEventRecord(ev[3], h->stream));
        VPK_HIP(h, hipGetLastError());
        if (times) {
            float ms[3], ms_points = 0.f;
            VPK_HIP(h, hipEventSynchronize(ev[3]));
            for (int q = 0; q < 3; ++q) VPK_HIP(h, hipEventElapsedTime(&ms[q], ev[q], ev[q + 1]));
            VPK_HIP(h, hipEventElapsedTime(&ms_points, ev[0], ev[4]));
            unsigned long long bump[1];
            VPK_HIP(h, hipMemcpy(bump, A.bump, 8, hipMemcpyDeviceToHost));
            int dbg[16];
            VPK_HIP(h, hipMemcpy(dbg, A.ctr, sizeof(dbg), hipMemcpyDeviceToHost));
            fprintf(stderr, "vpk_sphere_raster: %d images, %lld lines: simplify %.2f + outlines %.2f ms, coverage %.2f ms, blend %.2f ms; %.0f coverage bytes "
                    "per line; workgroup 0 of the coverage kernel: bounds %.2f, scan %.2f, cells %.2f, sweep %.2f ms\n", A.batch,
                    A.nlines, ms_points, ms[0] - ms_points, ms[1], ms[2], (double)bump[0] / nt, dbg[8] * 1e-5, dbg[9] * 1e-5, dbg[10] * 1e-5,
                    dbg[11] * 1e-5);
            for (int q = 0; q < 5; ++q) (void)hipEventDestroy(ev[q]);
        }
    }
    return VPK_OK;
}

/* per-image flags of the last vpk_sphere_raster call on this handle (bit 0: a line's outline or coverage exceeded the
 * kernel's buffers and was truncated / dropped); waits for the call to finish */
int vpk_sphere_raster_flags(vpk_handle* h, int batch, uint32_t* flags_out) {
    if (!h || !flags_out || batch < 1 || !h->raster_hdr) return vpk_fail(h, VPK_ERR_ARG, "vpk_sphere_raster_flags: bad argument");
    if (batch != h->raster_last_batch)                    // the flags' place in the workspace depends on the call's batch
        return vpk_fail(h, VPK_ERR_ARG, "vpk_sphere_raster_flags: batch differs from the last vpk_sphere_raster call's");
    VPK_HIP(h, hipSetDevice(h->device));
    VPK_HIP(h, hipStreamSynchronize(h->stream));
    const size_t ob = vpk::em_align((size_t)(batch + 1) * 8 + (size_t)batch * 4, 256);   // (the layout of vpk_sphere_raster)
    VPK_HIP(h, hipMemcpy(flags_out, (char*)h->raster_hdr + ob + 256, (size_t)batch * 4, hipMemcpyDeviceToHost));
    return VPK_OK;
}

int vpk_sphere_raster_set_alternative(vpk_handle* h, int on) {
    if (!h) return VPK_ERR_ARG;
    if (h->raster_alternative != (on ? 1 : 0)) h->raster_offsets.clear();   // (the chunking, hence the cached image order, differs)
    h->raster_alternative = on ? 1 : 0;
    return VPK_OK;
}

}  // extern "C"
