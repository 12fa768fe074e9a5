"""CPU restatement of the reference's sphere rasteriser (TEST INFRASTRUCTURE: only tests/, smoke() and the scripts that
make goldens may import this; the product path is csrc/vpk_raster.hip).

The reference (sphere_mapping.py:36-72) draws, per line l = (a, b, c), the curve beta(alpha) = atan((-a sin alpha -
c cos alpha) / b) at 10 000 alpha in [-pi/2, pi/2] with matplotlib: ``ax.plot(a, b, '-', c=[1, 1, 1, 0.1])`` on a black
500 x 500 axes that fills the figure, then reads the Agg canvas back and averages R, G, B.  Its arithmetic therefore
lives in third-party code -- matplotlib's Agg backend (pinned 1.5.1 in the reference's requirements.txt; 3.10.8 in
this image, with ``lines.linewidth`` set to the old default 1.0, which is what tests/golden/*.npz were rendered with) and
the Anti-Grain Geometry library 2.4 it embeds.  Both are public; this module restates the stages a solid, anti-aliased,
unsnapped 1-pt Line2D goes through, in their order:

  Line2D.draw -> RendererAgg::draw_path (matplotlib src/_backend_agg.h):
    data -> pixels  (affine: x = (alpha + pi/2) / pi * W, y = H - (beta + pi/2) / pi * H: Agg's y axis points down)
    PathNanRemover, PathClipper (no-ops for this curve: it lies inside the canvas), PathSnapper (off: > 1024 vertices)
    PathSimplifier  (src/path_converters.h: merges runs of segments whose points stay within 1/9 px of the run's
                     first segment's line; emits the farthest forward / backward points of a run)
    agg::conv_stroke (agg_vcgen_stroke.cpp, agg_math_stroke.h): width 100/72 px, projecting (square) caps, round joins,
                     inner joins mitred, approximation scale 1
    agg::rasterizer_scanline_aa<rasterizer_sl_clip_dbl> (agg_rasterizer_cells_aa.h): 24.8 fixed point cells, non-zero
                     winding, clip box = the canvas
    agg::renderer_scanline_aa_solid over pixfmt_rgba32_plain with matplotlib's fixed_blender_rgba_plain: colour
                     (255, 255, 255, 26) -- uround(0.1 * 255) -- blended with alpha = round(26 * cover / 255) per pixel,
                     one line after the other

PINNED against matplotlib itself in the build container (oracle/check_agg_raster.py: bit-identical on single lines and on
whole line sets) and against the rasters stored in tests/golden/*.npz (tests/test_agg_raster.py).
"""
import math

import numpy as np

SUB = 256              # poly_subpixel_scale
SHIFT = 8


def iround(v):
    """agg::iround: int((v < 0.0) ? v - 0.5 : v + 0.5) (truncation toward zero)."""
    return int(v - 0.5) if v < 0.0 else int(v + 0.5)


# -----------------------------------------------------------------------------------------------------------------
# the curve in pixel coordinates
# -----------------------------------------------------------------------------------------------------------------
def curve_pixels(line, size=500, samples=10000, alternative=False):
    """sphere_mapping.py:40,58-63 + the axes' data -> display transform + RendererAgg's y flip."""
    a = np.linspace(-np.pi / 2, np.pi / 2, num=samples)
    with np.errstate(divide="ignore", invalid="ignore"):
        if alternative:
            b = -np.arctan(-line[2] / (np.cos(a) * line[0] + np.sin(a) * line[1]))    # :59
        else:
            b = -np.arctan((-line[0] * np.sin(a) - line[2] * np.cos(a)) / line[1])    # :61
    b = b * -1
    lo, hi = -np.pi / 2, np.pi / 2
    # matplotlib composes transScale + transLimits + transAxes into one affine; the products below agree with it to a few
    # ulp of a pixel coordinate, far below the 1/256 px grid the rasteriser rounds to
    x = (a - lo) / (hi - lo) * size
    y = size - (b - lo) / (hi - lo) * size
    return x, y


# -----------------------------------------------------------------------------------------------------------------
# PathSimplifier (matplotlib src/path_converters.h), for one polyline without NaNs
# -----------------------------------------------------------------------------------------------------------------
def simplify(xs, ys, threshold=1.0 / 9.0):
    """Returns the simplified polyline as a list of (x, y) (the first is the move_to)."""
    thr2 = threshold * threshold
    out = []
    n = len(xs)
    if n == 0:
        return out
    lastx, lasty = float(xs[0]), float(ys[0])
    orig_norm2 = 0.0
    origdx = origdy = 0.0
    fwd_max = bwd_max = 0.0
    last_fwd = last_bwd = False
    nextx = nexty = nbx = nby = 0.0
    startx = starty = 0.0
    clipped = True            # set by the initial move_to

    for i in range(1, n):
        x, y = float(xs[i]), float(ys[i])
        if orig_norm2 == 0.0:
            if clipped:
                out.append((lastx, lasty))
                clipped = False
            origdx = x - lastx
            origdy = y - lasty
            orig_norm2 = origdx * origdx + origdy * origdy
            fwd_max = orig_norm2
            bwd_max = 0.0
            last_fwd, last_bwd = True, False
            startx, starty = lastx, lasty
            nextx = lastx = x
            nexty = lasty = y
            continue
        totdx = x - startx
        totdy = y - starty
        totdot = origdx * totdx + origdy * totdy
        paradx = totdot * origdx / orig_norm2
        parady = totdot * origdy / orig_norm2
        perpdx = totdx - paradx
        perpdy = totdy - parady
        perp2 = perpdx * perpdx + perpdy * perpdy
        if perp2 < thr2:
            para2 = paradx * paradx + parady * parady
            last_fwd = last_bwd = False
            if totdot > 0.0:
                if para2 > fwd_max:
                    last_fwd = True
                    fwd_max = para2
                    nextx, nexty = x, y
            else:
                if para2 > bwd_max:
                    last_bwd = True
                    bwd_max = para2
                    nbx, nby = x, y
            lastx, lasty = x, y
            continue
        # _push
        if bwd_max > 0.0:
            if last_fwd:
                out.append((nbx, nby))
                out.append((nextx, nexty))
            else:
                out.append((nextx, nexty))
                out.append((nbx, nby))
        else:
            out.append((nextx, nexty))
        if clipped:
            out.append((lastx, lasty))        # (a move_to in the original; cannot happen here)
        elif (not last_fwd) and (not last_bwd):
            out.append((lastx, lasty))
        origdx = x - lastx
        origdy = y - lasty
        orig_norm2 = origdx * origdx + origdy * origdy
        fwd_max = orig_norm2
        last_fwd = True
        startx, starty = out[-1]
        lastx = nextx = x
        lasty = nexty = y
        bwd_max = 0.0
        last_bwd = False
        clipped = False
    # end of path
    if orig_norm2 != 0.0:
        out.append((nextx, nexty))
        if bwd_max > 0.0:
            out.append((nbx, nby))
    out.append((lastx, lasty))
    return out


# -----------------------------------------------------------------------------------------------------------------
# agg::conv_stroke: square caps, round joins, inner miter joins (agg_math_stroke.h, agg_vcgen_stroke.cpp)
# -----------------------------------------------------------------------------------------------------------------
VERTEX_DIST_EPSILON = 1e-14
INTERSECTION_EPSILON = 1.0e-30


def _calc_intersection(ax, ay, bx, by, cx, cy, dx, dy):
    num = (ay - cy) * (dx - cx) - (ax - cx) * (dy - cy)
    den = (bx - ax) * (dy - cy) - (by - ay) * (dx - cx)
    if abs(den) < INTERSECTION_EPSILON:
        return None
    r = num / den
    return ax + r * (bx - ax), ay + r * (by - ay)


def _cross(x1, y1, x2, y2, x, y):
    return (x - x2) * (y2 - y1) - (y - y2) * (x2 - x1)


class Stroker(object):
    def __init__(self, width):
        self.w = width * 0.5
        self.w_abs = abs(self.w)
        self.w_eps = self.w / 1024.0
        self.approx = 1.0
        self.inner_miter_limit = 1.01

    def cap(self, v0, v1, length):
        dx1 = (v1[1] - v0[1]) / length
        dy1 = (v1[0] - v0[0]) / length
        dx1 *= self.w
        dy1 *= self.w
        dx2 = dy1            # square cap (width_sign = 1)
        dy2 = dx1
        return [(v0[0] - dx1 - dx2, v0[1] + dy1 - dy2), (v0[0] + dx1 - dx2, v0[1] - dy1 - dy2)]

    def _miter(self, v0, v1, v2, dx1, dy1, dx2, dy2, mlimit):
        out = []
        lim = self.w_abs * mlimit
        exceeded = True
        p = _calc_intersection(v0[0] + dx1, v0[1] - dy1, v1[0] + dx1, v1[1] - dy1,
                               v1[0] + dx2, v1[1] - dy2, v2[0] + dx2, v2[1] - dy2)
        if p is not None:
            di = math.sqrt((p[0] - v1[0]) * (p[0] - v1[0]) + (p[1] - v1[1]) * (p[1] - v1[1]))
            if di <= lim:
                out.append(p)
                exceeded = False
        else:
            x2, y2 = v1[0] + dx1, v1[1] - dy1
            if (_cross(v0[0], v0[1], v1[0], v1[1], x2, y2) < 0.0) == (_cross(v1[0], v1[1], v2[0], v2[1], x2, y2) < 0.0):
                out.append((v1[0] + dx1, v1[1] - dy1))
                exceeded = False
        if exceeded:         # miter_join_revert
            out.append((v1[0] + dx1, v1[1] - dy1))
            out.append((v1[0] + dx2, v1[1] - dy2))
        return out

    def _arc(self, x, y, dx1, dy1, dx2, dy2):
        a1 = math.atan2(dy1, dx1)
        a2 = math.atan2(dy2, dx2)
        da = math.acos(self.w_abs / (self.w_abs + 0.125 / self.approx)) * 2
        out = [(x + dx1, y + dy1)]
        if a1 > a2:
            a2 += 2 * math.pi
        n = int((a2 - a1) / da)
        da = (a2 - a1) / (n + 1)
        a1 += da
        for _ in range(n):
            out.append((x + math.cos(a1) * self.w, y + math.sin(a1) * self.w))
            a1 += da
        out.append((x + dx2, y + dy2))
        return out

    def join(self, v0, v1, v2, len1, len2):
        w = self.w
        dx1 = w * (v1[1] - v0[1]) / len1
        dy1 = w * (v1[0] - v0[0]) / len1
        dx2 = w * (v2[1] - v1[1]) / len2
        dy2 = w * (v2[0] - v1[0]) / len2
        cp = _cross(v0[0], v0[1], v1[0], v1[1], v2[0], v2[1])
        if cp != 0 and (cp > 0) == (w > 0):
            limit = (len1 if len1 < len2 else len2) / self.w_abs
            if limit < self.inner_miter_limit:
                limit = self.inner_miter_limit
            return self._miter(v0, v1, v2, dx1, dy1, dx2, dy2, limit)        # inner_miter
        dx = (dx1 + dx2) / 2
        dy = (dy1 + dy2) / 2
        dbevel = math.sqrt(dx * dx + dy * dy)
        if self.approx * (self.w_abs - dbevel) < self.w_eps:
            p = _calc_intersection(v0[0] + dx1, v0[1] - dy1, v1[0] + dx1, v1[1] - dy1,
                                   v1[0] + dx2, v1[1] - dy2, v2[0] + dx2, v2[1] - dy2)
            if p is not None:
                return [p]
            return [(v1[0] + dx1, v1[1] - dy1)]
        return self._arc(v1[0], v1[1], dx1, -dy1, dx2, -dy2)                 # round join


def stroke_outline(points, width):
    """agg::vcgen_stroke on an open polyline: the closed outline polygon as a list of (x, y)."""
    # vertex_sequence<vertex_dist>: a vertex is dropped when it coincides with its predecessor
    seq = []          # [x, y, dist to next]
    for (x, y) in points:
        if len(seq) > 1:
            d = math.sqrt((seq[-1][0] - seq[-2][0]) ** 2 + (seq[-1][1] - seq[-2][1]) ** 2)
            seq[-2][2] = d
            if not (d > VERTEX_DIST_EPSILON):
                seq.pop()
        seq.append([x, y, 0.0])
    # close(false): remove trailing coincident vertices
    while len(seq) > 1:
        d = math.sqrt((seq[-1][0] - seq[-2][0]) ** 2 + (seq[-1][1] - seq[-2][1]) ** 2)
        seq[-2][2] = d
        if d > VERTEX_DIST_EPSILON:
            break
        seq.pop()
    n = len(seq)
    if n < 2:
        return []
    st = Stroker(width)
    out = []
    out += st.cap(seq[0], seq[1], seq[0][2])
    for i in range(1, n - 1):
        out += st.join(seq[i - 1], seq[i], seq[i + 1], seq[i - 1][2], seq[i][2])
    out += st.cap(seq[n - 1], seq[n - 2], seq[n - 2][2])
    for i in range(n - 2, 0, -1):
        out += st.join(seq[i + 1], seq[i], seq[i - 1], seq[i][2], seq[i - 1][2])
    return out


# -----------------------------------------------------------------------------------------------------------------
# agg::rasterizer_scanline_aa<rasterizer_sl_clip_dbl>: clipping + cells
# -----------------------------------------------------------------------------------------------------------------
class Cells(object):
    """rasterizer_cells_aa: accumulates (cover, area) per pixel.  Sums are order independent, so a dense pair of int
    arrays replaces the sorted cell list."""

    def __init__(self, width, height):
        self.w, self.h = width, height
        self.cover = np.zeros((height, width + 2), dtype=np.int64)       # x index shifted by 1: cells at x = -1 exist
        self.area = np.zeros((height, width + 2), dtype=np.int64)
        self.touched = set()

    def add(self, ex, ey, cover, area):
        if cover == 0 and area == 0:
            return
        if 0 <= ey < self.h and -1 <= ex <= self.w:
            self.cover[ey, ex + 1] += cover
            self.area[ey, ex + 1] += area
            self.touched.add(ey)

    def hline(self, ey, x1, y1, x2, y2):
        ex1, ex2 = x1 >> SHIFT, x2 >> SHIFT
        fx1, fx2 = x1 & (SUB - 1), x2 & (SUB - 1)
        if y1 == y2:
            return
        if ex1 == ex2:
            delta = y2 - y1
            self.add(ex1, ey, delta, (fx1 + fx2) * delta)
            return
        p = (SUB - fx1) * (y2 - y1)
        first = SUB
        incr = 1
        dx = x2 - x1
        if dx < 0:
            p = fx1 * (y2 - y1)
            first = 0
            incr = -1
            dx = -dx
        delta = int(math.trunc(p / dx)) if False else _cdiv(p, dx)
        mod = _cmod(p, dx)
        if mod < 0:
            delta -= 1
            mod += dx
        self.add(ex1, ey, delta, (fx1 + first) * delta)
        ex1 += incr
        y1 += delta
        if ex1 != ex2:
            p = SUB * (y2 - y1 + delta)
            lift = _cdiv(p, dx)
            rem = _cmod(p, dx)
            if rem < 0:
                lift -= 1
                rem += dx
            mod -= dx
            while ex1 != ex2:
                delta = lift
                mod += rem
                if mod >= 0:
                    mod -= dx
                    delta += 1
                self.add(ex1, ey, delta, SUB * delta)
                y1 += delta
                ex1 += incr
        delta = y2 - y1
        self.add(ex1, ey, delta, (fx2 + SUB - first) * delta)

    def line(self, x1, y1, x2, y2):
        dx = x2 - x1
        # (the dx_limit split of very long lines never triggers on a 500 px canvas)
        dy = y2 - y1
        ex1, ex2 = x1 >> SHIFT, x2 >> SHIFT
        ey1, ey2 = y1 >> SHIFT, y2 >> SHIFT
        fy1, fy2 = y1 & (SUB - 1), y2 & (SUB - 1)
        if ey1 == ey2:
            self.hline(ey1, x1, fy1, x2, fy2)
            return
        incr = 1
        if dx == 0:
            ex = x1 >> SHIFT
            two_fx = (x1 - (ex << SHIFT)) << 1
            first = SUB
            if dy < 0:
                first = 0
                incr = -1
            delta = first - fy1
            self.add(ex, ey1, delta, two_fx * delta)
            ey1 += incr
            delta = first + first - SUB
            area = two_fx * delta
            while ey1 != ey2:
                self.add(ex, ey1, delta, area)
                ey1 += incr
            delta = fy2 - SUB + first
            self.add(ex, ey1, delta, two_fx * delta)
            return
        p = (SUB - fy1) * dx
        first = SUB
        if dy < 0:
            p = fy1 * dx
            first = 0
            incr = -1
            dy = -dy
        delta = _cdiv(p, dy)
        mod = _cmod(p, dy)
        if mod < 0:
            delta -= 1
            mod += dy
        x_from = x1 + delta
        self.hline(ey1, x1, fy1, x_from, first)
        ey1 += incr
        if ey1 != ey2:
            p = SUB * dx
            lift = _cdiv(p, dy)
            rem = _cmod(p, dy)
            if rem < 0:
                lift -= 1
                rem += dy
            mod -= dy
            while ey1 != ey2:
                delta = lift
                mod += rem
                if mod >= 0:
                    mod -= dy
                    delta += 1
                x_to = x_from + delta
                self.hline(ey1, x_from, SUB - first, x_to, first)
                x_from = x_to
                ey1 += incr
        self.hline(ey1, x_from, SUB - first, x2, fy2)


def _cdiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _cmod(a, b):
    return a - _cdiv(a, b) * b


def _flags(x, y, box):
    return (x > box[2]) | ((y > box[3]) << 1) | ((x < box[0]) << 2) | ((y < box[1]) << 3)


def _flags_y(y, box):
    return ((y > box[3]) << 1) | ((y < box[1]) << 3)


class Clipper(object):
    """rasterizer_sl_clip<ras_conv_dbl> with a clip box."""

    def __init__(self, cells, box):
        self.c = cells
        self.box = box
        self.x1 = self.y1 = 0.0
        self.f1 = 0

    def move_to(self, x, y):
        self.x1, self.y1 = x, y
        self.f1 = _flags(x, y, self.box)

    def _clip_y(self, x1, y1, x2, y2, f1, f2):
        f1 &= 10
        f2 &= 10
        box = self.box
        if (f1 | f2) == 0:
            self.c.line(iround(x1 * SUB), iround(y1 * SUB), iround(x2 * SUB), iround(y2 * SUB))
            return
        if f1 == f2:
            return
        tx1, ty1, tx2, ty2 = x1, y1, x2, y2
        if f1 & 8:
            tx1 = x1 + (box[1] - y1) * (x2 - x1) / (y2 - y1)
            ty1 = box[1]
        if f1 & 2:
            tx1 = x1 + (box[3] - y1) * (x2 - x1) / (y2 - y1)
            ty1 = box[3]
        if f2 & 8:
            tx2 = x1 + (box[1] - y1) * (x2 - x1) / (y2 - y1)
            ty2 = box[1]
        if f2 & 2:
            tx2 = x1 + (box[3] - y1) * (x2 - x1) / (y2 - y1)
            ty2 = box[3]
        self.c.line(iround(tx1 * SUB), iround(ty1 * SUB), iround(tx2 * SUB), iround(ty2 * SUB))

    def line_to(self, x2, y2):
        box = self.box
        f2 = _flags(x2, y2, box)
        if (self.f1 & 10) == (f2 & 10) and (self.f1 & 10) != 0:
            self.x1, self.y1, self.f1 = x2, y2, f2
            return
        x1, y1, f1 = self.x1, self.y1, self.f1
        case = ((f1 & 5) << 1) | (f2 & 5)
        if case == 0:
            self._clip_y(x1, y1, x2, y2, f1, f2)
        elif case == 1:
            y3 = y1 + (box[2] - x1) * (y2 - y1) / (x2 - x1)
            f3 = _flags_y(y3, box)
            self._clip_y(x1, y1, box[2], y3, f1, f3)
            self._clip_y(box[2], y3, box[2], y2, f3, f2)
        elif case == 2:
            y3 = y1 + (box[2] - x1) * (y2 - y1) / (x2 - x1)
            f3 = _flags_y(y3, box)
            self._clip_y(box[2], y1, box[2], y3, f1, f3)
            self._clip_y(box[2], y3, x2, y2, f3, f2)
        elif case == 3:
            self._clip_y(box[2], y1, box[2], y2, f1, f2)
        elif case == 4:
            y3 = y1 + (box[0] - x1) * (y2 - y1) / (x2 - x1)
            f3 = _flags_y(y3, box)
            self._clip_y(x1, y1, box[0], y3, f1, f3)
            self._clip_y(box[0], y3, box[0], y2, f3, f2)
        elif case == 6:
            y3 = y1 + (box[2] - x1) * (y2 - y1) / (x2 - x1)
            y4 = y1 + (box[0] - x1) * (y2 - y1) / (x2 - x1)
            f3 = _flags_y(y3, box)
            f4 = _flags_y(y4, box)
            self._clip_y(box[2], y1, box[2], y3, f1, f3)
            self._clip_y(box[2], y3, box[0], y4, f3, f4)
            self._clip_y(box[0], y4, box[0], y2, f4, f2)
        elif case == 8:
            y3 = y1 + (box[0] - x1) * (y2 - y1) / (x2 - x1)
            f3 = _flags_y(y3, box)
            self._clip_y(box[0], y1, box[0], y3, f1, f3)
            self._clip_y(box[0], y3, x2, y2, f3, f2)
        elif case == 9:
            y3 = y1 + (box[0] - x1) * (y2 - y1) / (x2 - x1)
            y4 = y1 + (box[2] - x1) * (y2 - y1) / (x2 - x1)
            f3 = _flags_y(y3, box)
            f4 = _flags_y(y4, box)
            self._clip_y(box[0], y1, box[0], y3, f1, f3)
            self._clip_y(box[0], y3, box[2], y4, f3, f4)
            self._clip_y(box[2], y4, box[2], y2, f4, f2)
        elif case == 12:
            self._clip_y(box[0], y1, box[0], y2, f1, f2)
        self.f1 = f2
        self.x1, self.y1 = x2, y2


def polygon_coverage(poly, width, height):
    """Coverage (0..255) of one closed polygon under non-zero winding: {(y, x): cover} for the touched pixels, as
    sweep_scanline + scanline_u8 produce it."""
    cells = Cells(width, height)
    clip = Clipper(cells, (0.0, 0.0, float(width), float(height)))
    clip.move_to(poly[0][0], poly[0][1])
    for (x, y) in poly[1:]:
        clip.line_to(x, y)
    clip.line_to(poly[0][0], poly[0][1])          # close_polygon
    out = {}
    for y in sorted(cells.touched):
        cov_row, area_row = cells.cover[y], cells.area[y]
        nz = np.nonzero((cov_row != 0) | (area_row != 0))[0]
        cover = 0
        k = 0
        while k < len(nz):
            xi = int(nz[k])
            x = xi - 1
            cover += int(cov_row[xi])
            area = int(area_row[xi])
            if area:
                a = _alpha((cover << (SHIFT + 1)) - area)
                if a and 0 <= x < width:
                    out[(y, x)] = a
                x += 1
            k += 1
            nxt = int(nz[k]) - 1 if k < len(nz) else None
            if nxt is not None and nxt > x:
                a = _alpha(cover << (SHIFT + 1))
                if a:
                    for xx in range(max(x, 0), min(nxt, width)):
                        out[(y, xx)] = a
    return out


def _alpha(area):
    cover = area >> (SHIFT * 2 + 1 - 8)
    if cover < 0:
        cover = -cover
    if cover > 255:
        cover = 255
    return cover


# -----------------------------------------------------------------------------------------------------------------
# blending: fixed_blender_rgba_plain (matplotlib src/agg_workaround.h) of white with alpha 26 over the canvas
# -----------------------------------------------------------------------------------------------------------------
COLOR_A = 26            # agg::rgba8(rgba(1, 1, 1, 0.1)).a = uround(0.1 * 255)


def blend_white(p, cover, color_a=COLOR_A):
    """One channel of an opaque grey pixel p (R = G = B, A = 255) after blending white with this coverage."""
    if color_a == 255 and cover == 255:
        return 255
    t = color_a * cover + 128                  # rgba8::mult_cover = multiply(a, cover): a * cover / 255, rounded
    alpha = ((t >> 8) + t) >> 8
    if alpha == 0:
        return p
    a = 255
    r = p * a
    a = ((alpha + a) << 8) - alpha * a
    return (((255 << 8) - r) * alpha + (r << 8)) // a


def line_coverage(line, size=500, samples=10000, width_px=100.0 / 72.0, alternative=False):
    """{(y, x): cover} of one line's stroke.  A non-finite sample breaks the path (PathNanRemover: the next finite sample
    is a move_to); the sub-paths are stroked one by one and their covers blended one after the other by the caller --
    returned here merged, which is the same thing wherever the strokes do not overlap (they are separated by a gap)."""
    x, y = curve_pixels(line, size, samples, alternative)
    ok = np.isfinite(x) & np.isfinite(y)
    out = {}
    idx = np.nonzero(ok)[0]
    if len(idx) == 0:
        return out
    runs = np.split(idx, np.nonzero(np.diff(idx) > 1)[0] + 1)
    for run in runs:
        pts = simplify(x[run], y[run])
        poly = stroke_outline(pts, width_px)
        if len(poly) < 3:
            continue
        for k, c in polygon_coverage(poly, size, size).items():
            out[k] = c if k not in out else out[k]          # (overlapping sub-paths of one line: first wins; not expected)
    return out


def blend_black(p, cover):
    """The same blender with colour (0, 0, 0, 255): the axes' frame drawn over the lines."""
    if cover == 255:
        return 0                                   # opaque colour at full coverage: the pixel is copied
    t = 255 * cover + 128
    alpha = ((t >> 8) + t) >> 8
    if alpha == 0:
        return p
    r = p * 255
    a = ((alpha + 255) << 8) - alpha * 255
    return ((0 - r) * alpha + (r << 8)) // a


def spine_coverage(size=500, linewidth_pt=0.8, dpi=100.0):
    """The four axes spines (matplotlib draws them after the lines, zorder 2.5 > 2: black, axes.linewidth = 0.8 pt,
    projecting caps): two-vertex rectilinear paths, which PathSnapper moves to pixel centres (stroke width rounds to an
    odd number of pixels: snap value 0.5), stroked and rasterised like any other path.  Returns one coverage dict per
    spine in drawing order (left, right, bottom, top)."""
    w = linewidth_pt * dpi / 72.0

    def snap(v):
        return math.floor(v + 0.5) + 0.5

    s = float(size)
    paths = [[(0.0, s), (0.0, 0.0)],          # left:   axes (0, 0) -> (0, 1), y flipped
             [(s, s), (s, 0.0)],              # right
             [(0.0, s), (s, s)],              # bottom: axes (0, 0) -> (1, 0)
             [(0.0, 0.0), (s, 0.0)]]          # top
    out = []
    for pts in paths:
        pts = [(snap(x), snap(y)) for (x, y) in pts]
        out.append(polygon_coverage(stroke_outline(pts, w), size, size))
    return out


_SPINES = {}


def raster(lines, size=500, samples=10000, alpha=0.1, alternative=False):
    """sphere_line_plot(lines, size, alpha, alternative=...) -> uint8 (size, size)."""
    img = np.zeros((size, size), dtype=np.int64)
    color_a = int(alpha * 255 + 0.5)                   # agg::rgba8(rgba): uround
    for line in np.asarray(lines, dtype=np.float64).reshape(-1, 3):
        for (y, x), c in line_coverage(line, size, samples, alternative=alternative).items():
            img[y, x] = blend_white(int(img[y, x]), c, color_a)
    if size not in _SPINES:
        _SPINES[size] = spine_coverage(size)
    for cov in _SPINES[size]:
        for (y, x), c in cov.items():
            if img[y, x]:
                img[y, x] = blend_black(int(img[y, x]), c)
    return img.astype(np.uint8)
