// vpk_horizon.hip -- horizon line from the best orthogonal VP triplet, batched (calc_horizon.py:19-225).
//
// The reference scores all i<j<k triplets of the (up to 20) best-supported VPs of an image in Python
// (C(20,3) = 1140 triplets, ~20 ms per image): after the EM itself this is the slowest stage of its
// benchmark loop (benchmark.py:229-243).  Here one workgroup scores the triplets of one image, one triplet
// per thread and round, and picks the first maximum in itertools.combinations order (:45-50, :190-196).
// The order of the best VPs (np.argsort(counts)[::-1][:maxbest], :34-36) is an INPUT: its tie order is a
// property of the caller's NumPy sort, so the caller supplies it.
// Compiled with -ffp-contract=off; dot products and norms round like NumPy's (fused chains, see
// em_device.hpp), cross products like np.cross (two roundings per component).
#include "vpk_internal.hpp"

namespace {

__device__ __forceinline__ double dot3(double ax, double ay, double az, double bx, double by, double bz) {
    return fma(az, bz, fma(ay, by, ax * bx));
}
__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt(dot3(x, y, z, x, y, z)); }
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }

struct Triplet {
    double score;
    double h1[3], h2[3], zv[3], hl[3];
};

// calc_horizon.py:11-16
__device__ __forceinline__ bool vp_in_image(const double* v) {
    const double qx = v[0] / v[2], qy = v[1] / v[2];
    return qx <= 1 && qx >= -1 && qy <= 1 && qy >= -1;
}

// end points of a homogeneous line at x = +1 / x = -1: cross(hl, (1,0,1)) / z, cross(hl, (-1,0,1)) / z
__device__ __forceinline__ void end_points(const double* hl, double* p1, double* p2) {
    const double a0 = hl[0], a1 = hl[1], a2 = hl[2];
    const double c1x = a1 * 1.0 - a2 * 0.0, c1y = a2 * 1.0 - a0 * 1.0, c1z = a0 * 0.0 - a1 * 1.0;
    const double c2x = a1 * 1.0 - a2 * 0.0, c2y = a2 * -1.0 - a0 * 1.0, c2z = a0 * 0.0 - a1 * -1.0;
    p1[0] = c1x / c1z; p1[1] = c1y / c1z; p1[2] = c1z / c1z;
    p2[0] = c2x / c2z; p2[1] = c2y / c2z; p2[2] = c2z / c2z;
}

// one triplet (ia, ib, ic) of VP indices (calc_horizon.py:52-196)
__device__ void score_triplet(const double* vps, const double* counts, int ia, int ib, int ic, double sin_tz,
                              double costh, double max_tilt, Triplet& T) {
    const double* va = vps + 3 * ia;
    const double* vb = vps + 3 * ib;
    const double* vc = vps + 3 * ic;
    const double ab = fabs(dot3(va[0], va[1], va[2], vb[0], vb[1], vb[2]));
    const double bc = fabs(dot3(vb[0], vb[1], vb[2], vc[0], vc[1], vc[2]));
    const double ac = fabs(dot3(va[0], va[1], va[2], vc[0], vc[1], vc[2]));
    int num_zenith = 0;
    const double* zenith = va;
    if (fabs(va[1]) > sin_tz) { ++num_zenith; zenith = va; }              // :82-91, the last member wins
    if (fabs(vb[1]) > sin_tz) { ++num_zenith; zenith = vb; }
    if (fabs(vc[1]) > sin_tz) { ++num_zenith; zenith = vc; }
    const int num_central = (int)vp_in_image(va) + (int)vp_in_image(vb) + (int)vp_in_image(vc);
    const double ya = fabs(va[1]), yb = fabs(vb[1]), yc = fabs(vc[1]);
    const double *h1, *h2, *zv;
    double c1, c2;
    if (ya > yb && ya > yc) { h1 = vb; h2 = vc; zv = va; c1 = counts[ib]; c2 = counts[ic]; }          // :105-125
    else if (yb > ya && yb > yc) { h1 = va; h2 = vc; zv = vb; c1 = counts[ia]; c2 = counts[ic]; }
    else { h1 = va; h2 = vb; zv = vc; c1 = counts[ia]; c2 = counts[ib]; }
    // zlin = cross(zv, e_z) = (zv_y * 1 - zv_z * 0, zv_z * 0 - zv_x * 1, zv_x * 0 - zv_y * 0)
    double zl0 = zv[1] * 1.0 - zv[2] * 0.0, zl1 = zv[2] * 0.0 - zv[0] * 1.0;
    const double zn = norm2(zl0, zl1);
    const double l1 = zl0 / zn, l2 = zl1 / zn;
    const double q1x = h1[0] / h1[2], q1y = h1[1] / h1[2], q1z = h1[2] / h1[2];
    const double q2x = h2[0] / h2[2], q2y = h2[1] / h2[2], q2z = h2[2] / h2[2];
    const double d1 = norm3(0.0 - q1x, 0.0 - q1y, 1.0 - q1z);
    const double d2 = norm3(0.0 - q2x, 0.0 - q2y, 1.0 - q2z);
    const double h3 = ((h1[0] * l2 - h1[1] * l1) / h1[2] * (d2 * c1) + (h2[0] * l2 - h2[1] * l1) / h2[2] * (d1 * c2)) /
                      ((d1 * c2) + (d2 * c1));                                                          // :147
    T.hl[0] = -l2; T.hl[1] = l1; T.hl[2] = h3;
    const double hx = q1x - q2x, hy = q1y - q2y, hz = q1z - q2z;
    const double hn = norm3(hx, hy, hz);
    const double hang = acos(fabs(dot3(hx, hy, hz, 1.0, 0.0, 0.0)) / hn);
    double p1[3], p2[3];
    end_points(T.hl, p1, p2);
    double ortho = 0.0;
    if (num_zenith == 1) {                                                                            // :164-167
        const double zn3 = norm3(zenith[0], zenith[1], zenith[2]);
        const double cosphi = fabs(dot3(hx / hn, hy / hn, hz / hn, zenith[0] / zn3, zenith[1] / zn3, zenith[2] / zn3));
        const double cl = cosphi < 0.0 ? 0.0 : (cosphi > 1.0 ? 1.0 : cosphi);    // np.clip; NaN passes through
        ortho = 1.0 - (cosphi != cosphi ? cosphi : cl);
    }
    const int zenith_pos = zv[1] > 0 ? 1 : -1;
    const int hor_pos = (p1[1] + p2[1]) / 2 < 0 ? 1 : -1;
    const bool ok = ab < costh && bc < costh && ac < costh && num_zenith == 1 && num_central <= 1 && hang < max_tilt &&
                    zenith_pos * hor_pos == 1;                                                        // :176-179
    T.score = (ok ? 1.0 : 0.0) * ((counts[ia] + counts[ib]) + counts[ic]) * ortho;                    // :182-185
    for (int q = 0; q < 3; ++q) { T.h1[q] = h1[q]; T.h2[q] = h2[q]; T.zv[q] = zv[q]; }
}

constexpr int HZ_THREADS = 256;
constexpr int HZ_MAXBEST = 64;

__global__ __launch_bounds__(HZ_THREADS) void horizon_kernel(int max_vp, const double* __restrict__ vp,
                                                            const double* __restrict__ counts,
                                                            const int* __restrict__ num_vp, const int* __restrict__ order,
                                                            int maxbest, double sin_tz, double costh, double max_tilt,
                                                            double* __restrict__ out, int* __restrict__ combo_out) {
    __shared__ double s_vp[3 * 64], s_cnt[64];
    __shared__ int s_ord[HZ_MAXBEST];
    __shared__ double s_score[HZ_THREADS];
    __shared__ int s_idx[HZ_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    int M = num_vp[b];
    M = M < 0 ? 0 : (M > max_vp ? max_vp : M);
    const int nb = M < maxbest ? M : maxbest;
    for (int q = tid; q < 3 * M && q < 3 * 64; q += HZ_THREADS) s_vp[q] = vp[(size_t)b * max_vp * 3 + q];
    for (int q = tid; q < M && q < 64; q += HZ_THREADS) s_cnt[q] = counts[(size_t)b * max_vp + q];
    for (int q = tid; q < nb; q += HZ_THREADS) s_ord[q] = order[(size_t)b * maxbest + q];
    __syncthreads();
    double* o = out + (size_t)b * 15;              // hP1 | hP2 | zVP | hVP1 | hVP2
    int* oc = combo_out + (size_t)b * 3;
    if (nb > 2) {
        double best = -1.0;
        int best_idx = 0x7fffffff;
        int idx = 0;
        for (int a = 0; a < nb - 2; ++a)
            for (int bb = a + 1; bb < nb - 1; ++bb)
                for (int c = bb + 1; c < nb; ++c, ++idx) {
                    if (idx % HZ_THREADS != tid) continue;
                    Triplet T;
                    score_triplet(s_vp, s_cnt, s_ord[a], s_ord[bb], s_ord[c], sin_tz, costh, max_tilt, T);
                    if (T.score > best) { best = T.score; best_idx = idx; }          // first maximum, NaN never wins
                }
        s_score[tid] = best;
        s_idx[tid] = best_idx;
        __syncthreads();
        if (tid == 0) {
            double g = -1.0;
            int gi = 0;                                                               // :190 best_idx starts at 0
            bool any = false;
            for (int t = 0; t < HZ_THREADS; ++t) {
                const double sc = s_score[t];
                const int si = s_idx[t];
                if (si == 0x7fffffff) continue;
                if (!any || sc > g || (sc == g && si < gi)) { g = sc; gi = si; any = true; }
            }
            if (!any) gi = 0;
            // decode gi -> (a, b, c) and recompute the winner
            int a = 0, bb = 1, c = 2, k = 0;
            bool found = false;
            for (a = 0; a < nb - 2 && !found; ++a)
                for (bb = a + 1; bb < nb - 1 && !found; ++bb)
                    for (c = bb + 1; c < nb; ++c, ++k)
                        if (k == gi) { found = true; break; }
            --a; --bb;                                                                // undo the loop increments
            Triplet T;
            score_triplet(s_vp, s_cnt, s_ord[a], s_ord[bb], s_ord[c], sin_tz, costh, max_tilt, T);
            double p1[3], p2[3];
            end_points(T.hl, p1, p2);
            for (int q = 0; q < 3; ++q) { o[q] = p1[q]; o[3 + q] = p2[q]; o[6 + q] = T.zv[q]; o[9 + q] = T.h1[q]; o[12 + q] = T.h2[q]; }
            oc[0] = s_ord[a]; oc[1] = s_ord[bb]; oc[2] = s_ord[c];
        }
    } else if (tid == 0) {
        double hv1[3], hv2[3], zv[3] = {0.0, 1.0, 0.0}, hl[3];
        if (nb > 1) {                                                                 // :200-205
            for (int q = 0; q < 3; ++q) { hv1[q] = s_vp[q]; hv2[q] = s_vp[3 + q]; }
            hl[0] = hv1[1] * hv2[2] - hv1[2] * hv2[1];
            hl[1] = hv1[2] * hv2[0] - hv1[0] * hv2[2];
            hl[2] = hv1[0] * hv2[1] - hv1[1] * hv2[0];
            oc[0] = 0; oc[1] = 1; oc[2] = -1;
        } else {
            if (nb > 0) { for (int q = 0; q < 3; ++q) { hv1[q] = s_vp[q]; hv2[q] = s_vp[q]; } }       // :206-211
            else { hv1[0] = -1; hv1[1] = 0; hv1[2] = 0; hv2[0] = 1; hv2[1] = 0; hv2[2] = 0; }          // :212-217
            hl[0] = 0.0 * 1.0 - 1.0 * 0.0; hl[1] = 1.0 * 1.0 - 0.0 * 1.0; hl[2] = 0.0 * 0.0 - 0.0 * 1.0;   // cross(e_z, (1,0,1))
            oc[0] = 0; oc[1] = 0; oc[2] = -1;
        }
        double p1[3], p2[3];
        end_points(hl, p1, p2);
        for (int q = 0; q < 3; ++q) { o[q] = p1[q]; o[3 + q] = p2[q]; o[6 + q] = zv[q]; o[9 + q] = hv1[q]; o[12 + q] = hv2[q]; }
    }
}

}  // namespace

extern "C" {

int vpk_horizon_batch(vpk_handle* h, int batch, int max_vp, const double* vp, const double* counts,
                      const int32_t* num_vp, const int32_t* order, int maxbest, double theta_vmin, double theta_z,
                      double* out, int32_t* combo_out) {
    if (!h || batch < 1 || max_vp < 1 || max_vp > 64 || !vp || !counts || !num_vp || !order || !out || !combo_out ||
        maxbest < 1 || maxbest > HZ_MAXBEST)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_horizon_batch: bad argument (max_vp, maxbest <= 64)");
    VPK_HIP(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(horizon_kernel, dim3(batch), dim3(HZ_THREADS), 0, h->stream, max_vp, vp, counts, (const int*)num_vp,
                       (const int*)order, maxbest, sin(theta_z), cos(theta_vmin), 30.0 * 3.14159265358979323846 / 180.0,
                       out, (int*)combo_out);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

}  // extern "C"
