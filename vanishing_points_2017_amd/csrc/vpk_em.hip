// vpk_em.hip -- EM kernels and their C-ABI entry points (see include/vpk.h).
//
// Compiled with -ffp-contract=off (see em_device.hpp).  One persistent workgroup of EM_THREADS
// threads per slot; workgroups pull images from a device-side queue (largest N first), so a
// ragged batch keeps every CU busy without host round trips.
#include "em_device.hpp"
#include "vpk_internal.hpp"

#include <algorithm>
#include <string.h>
#include <numeric>
#include <vector>

using namespace vpk;

namespace {

constexpr int EM_THREADS = 512;   // 8 waves: 2 per SIMD
// One workgroup per CU: the (non-inlined) phase functions touch ~250 VGPRs each -- the calling
// convention's callee-saved registers are striped through the file, so the allocator spreads over all of
// it -- and two waves per SIMD fill the register file.  Declaring the kernels for 1024 threads makes
// hipcc give the phases 128 VGPRs and two workgroups fit, but measured (r1): every phase slows down by
// 15-40 % (spills, narrower smoother) and the batch gains nothing at YUD sizes, +5 % at the stress shape.
constexpr int EM_BOUND = EM_THREADS;
constexpr int EM_WAVES = EM_THREADS / 64;
// dynamic LDS: [Shared | smoother operand tile]; ~77 KiB -> two workgroups per CU (160 KiB)
constexpr size_t EM_LDS_BYTES = SH_BYTES + WT_DOUBLES * sizeof(double);
// when a batch leaves at most one workgroup per CU, the workgroup takes (almost) the whole 160 KiB so
// that the smoother's operand panel of any YUD/ECD-sized image fits and lsim is read once per E-step
constexpr int WT_DOUBLES_BIG = (int)((163840 - SH_BYTES) / sizeof(double)) / 32 * 32;   // everything a CU has (~147 KiB)
constexpr size_t EM_LDS_BYTES_BIG = SH_BYTES + WT_DOUBLES_BIG * sizeof(double);
static_assert(EM_LDS_BYTES_BIG <= 163840, "LDS per CU");
#define VPK_SHARED_DECL Shared& sh = SH()

struct EmBatchArgs {
    int B;
    const long long* offsets;   // device, B+1
    const int* order;           // device, B (image indices, largest first)
    int* queue;                 // device counter
    double* l;
    const double* lp;
    const float* cnn;
    const unsigned char* sphere;
    int ssize;
    const double* init_vp;
    int n_init;
    vpk_em_params prm;
    EmLayout L;
    double* scratch;
    int max_vp;
    double* vp_out;
    double* sigma_out;
    double* counts_out;
    double* counts_w_out;
    int* num_vp_out;
    long long* assoc_out;
    int* iterations_out;
    int* status_out;
    unsigned* flags_out;
    double* metric_out;
    double* trace_out;
    int wt_doubles;
    int smoother;               // vpk_em_set_smoother
    vpk_em_dist_out dist;       // all null = not requested
};

// ---- time-sliced launches (vpk_em_set_time_slice) ------------------------------------------------
// An image that is not finished when its launch's deadline passes is parked as an EmCarry entry: the next
// launch on the handle drains the entries of the previous one before it takes fresh images.  Entries of
// images that were never started (the deadline passed first) carry slot = -1.
struct EmCarry {
    EmCtx c;
    EmOut o;
    int iter;     // -1: not started; >= 0: resume at the top of this iteration (state in the slot)
    int slot;     // -1: none yet
};
// Two lists per launch: images that have been STARTED (they own a slot) and images that have not.  A launch
// drains the started list before it touches anything else, which bounds the slots in use: every started entry
// is taken by a workgroup right at the beginning of the launch, so before the deadline each workgroup holds
// exactly one image, a new image starts only on a free workgroup, and after the deadline nothing starts --
// at most one started image per workgroup is ever parked (S_k <= workgroups for every launch k, by induction),
// and a free workgroup always finds a free slot among 2 x workgroups + 8.
struct EmSliceArgs {
    int enabled;
    long long budget_ticks;   // 0 = no deadline (flush): everything runs to completion
    int* ctr;                 // started: [0] consumed, [1] count of `in`, [2] appended to `out`; unstarted: [3], [4], [5]
    EmCarry* in_started;
    EmCarry* out_started;
    EmCarry* in_waiting;
    EmCarry* out_waiting;
    int cap_started, cap_waiting;
    int* busy;                // slot flags
    int nslots;
};

// take a free slot (session mode: slots outlive the launch, so they are not tied to the workgroup).  The invariant
// below (EmSliceArgs) guarantees a free slot; should it ever be broken the search gives up after a bounded number of
// sweeps (slots are released by images that FINISH, so a short sleep between sweeps is all that helps) and returns -1:
// the image is reported with VPK_EM_NO_SLOT instead of hanging the GPU.
constexpr int EM_SLOT_SWEEPS = 1 << 16;
VPK_DEV int acquire_slot(const EmSliceArgs& ss) {
    Shared& sh = SH();
    if (tid() == 0) {
        int got = -1;
        const int start = (block_id() * 2) % ss.nslots;
        for (int sweep = 0; sweep < EM_SLOT_SWEEPS && got < 0; ++sweep) {
            for (int k = 0; k < ss.nslots && got < 0; ++k) {
                const int s = (start + k) % ss.nslots;
                if (atomicCAS(&ss.busy[s], 0, 1) == 0) got = s;
            }
            if (got < 0) __builtin_amdgcn_s_sleep(127);
        }
        sh.ibuf[7] = got;
    }
    block_sync();
    const int slot = sh.ibuf[7];
    block_sync();
    return slot;
}

// pop the next entry of a parked list (workgroup-uniform result; -1 = exhausted)
VPK_DEV int pop_entry(int* head, const int* count) {
    Shared& sh = SH();
    if (tid() == 0) {
        int e = -1;
        const int n = *count;
        if (n > 0) { e = atomicAdd(head, 1); if (e >= n) e = -1; }
        sh.ibuf[7] = e;
    }
    block_sync();
    const int e = sh.ibuf[7];
    block_sync();
    return e;
}

__global__ __launch_bounds__(EM_BOUND) void em_batch_kernel(EmBatchArgs a, EmSliceArgs ss) {
    VPK_SHARED_DECL;
    const long long t_start = clock_ticks();
    const long long deadline = (ss.enabled && ss.budget_ticks > 0) ? t_start + ss.budget_ticks : EM_NO_DEADLINE;
    bool started_left = ss.enabled != 0, waiting_left = ss.enabled != 0;
    for (;;) {
        EmCtx c;
        EmOut o;
        EmSlice sl;
        sl.deadline = deadline;
        sl.start_iter = -1;
        int slot = -1;
        bool have = false;
        // 1. images suspended by the previous launch (they own their slots), 2. images it could not start
        if (started_left) {
            const int e = pop_entry(&ss.ctr[0], &ss.ctr[1]);
            if (e >= 0) {
                const EmCarry& k = ss.in_started[e];
                c = k.c; o = k.o; sl.start_iter = k.iter; slot = k.slot;
                have = true;
            } else {
                started_left = false;
            }
        }
        if (!have && waiting_left) {
            const int e = pop_entry(&ss.ctr[3], &ss.ctr[4]);
            if (e >= 0) {
                const EmCarry& k = ss.in_waiting[e];
                c = k.c; o = k.o;
                have = true;
            } else {
                waiting_left = false;
            }
        }
        // 3. fresh images of this call
        if (!have) {
            if (tid() == 0) sh.ibuf[7] = atomicAdd(a.queue, 1);
            block_sync();
            const int q = sh.ibuf[7];
            block_sync();
            if (q >= a.B) break;
            const int img = a.order[q];
            const long long off = a.offsets[img];
            c.N = (int)(a.offsets[img + 1] - off);
            c.l = (gdp)(a.l + 3 * off);
            c.lp = (cgdp)(a.lp + 4 * off);
            c.cnn = (cgfp)(a.cnn + (size_t)img * NCELL);
            c.sphere = (cgbp)(a.sphere + (size_t)img * a.ssize * a.ssize);
            c.ssize = a.ssize;
            c.init_vp = a.init_vp ? (cgdp)(a.init_vp + (size_t)img * a.n_init * 3) : (cgdp) nullptr;
            c.n_init = a.n_init;
            c.prm = a.prm;
            c.wt_doubles = a.wt_doubles;
            c.smoother = a.smoother;
            o.max_vp = a.max_vp;
            o.vp = a.vp_out + (size_t)img * a.max_vp * 3;
            o.sigma = a.sigma_out + (size_t)img * a.max_vp;
            o.counts = a.counts_out + (size_t)img * a.max_vp;
            o.counts_w = a.counts_w_out + (size_t)img * a.max_vp;
            o.num_vp = a.num_vp_out + img;
            o.assoc = a.assoc_out + off;
            o.iterations = a.iterations_out + img;
            o.status = a.status_out + img;
            o.flags = a.flags_out + img;
            o.metric = a.metric_out ? a.metric_out + (size_t)off * a.max_vp : nullptr;
            o.trace = a.trace_out ? a.trace_out + (size_t)img * (a.prm.num_iter + 1) * TRACE_COLS : nullptr;
            if (a.dist.p_v) {
                o.d_pv = a.dist.p_v + (size_t)img * a.max_vp;
                o.d_angles = a.dist.angles + (size_t)img * a.max_vp * 2;
                o.d_pl = a.dist.p_l + off;
                o.d_plv = a.dist.p_lv + (size_t)off * a.max_vp;
                o.d_pvl = a.dist.p_vl + (size_t)off * a.max_vp;
                o.d_lvsq = a.dist.lvsq + (size_t)off * a.max_vp;
            }
        }
        // An image that has not been started when the deadline has passed is parked as it is -- unless its
        // list is full: then it runs now, TO COMPLETION (a longer launch, never a lost image; with the launch's
        // deadline it would take a slot, suspend at its first checkpoint and the workgroup would start the next one:
        // more suspended images than workgroups, which is what the slot count rules out).
        bool park_unstarted = false;
        bool list_full = false;
        if (ss.enabled && slot < 0 && deadline != EM_NO_DEADLINE) {
            if (tid() == 0) {
                int e = -1;
                if (clock_ticks() >= deadline) {
                    e = atomicAdd(&ss.ctr[5], 1);
                    if (e >= ss.cap_waiting) e = -2;
                }
                sh.ibuf[7] = e;
            }
            block_sync();
            const int e = sh.ibuf[7];
            block_sync();
            list_full = e == -2;
            if (e >= 0) {
                if (tid() == 0) {
                    EmCarry& k = ss.out_waiting[e];
                    k.c = c; k.o = o; k.iter = -1; k.slot = -1;
                }
                park_unstarted = true;
            }
        }
        if (park_unstarted) continue;
        if (list_full) sl.deadline = EM_NO_DEADLINE;
        if (slot < 0) {
            slot = ss.enabled ? acquire_slot(ss) : block_id();
            if (slot < 0) {                    // bounded wait expired (see acquire_slot): report, do not hang
                if (tid() == 0) { *o.status = VPK_EM_NO_SLOT; *o.num_vp = 0; *o.iterations = 0; *o.flags = 0; }
                block_sync();
                continue;
            }
            bind_scratch(c, a.scratch + (size_t)slot * a.L.total_doubles, a.L, c.prm.do_split != 0);
        }
        int result = em_run(c, o, sl);
        while (result == EM_SUSPENDED) {       // only with a deadline; at most one per workgroup and launch (see above)
            if (tid() == 0) {
                int e = atomicAdd(&ss.ctr[2], 1);
                if (e < ss.cap_started) {
                    EmCarry& k = ss.out_started[e];
                    k.c = c; k.o = o; k.iter = sl.start_iter; k.slot = slot;
                } else {
                    e = -1;                    // the list is full: this image is resumed right here and runs on
                }
                sh.ibuf[7] = e;
            }
            block_sync();
            const bool parked = sh.ibuf[7] >= 0;
            block_sync();
            if (parked) break;
            sl.deadline = EM_NO_DEADLINE;
            result = em_run(c, o, sl);         // resumes from the state it has just saved in its slot
        }
        if (result == EM_SUSPENDED) {
        } else if (ss.enabled) {
            block_sync();
            if (tid() == 0) { __threadfence(); atomicExch(&ss.busy[slot], 0); }
        }
    }
}

// between two time-sliced launches: what the last launch parked becomes the next launch's input list
__global__ void em_rotate_kernel(int* ctr, int cap_waiting, int cap_started) {
    ctr[1] = ctr[2] < cap_started ? ctr[2] : cap_started;   // (the counters keep counting when a list is full)
    ctr[0] = 0;
    ctr[2] = 0;
    ctr[4] = ctr[5] < cap_waiting ? ctr[5] : cap_waiting;   // the counter keeps counting when the list is full
    ctr[3] = 0;
    ctr[5] = 0;
}

// ---- fine-grained kernels (one workgroup, unit parity) ------------------------------------------
__global__ __launch_bounds__(EM_BOUND) void pairwise_kernel(int n, const double* lp, EmLayout L, double* ws,
                                                              double* lsim_out, double* lscore_out,
                                                              double* langle_out, int smoother) {
    EmCtx c;
    c.N = n; c.lp = (cgdp)lp; c.wt_doubles = WT_DOUBLES; c.smoother = smoother;
    c.prm.use_weights = 1;
    bind_scratch(c, ws, L, false);
    pairwise_setup(c, true);
    for (int p = tid(); p < n * n; p += nthreads()) lsim_out[p] = c.lsim[(size_t)(p / n) * c.ld + p % n];
    for (int i = tid(); i < n; i += nthreads()) { lscore_out[i] = c.lscore[i]; langle_out[i] = c.langle[i]; }
}

__global__ __launch_bounds__(EM_BOUND) void init_vps_kernel(const float* cnn, const unsigned char* sphere,
                                                              int ssize, int num_max, double* v0_out,
                                                              int* m0_out, float* weights_out) {
    VPK_SHARED_DECL;
    EmCtx c;
    c.N = 0; c.cnn = (cgfp)cnn; c.sphere = (cgbp)sphere; c.ssize = ssize; c.wt_doubles = WT_DOUBLES;
    c.prm.num_init_vp = num_max;
    initial_vps(c);
    for (int k = tid(); k < 3 * sh.M; k += nthreads()) v0_out[k] = sh.cur[k];
    if (tid() == 0) *m0_out = sh.M;
    block_sync();
    prior_setup(c);
    for (int k = tid(); k < NCELL; k += nthreads()) weights_out[k] = sh.wts[k];
}

__global__ __launch_bounds__(EM_BOUND) void estep_kernel(int n, int m, const double* lp, const float* cnn,
                                                           const double* v, double* s, EmLayout L, double* ws,
                                                           double* p_v_out, double* lvsq_out, double* p_vl_out,
                                                           double* p_l_out) {
    VPK_SHARED_DECL;
    EmCtx c;
    c.N = n; c.lp = (cgdp)lp; c.cnn = (cgfp)cnn; c.wt_doubles = WT_DOUBLES;
    c.prm.use_weights = 1;
    bind_scratch(c, ws, L, false);
    prior_setup(c);
    for (int k = tid(); k < n; k += nthreads()) c.lweight[k] = 1.0;
    for (int k = tid(); k < 3 * m; k += nthreads()) sh.cur[k] = v[k];
    for (int k = tid(); k < m; k += nthreads()) sh.s[k] = s[k];
    if (tid() == 0) sh.M = m;
    block_sync();
    line_geometry_setup(c);
    estep(c, sh.cur);
    for (int k = tid(); k < m; k += nthreads()) { s[k] = sh.s[k]; p_v_out[k] = sh.pv[k]; }
    for (int p = tid(); p < m * n; p += nthreads()) {
        int k = p / n, q = p % n;
        lvsq_out[p] = c.lvsq[(size_t)k * c.ldn + q];
        p_vl_out[p] = c.pvl[(size_t)k * c.ldn + q];
    }
    // p_l is not kept by the E-step; re-evaluate sum_m p_lv * p_v with the floor (:116-117)
    for (int q = tid(); q < n; q += nthreads()) {
        double pl = 0.0;
        for (int k = 0; k < m; ++k) {
            double lv = c.lvsq[(size_t)k * c.ldn + q];
            pl += exp(-(lv / (2 * sh.s[k]))) * sh.k2[k] * sh.pv[k];
        }
        p_l_out[q] = (pl > 1e-12 || pl != pl) ? pl : 1e-12;
    }
}

__global__ __launch_bounds__(EM_BOUND) void weight_matrix_kernel(int n, int m, const double* p_vl,
                                                                   const double* lweight, const double* lsim,
                                                                   double bias, EmLayout L, double* ws,
                                                                   double* w_out, int smoother, int wt_doubles) {
    VPK_SHARED_DECL;
    EmCtx c;
    c.N = n; c.wt_doubles = wt_doubles; c.smoother = smoother;
    c.prm.use_weights = 1;
    c.prm.wbias = bias;
    bind_scratch(c, ws, L, false);
    for (int p = tid(); p < n * n; p += nthreads())      // caller's matrix (row stride n) -> padded rows
        c.lsim[(size_t)(p / n) * c.ld + p % n] = lsim[p];
    if (tid() == 0) { sh.M = m; sh.ibuf[5] = 0; sh.ibuf[2] = 0; }   // no E-step ran: the operand panel is not in LDS
    for (int i = tid(); i < n; i += nthreads()) c.lweight[i] = lweight[i];
    for (int p = tid(); p < m * n; p += nthreads()) c.pvl[(size_t)(p / n) * c.ldn + p % n] = p_vl[p];   // (the sparse smoother's source)
    for (int p = tid(); p < n * c.mcap; p += nthreads()) {
        int i = p / c.mcap, k = p % c.mcap;
        c.wsrc[(size_t)i * c.mcap + k] = k < m ? p_vl[(size_t)k * n + i] * lweight[i] : 0.0;
    }
    block_sync();
    for (int k = tid(); k < n; k += nthreads()) {
        double sum = 0.0;
        for (int j = 0; j < n; ++j) sum += lsim[(size_t)j * n + k];
        c.den[k] = 1 + bias * c.lweight[k] * sum;
        if (!(fabs(sum) <= 1.7976931348623157e308)) sh.ibuf[2] = 1;       // (see weights_setup)
    }
    block_sync();
    zero_tail_rows(c);
    smooth(c);
    for (int p = tid(); p < m * n; p += nthreads()) w_out[p] = c.w[(size_t)(p / n) * c.ldn + p % n];
}

__global__ __launch_bounds__(EM_BOUND) void mstep_kernel(int n, int m, const double* l, const double* w,
                                                           EmLayout L, double* ws, double* vp_out,
                                                           int* valid_out) {
    VPK_SHARED_DECL;
    EmCtx c;
    c.N = n; c.l = (gdp) const_cast<double*>(l); c.wt_doubles = WT_DOUBLES;
    c.prm.s_thresh = 1e-200;
    bind_scratch(c, ws, L, false);
    if (tid() == 0) sh.M = m;
    for (int p = tid(); p < m * n; p += nthreads()) {
        int k = p / n, q = p % n;
        c.w[(size_t)k * c.ldn + q] = w[p];
        c.lvsq[(size_t)k * c.ldn + q] = 1.0;
        c.pvl[(size_t)k * c.ldn + q] = 1.0;
    }
    for (int k = tid(); k < 3 * m; k += nthreads()) { sh.cur[k] = (k % 3 == 2) ? 1.0 : 0.0; sh.nxt[k] = 0.0; }
    block_sync();
    mstep(c, 0, 1e-6);
    for (int k = tid(); k < m; k += nthreads()) {
        // "valid" mirrors calc_new_vanishing_point returning a vector (not None)
        bool none = sh.removed[k] && sh.err[k] == -1.0 && !(sh.s[k] != sh.s[k]);
        valid_out[k] = none ? 0 : 1;
        for (int d = 0; d < 3; ++d) vp_out[3 * k + d] = none ? 0.0 : sh.nxt[3 * k + d];
    }
}

// calc_vp_line_counts (vp_localisation.py:482-512) on its own: argmax VP per line, the outlier test against
// calc_lvsq_single of that VP (:504) and lweight == 0 (:506), counts and weighted counts per VP.
__global__ __launch_bounds__(EM_BOUND) void line_counts_kernel(int n, int m, const double* lp, const double* v,
                                                                 const double* s, const double* w, const double* lweight,
                                                                 double thresh, EmLayout L, double* ws, double* counts_out,
                                                                 double* counts_w_out, long long* assoc_out) {
    VPK_SHARED_DECL;
    EmCtx c;
    c.N = n; c.lp = (cgdp)lp; c.wt_doubles = WT_DOUBLES;
    c.prm.use_weights = 1;
    c.prm.outlier_thresh = thresh;
    bind_scratch(c, ws, L, false);
    for (int k = tid(); k < n; k += nthreads()) c.lweight[k] = lweight[k];
    for (int k = tid(); k < 3 * m; k += nthreads()) sh.cur[k] = v[k];
    for (int k = tid(); k < m; k += nthreads()) sh.s[k] = s[k];
    if (tid() == 0) { sh.M = m; sh.ncomp = 0; sh.sigma_prior = 1.0; }     // no prior: only lvsq is wanted from the E-step
    block_sync();
    line_geometry_setup(c);
    estep(c, sh.cur);                                                     // lvsq[m][n] (probability_functions.py:157-176)
    for (int p = tid(); p < m * n; p += nthreads()) c.w[(size_t)(p / n) * c.ldn + p % n] = w[p];
    block_sync();
    assign_lines(c, true);
    count_lines(c);
    for (int k = tid(); k < m; k += nthreads()) { counts_out[k] = sh.cnt[k]; counts_w_out[k] = sh.cntw[k]; }
    for (int k = tid(); k < n; k += nthreads()) assoc_out[k] = c.assoc[k];
}

__global__ __launch_bounds__(EM_BOUND) void cluster2_kernel(int n, double* D, int* member, int* csize,
                                                              int* labels_out, unsigned* flags_out) {
    VPK_SHARED_DECL;
    if (tid() == 0) sh.flags = 0;
    block_sync();
    const int ld = n | 1;
    if (n <= CLUSTER_LDS_MAX && cluster_lds_doubles(n) <= WT_DOUBLES) {   // same choice as split_vp
        double* DL = WT();
        for (int p = tid(); p < n * n; p += nthreads()) {
            const int a = p / n, b = p % n;
            const double v = D[p];
            DL[a * ld + b] = (a == b || !(v + D[(size_t)b * n + a] != 0.0)) ? -1.0 : v;
        }
        block_sync();
        cluster2_lds(n);
        const int* lmember = cluster_lds_labels(DL, n);
        for (int q = tid(); q < n; q += nthreads()) member[q] = lmember[q];
        block_sync();
    } else {
        cluster2(sh, n, (gdp)D, (gip)member, (gip)csize);
    }
    for (int q = tid(); q < n; q += nthreads()) labels_out[q] = member[q];
    if (tid() == 0) *flags_out = sh.flags;
}

template <typename K>
int allow_lds(vpk_handle* h, K kernel) {
    VPK_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)EM_LDS_BYTES_BIG));
    return VPK_OK;
}
int em_prepare(vpk_handle* h) {
    if (h->em_ready) return VPK_OK;
    int rc;
    if ((rc = allow_lds(h, em_batch_kernel))) return rc;
    if ((rc = allow_lds(h, pairwise_kernel))) return rc;
    if ((rc = allow_lds(h, init_vps_kernel))) return rc;
    if ((rc = allow_lds(h, estep_kernel))) return rc;
    if ((rc = allow_lds(h, weight_matrix_kernel))) return rc;
    if ((rc = allow_lds(h, mstep_kernel))) return rc;
    if ((rc = allow_lds(h, cluster2_kernel))) return rc;
    if ((rc = allow_lds(h, line_counts_kernel))) return rc;
    h->em_ready = true;
    return VPK_OK;
}

// One workgroup per CU (see EM_BOUND), with the whole remaining LDS (128 KiB) as the smoother's operand
// panel: single-pass smoothing for every image whose N x W panel fits.
struct EmMode { int per_cu; int wt_doubles; size_t lds_bytes; };
EmMode em_mode(const vpk_handle* h) {
    if (h->em_lds_doubles <= 0) return EmMode{1, WT_DOUBLES_BIG, EM_LDS_BYTES_BIG};
    // vpk_em_set_lds_panel: the budget the phases PLAN with; the launch still gets at least the setup phases' scratch
    const int wt = std::min(std::max(h->em_lds_doubles, 64), WT_DOUBLES_BIG);
    return EmMode{1, wt, SH_BYTES + (size_t)std::max(wt, PART_DOUBLES) * sizeof(double)};
}

int em_slots(const vpk_handle* h, int batch, size_t slot_bytes, int per_cu) {
    int slots = h->cu_share * per_cu;
    if (h->em_max_workgroups > 0 && slots > h->em_max_workgroups) slots = h->em_max_workgroups;
    if (slots > batch) slots = batch;
    size_t budget = h->total_mem / 2;                // never claim more than half of HBM
    while (slots > 1 && (size_t)slots * slot_bytes > budget) slots /= 2;
    return slots < 1 ? 1 : slots;
}

int check_params(vpk_handle* h, const vpk_em_params* p, int n_init, bool has_init) {
    if (!p) return vpk_fail(h, VPK_ERR_ARG, "params is null");
    if (p->num_iter < 1 || p->num_iter > 100000) return vpk_fail(h, VPK_ERR_ARG, "num_iter out of range");
    if (p->split_merge_freq < 1) return vpk_fail(h, VPK_ERR_ARG, "split_merge_freq < 1");
    if (p->num_init_vp < 1 || p->num_init_vp > MAXM) return vpk_fail(h, VPK_ERR_LIMIT, "num_init_vp must be 1..64");
    if (has_init && (n_init < 1 || n_init > MAXM)) return vpk_fail(h, VPK_ERR_LIMIT, "n_init must be 1..64");
    return VPK_OK;
}

// list capacities: vpk_handle::em_wait_cap (images parked before they were started; a full list makes further ones
// run on) and ::em_started_cap (suspended images: at most one per workgroup, vpk_em_set_workgroups <= CUs)
constexpr size_t EM_SESS_HEAD = 256;    // counters

struct SessView { int* ctr; int* busy; EmCarry* started[2]; EmCarry* waiting[2]; };
SessView sess_view(vpk_handle* h) {
    char* base = (char*)h->em_sess;
    SessView v;
    v.ctr = (int*)base;
    v.busy = (int*)(base + EM_SESS_HEAD);
    char* p = base + EM_SESS_HEAD + em_align((size_t)h->em_sess_slots * 4, 256);
    const size_t sb = em_align(sizeof(EmCarry) * h->em_started_cap, 256), wb = em_align(sizeof(EmCarry) * h->em_wait_cap, 256);
    v.started[0] = (EmCarry*)p; v.started[1] = (EmCarry*)(p + sb);
    v.waiting[0] = (EmCarry*)(p + 2 * sb); v.waiting[1] = (EmCarry*)(p + 2 * sb + wb);
    return v;
}
size_t sess_bytes(const vpk_handle* h, int slots) {
    return EM_SESS_HEAD + em_align((size_t)slots * 4, 256) + 2 * em_align(sizeof(EmCarry) * h->em_started_cap, 256) +
           2 * em_align(sizeof(EmCarry) * h->em_wait_cap, 256);
}
// the previous launch's output lists become this launch's input lists
EmSliceArgs sess_next(vpk_handle* h, double slice_ms) {
    SessView v = sess_view(h);
    hipLaunchKernelGGL(em_rotate_kernel, dim3(1), dim3(1), 0, h->stream, v.ctr, h->em_wait_cap, h->em_started_cap);
    h->em_sess_in ^= 1;
    EmSliceArgs ss;
    ss.enabled = 1;
    ss.budget_ticks = slice_ms > 0 ? (long long)(slice_ms * 1e-3 / (CLOCK_US * 1e-6)) : 0;
    if (slice_ms > 0 && ss.budget_ticks < 1) ss.budget_ticks = 1;
    ss.ctr = v.ctr;
    ss.in_started = v.started[h->em_sess_in]; ss.out_started = v.started[h->em_sess_in ^ 1];
    ss.in_waiting = v.waiting[h->em_sess_in]; ss.out_waiting = v.waiting[h->em_sess_in ^ 1];
    ss.cap_started = h->em_started_cap; ss.cap_waiting = h->em_wait_cap;
    ss.busy = v.busy; ss.nslots = h->em_sess_slots;
    return ss;
}

// launch that finishes every parked image (no deadline, no fresh images); asynchronous on the handle's stream
int em_flush(vpk_handle* h) {
    if (!h->em_unflushed) return VPK_OK;
    EmSliceArgs ss = sess_next(h, 0.0);
    EmBatchArgs a = {};
    a.B = 0;
    a.queue = ss.ctr + 8;                      // a counter that is never below B = 0
    a.L = h->em_sess_layout;
    a.scratch = (double*)h->em_ws;
    a.wt_doubles = h->em_sess_wt_doubles;
    a.smoother = h->em_smoother;
    hipLaunchKernelGGL(em_batch_kernel, dim3(h->em_sess_wgs), dim3(EM_THREADS), h->em_sess_lds, h->stream, a, ss);
    VPK_HIP(h, hipGetLastError());
    h->em_unflushed = false;
    return VPK_OK;
}

EmLayout small_layout(int n, int m) {
    return em_layout(n, (int)em_align((size_t)(m > 0 ? m : 1), 8), EM_WAVES, true, false);
}

}  // namespace

namespace {
// vpk_math_probe: the elementary functions exactly as this translation unit's kernels get them (same compiler flags, same
// ocml entry points as em_device.hpp's calls), one argument per thread.
__global__ void math_probe_kernel(int fn, long long n, const double* __restrict__ x, double* __restrict__ y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r;
    switch (fn) {
        case 0: r = exp(v); break;
        case 1: r = acos(v); break;
        case 2: r = asin(v); break;
        case 3: r = atan(v); break;
        case 4: r = sqrt(v); break;
        case 5: r = sin(v); break;
        case 6: r = cos(v); break;
        default: r = log(v); break;
    }
    y[i] = r;
}
}  // namespace

extern "C" {

int vpk_math_probe(vpk_handle* h, int fn, long long n, const double* x, double* y) {
    if (!h || fn < 0 || fn > 7 || n < 1 || !x || !y) return vpk_fail(h, VPK_ERR_ARG, "vpk_math_probe: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, fn, n, x, y);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_em_set_smoother(vpk_handle* h, int mode) {
    if (!h || mode < 0 || mode > 2) return vpk_fail(h, VPK_ERR_ARG, "vpk_em_set_smoother: mode must be 0, 1 or 2");
    h->em_smoother = mode;
    return VPK_OK;
}

int vpk_em_set_lds_panel(vpk_handle* h, int doubles) {
    if (!h || doubles < 0) return vpk_fail(h, VPK_ERR_ARG, "vpk_em_set_lds_panel: bad argument");
    if (h->em_unflushed) return vpk_fail(h, VPK_ERR_STATE, "vpk_em_set_lds_panel: images are parked: vpk_em_flush first");
    h->em_lds_doubles = doubles;
    return VPK_OK;
}

int vpk_em_set_workgroups(vpk_handle* h, int max_workgroups) {
    if (!h || max_workgroups < 0) return vpk_fail(h, VPK_ERR_ARG, "vpk_em_set_workgroups: bad argument");
    h->em_max_workgroups = max_workgroups;
    return VPK_OK;
}

size_t vpk_em_workspace_bytes(const vpk_handle* h, int batch, int n_max, const vpk_em_params* p, int n_init) {
    if (!h || !p || batch < 1) return 0;
    int mcap = em_mcap(p->num_init_vp, n_init, n_init > 0, p->do_split != 0, p->num_iter, p->split_merge_freq, MAXM);
    EmLayout L = em_layout(n_max, mcap, EM_WAVES, p->use_weights != 0, p->do_split != 0);
    size_t slot = L.total_doubles * sizeof(double);
    return (size_t)em_slots(h, batch, slot, em_mode(h).per_cu) * slot;
}

int vpk_em_batch(vpk_handle* h, int batch, const int64_t* offsets, double* l, const double* lp,
                 const float* cnn, const uint8_t* sphere, int sphere_size, const double* init_vp,
                 int n_init, const vpk_em_params* p, int max_vp, double* vp_out, double* sigma_out,
                 double* counts_out, double* counts_w_out, int32_t* num_vp_out, int64_t* assoc_out,
                 int32_t* iterations_out, int32_t* status_out, uint32_t* flags_out,
                 double* metric_out, double* trace_out) {
    if (!h) return VPK_ERR_ARG;
    if (batch < 1 || !offsets || !l || !lp || !cnn || !sphere || !vp_out || !sigma_out || !counts_out ||
        !counts_w_out || !num_vp_out || !assoc_out || !iterations_out || !status_out || !flags_out)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_em_batch: null buffer or batch < 1");
    if (sphere_size < GRIDN || max_vp < 1) return vpk_fail(h, VPK_ERR_ARG, "vpk_em_batch: bad sphere_size/max_vp");
    int rc = check_params(h, p, n_init, init_vp != nullptr);
    if (rc) return rc;
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    long long nmax = 0;
    for (int b = 0; b < batch; ++b) {
        long long n = offsets[b + 1] - offsets[b];
        if (n < 0) return vpk_fail(h, VPK_ERR_ARG, "vpk_em_batch: offsets not monotone");
        nmax = std::max(nmax, n);
    }
    if (nmax > 46000) return vpk_fail(h, VPK_ERR_LIMIT, "vpk_em_batch: more than 46000 lines in one image");
    const bool has_init = init_vp != nullptr;
    const bool sliced = h->em_slice_ms > 0.0;
    if (!sliced && h->em_unflushed) { rc = em_flush(h); if (rc) return rc; }
    if (sliced && (long long)h->em_slice_nmax > nmax) nmax = h->em_slice_nmax;
    int mcap = em_mcap(p->num_init_vp, n_init, has_init, p->do_split != 0, p->num_iter, p->split_merge_freq, MAXM);
    EmLayout L = em_layout((int)nmax, mcap, EM_WAVES, p->use_weights != 0, p->do_split != 0);
    const size_t slot_bytes = L.total_doubles * sizeof(double);
    const EmMode mode = em_mode(h);
    int slots = em_slots(h, batch, slot_bytes, mode.per_cu);
    int wgs = slots;
    if (sliced) {
        // the workgroups of a sliced launch also resume what earlier launches parked, so their number does not
        // follow the batch; slots outlive the launch: running (<= wgs) + parked and not yet resumed (<= wgs)
        wgs = h->cu_share * mode.per_cu;
        if (h->em_max_workgroups > 0 && wgs > h->em_max_workgroups) wgs = h->em_max_workgroups;
        slots = 2 * wgs + 8;
        if ((size_t)slots * slot_bytes > h->total_mem / 2)
            return vpk_fail(h, VPK_ERR_LIMIT, "vpk_em_batch: time-sliced slots exceed half of the device memory");
        const bool same = h->em_sess && h->em_sess_slot_bytes == slot_bytes && h->em_sess_slots == slots &&
                          h->em_sess_wgs == wgs && memcmp(&h->em_sess_layout, &L, sizeof(L)) == 0;
        if (!same) {
            if (h->em_unflushed)
                return vpk_fail(h, VPK_ERR_STATE, "vpk_em_batch: the slot layout changed (more lines than "
                                "vpk_em_set_time_slice was told, or other parameters) while images are parked: vpk_em_flush first");
            const size_t need = sess_bytes(h, slots);
            rc = vpk_reserve(h, &h->em_sess, &h->em_sess_bytes, need, "hipMalloc(EM session)");
            if (rc) return rc;
            VPK_HIP(h, hipMemsetAsync(h->em_sess, 0, need, h->stream));
            h->em_sess_slot_bytes = slot_bytes; h->em_sess_slots = slots; h->em_sess_wgs = wgs;
            h->em_sess_layout = L; h->em_sess_in = 0;
            h->em_sess_wt_doubles = mode.wt_doubles; h->em_sess_lds = mode.lds_bytes;
        }
    }
    if (h->em_unflushed && (size_t)slots * slot_bytes > h->em_ws_bytes)
        return vpk_fail(h, VPK_ERR_STATE, "vpk_em_batch: workspace would move while images are parked: vpk_em_flush first");
    rc = vpk_reserve(h, &h->em_ws, &h->em_ws_bytes, (size_t)slots * slot_bytes, "hipMalloc(EM workspace)");
    if (rc) return rc;
    // header: offsets (B+1 i64) | order (B i32) | queue counter
    const size_t off_bytes = em_align((size_t)(batch + 1) * 8, 256);
    const size_t ord_bytes = em_align((size_t)batch * 4, 256);
    rc = vpk_reserve(h, &h->em_hdr, &h->em_hdr_bytes, off_bytes + ord_bytes + 256, "hipMalloc(EM header)");
    if (rc) return rc;
    // The device header is rewritten by a stream-ordered copy (after the previous batch's kernel), so only
    // the pinned staging buffer needs care: take the next ring slot and wait for the copy that last used it.
    const int ring = h->em_hdr_next;
    h->em_hdr_next = (ring + 1) % vpk_handle::VPK_HDR_RING;
    if (!h->em_hdr_ev[ring]) VPK_HIP(h, hipEventCreateWithFlags(&h->em_hdr_ev[ring], hipEventDisableTiming));
    if (h->em_hdr_ev_valid[ring]) VPK_HIP(h, hipEventSynchronize(h->em_hdr_ev[ring]));
    if (h->em_hdr_host_bytes[ring] < off_bytes + ord_bytes) {
        if (h->em_hdr_host[ring]) VPK_HIP(h, hipHostFree(h->em_hdr_host[ring]));
        h->em_hdr_host[ring] = nullptr;
        h->em_hdr_host_bytes[ring] = 0;
        VPK_HIP(h, hipHostMalloc(&h->em_hdr_host[ring], (off_bytes + ord_bytes) * 2, hipHostMallocDefault));
        h->em_hdr_host_bytes[ring] = (off_bytes + ord_bytes) * 2;
    }
    long long* st_off = (long long*)h->em_hdr_host[ring];
    int* order = (int*)((char*)h->em_hdr_host[ring] + off_bytes);
    for (int b = 0; b <= batch; ++b) st_off[b] = offsets[b];
    std::iota(order, order + batch, 0);
    std::stable_sort(order, order + batch, [&](int x, int y) {
        return (offsets[x + 1] - offsets[x]) > (offsets[y + 1] - offsets[y]);   // largest image first
    });
    char* hdr = (char*)h->em_hdr;
    VPK_HIP(h, hipMemcpyAsync(hdr, h->em_hdr_host[ring], off_bytes + ord_bytes, hipMemcpyHostToDevice, h->stream));
    VPK_HIP(h, hipEventRecord(h->em_hdr_ev[ring], h->stream));
    h->em_hdr_ev_valid[ring] = true;
    VPK_HIP(h, hipMemsetAsync(hdr + off_bytes + ord_bytes, 0, 256, h->stream));

    EmBatchArgs a;
    a.B = batch;
    a.offsets = (const long long*)hdr;
    a.order = (const int*)(hdr + off_bytes);
    a.queue = (int*)(hdr + off_bytes + ord_bytes);
    a.l = l; a.lp = lp; a.cnn = cnn; a.sphere = sphere; a.ssize = sphere_size;
    a.init_vp = init_vp; a.n_init = n_init;
    a.prm = *p;
    a.L = L;
    a.scratch = (double*)h->em_ws;
    a.max_vp = max_vp;
    a.vp_out = vp_out; a.sigma_out = sigma_out; a.counts_out = counts_out; a.counts_w_out = counts_w_out;
    a.num_vp_out = num_vp_out; a.assoc_out = (long long*)assoc_out; a.iterations_out = iterations_out;
    a.status_out = status_out; a.flags_out = flags_out; a.metric_out = metric_out; a.trace_out = trace_out;
    a.dist = vpk_em_dist_out{};
    if (h->em_dist_set) {                  // consumed by this call
        a.dist = h->em_dist;
        h->em_dist_set = false;
    }
    a.wt_doubles = mode.wt_doubles;
    a.smoother = h->em_smoother;
    EmSliceArgs ss = {};
    if (sliced) {
        ss = sess_next(h, h->em_slice_ms);
        h->em_unflushed = true;
    }
    hipLaunchKernelGGL(em_batch_kernel, dim3(wgs), dim3(EM_THREADS), mode.lds_bytes, h->stream, a, ss);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_em_set_distribution_out(vpk_handle* h, const vpk_em_dist_out* d) {
    if (!h) return VPK_ERR_ARG;
    if (!d) { h->em_dist_set = false; return VPK_OK; }
    if (!d->p_v || !d->angles || !d->p_l || !d->p_lv || !d->p_vl || !d->lvsq)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_em_set_distribution_out: every buffer must be given");
    if (h->em_slice_ms > 0.0)
        return vpk_fail(h, VPK_ERR_STATE, "vpk_em_set_distribution_out: not available with time-sliced launches");
    h->em_dist = *d;
    h->em_dist_set = true;
    return VPK_OK;
}

int vpk_em_set_time_slice(vpk_handle* h, double slice_ms, int n_max) {
    if (!h || !(slice_ms >= 0.0) || n_max < 0) return vpk_fail(h, VPK_ERR_ARG, "vpk_em_set_time_slice: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    if (slice_ms == 0.0 && h->em_unflushed) { int rc = em_flush(h); if (rc) return rc; }
    h->em_slice_ms = slice_ms;
    h->em_slice_nmax = n_max;
    return VPK_OK;
}

int vpk_em_flush(vpk_handle* h) {
    if (!h) return VPK_ERR_ARG;
    VPK_HIP(h, hipSetDevice(h->device));
    return em_flush(h);
}

int vpk_pairwise(vpk_handle* h, int n, const double* lp, double* lsim_out, double* lscore_out,
                 double* langle_out) {
    if (!h || n < 1 || !lp || !lsim_out || !lscore_out || !langle_out) return vpk_fail(h, VPK_ERR_ARG, "vpk_pairwise: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    EmLayout L = small_layout(n, 8);
    int rc = vpk_reserve(h, &h->small_ws, &h->small_ws_bytes, L.total_doubles * 8, "hipMalloc(workspace)");
    if (rc) return rc;
    hipLaunchKernelGGL(pairwise_kernel, dim3(1), dim3(EM_THREADS), EM_LDS_BYTES, h->stream, n, lp, L, (double*)h->small_ws,
                       lsim_out, lscore_out, langle_out, h->em_smoother);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_init_vps(vpk_handle* h, const float* cnn, const uint8_t* sphere, int sphere_size, int num_max,
                 double* v0_out, int32_t* m0_out, float* weights_out) {
    if (!h || !cnn || !sphere || !v0_out || !m0_out || !weights_out || num_max < 1 || num_max > MAXM ||
        sphere_size < GRIDN)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_init_vps: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    hipLaunchKernelGGL(init_vps_kernel, dim3(1), dim3(EM_THREADS), EM_LDS_BYTES, h->stream, cnn, sphere, sphere_size, num_max,
                       v0_out, m0_out, weights_out);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_estep(vpk_handle* h, int n, int m, const double* lp, const float* cnn, const double* v, double* s,
              double* p_v_out, double* lvsq_out, double* p_vl_out, double* p_l_out) {
    if (!h || n < 1 || m < 1 || m > MAXM || !lp || !cnn || !v || !s || !p_v_out || !lvsq_out || !p_vl_out || !p_l_out)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_estep: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    EmLayout L = small_layout(n, m);
    int rc = vpk_reserve(h, &h->small_ws, &h->small_ws_bytes, L.total_doubles * 8, "hipMalloc(workspace)");
    if (rc) return rc;
    hipLaunchKernelGGL(estep_kernel, dim3(1), dim3(EM_THREADS), EM_LDS_BYTES, h->stream, n, m, lp, cnn, v, s, L,
                       (double*)h->small_ws, p_v_out, lvsq_out, p_vl_out, p_l_out);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_weight_matrix(vpk_handle* h, int n, int m, const double* p_vl, const double* lweight,
                      const double* lsim, double bias, double* w_out) {
    if (!h || n < 1 || m < 1 || m > MAXM || !p_vl || !lweight || !lsim || !w_out)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_weight_matrix: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    EmLayout L = em_layout(n, (int)em_align((size_t)m, 8), EM_WAVES, true, false);
    int rc = vpk_reserve(h, &h->small_ws, &h->small_ws_bytes, L.total_doubles * 8, "hipMalloc(workspace)");
    if (rc) return rc;
    // the batch kernel's LDS budget, so that this entry point takes the same smoother an image of this size takes there
    const EmMode mode = em_mode(h);
    hipLaunchKernelGGL(weight_matrix_kernel, dim3(1), dim3(EM_THREADS), mode.lds_bytes, h->stream, n, m, p_vl, lweight, lsim,
                       bias, L, (double*)h->small_ws, w_out, h->em_smoother, mode.wt_doubles);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_mstep(vpk_handle* h, int n, int m, const double* l, const double* w, double* vp_out, int32_t* valid_out) {
    if (!h || n < 1 || m < 1 || m > MAXM || !l || !w || !vp_out || !valid_out)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_mstep: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    EmLayout L = em_layout(n, (int)em_align((size_t)m, 8), EM_WAVES, false, false);
    int rc = vpk_reserve(h, &h->small_ws, &h->small_ws_bytes, L.total_doubles * 8, "hipMalloc(workspace)");
    if (rc) return rc;
    hipLaunchKernelGGL(mstep_kernel, dim3(1), dim3(EM_THREADS), EM_LDS_BYTES, h->stream, n, m, l, w, L, (double*)h->small_ws,
                       vp_out, valid_out);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_line_counts(vpk_handle* h, int n, int m, const double* lp, const double* v, const double* s, const double* w,
                    const double* lweight, double thresh, double* counts_out, double* counts_w_out, int64_t* assoc_out) {
    if (!h || n < 1 || m < 1 || m > MAXM || !lp || !v || !s || !w || !lweight || !counts_out || !counts_w_out || !assoc_out)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_line_counts: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    EmLayout L = small_layout(n, m);
    int rc = vpk_reserve(h, &h->small_ws, &h->small_ws_bytes, L.total_doubles * 8, "hipMalloc(workspace)");
    if (rc) return rc;
    hipLaunchKernelGGL(line_counts_kernel, dim3(1), dim3(EM_THREADS), EM_LDS_BYTES, h->stream, n, m, lp, v, s, w, lweight,
                       thresh, L, (double*)h->small_ws, counts_out, counts_w_out, (long long*)assoc_out);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_cluster2(vpk_handle* h, int n, const double* ldist, int32_t* labels_out, uint32_t* flags_out) {
    if (!h || n < 3 || !ldist || !labels_out || !flags_out) return vpk_fail(h, VPK_ERR_ARG, "vpk_cluster2: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    { int rc0 = em_prepare(h); if (rc0) return rc0; }
    size_t need = (size_t)n * n * 8 + (size_t)2 * n * 4 + 64;
    int rc = vpk_reserve(h, &h->small_ws, &h->small_ws_bytes, need, "hipMalloc(workspace)");
    if (rc) return rc;
    double* D = (double*)h->small_ws;
    int* member = (int*)(D + (size_t)n * n);
    VPK_HIP(h, hipMemcpyAsync(D, ldist, (size_t)n * n * 8, hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(cluster2_kernel, dim3(1), dim3(EM_THREADS), EM_LDS_BYTES, h->stream, n, D, member, member + n,
                       labels_out, flags_out);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

}  // extern "C"
