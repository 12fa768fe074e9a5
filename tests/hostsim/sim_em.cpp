// sim_em.cpp -- TEST-ONLY host build of the EM device source (see hip_sim.hpp).
#include "hip_sim.hpp"
#include "../../vanishing_points_2017_amd/csrc/em_device.hpp"

#include <stdlib.h>
#include <vector>

using namespace vpk;

#define g_sh (SH())
static std::vector<double> g_dbg;

static void make_ctx(EmCtx& c, std::vector<double>& buf, int n, const vpk_em_params& p, bool has_init,
                     int n_init) {
    int mcap = em_mcap(p.num_init_vp, n_init, has_init, p.do_split != 0, p.num_iter, p.split_merge_freq, MAXM);
    EmLayout L = em_layout(n, mcap, 1, p.use_weights != 0, p.do_split != 0);
    buf.assign(L.total_doubles, 0.0);
    memset(&c, 0, sizeof(c));
    c.N = n;
    c.prm = p;
    c.wt_doubles = WT_DOUBLES;
    if (const char* e = getenv("VPK_SIM_WT_DOUBLES")) c.wt_doubles = atoi(e);   // the LDS budget the phases plan with (vpk_em_set_lds_panel)
    bind_scratch(c, buf.data(), L, p.do_split != 0);
}

extern "C" {

int sim_em_single(int n, double* l, const double* lp, const float* cnn, const unsigned char* sphere,
                  int ssize, const double* init_vp, int n_init, const vpk_em_params* p, int max_vp,
                  double* vp_out, double* sigma_out, double* counts_out, double* counts_w_out,
                  int* num_vp_out, long long* assoc_out, int* iterations_out, int* status_out,
                  unsigned* flags_out, double* metric_out, double* trace_out) {
    EmCtx c;
    std::vector<double> buf;
    make_ctx(c, buf, n, *p, init_vp != nullptr, n_init);
    c.l = l; c.lp = lp; c.cnn = cnn; c.sphere = sphere; c.ssize = ssize;
    c.init_vp = init_vp; c.n_init = n_init;
    EmOut o;
    o.vp = vp_out; o.sigma = sigma_out; o.counts = counts_out; o.counts_w = counts_w_out;
    o.num_vp = num_vp_out; o.assoc = assoc_out; o.iterations = iterations_out; o.status = status_out;
    o.flags = flags_out; o.metric = metric_out; o.trace = trace_out; o.max_vp = max_vp;
    g_dbg.assign((size_t)p->num_iter * (1 + 4 * MAXM), 0.0);
    o.dbg = g_dbg.data();
    EmSlice sl;
    sl.deadline = EM_NO_DEADLINE;
    sl.start_iter = -1;
    if (getenv("VPK_SIM_SLICED")) {
        // time-sliced run: the stand-in clock always reads 0, so a deadline of 0 suspends the image at every
        // checkpoint; the LDS image is destroyed and the caller's l / lp arrays are poisoned between slices
        // (a suspended image must live in its slot only)
        sl.deadline = 0;
        int slices = 0;
        std::vector<double> lp_copy(lp, lp + 4 * (size_t)n);
        c.lp = lp_copy.data();
        std::vector<double> l_keep;
        while (em_run(c, o, sl) == EM_SUSPENDED) {
            memset(g_sim_lds, 0xff, sizeof(g_sim_lds));
            if (slices == 0) {
                l_keep.assign(l, l + 3 * (size_t)n);
                for (size_t q = 0; q < lp_copy.size(); ++q) lp_copy[q] = 1e300;
                for (size_t q = 0; q < 3 * (size_t)n; ++q) l[q] = -1e300;
            }
            ++slices;
        }
        if (slices) memcpy(l, l_keep.data(), l_keep.size() * sizeof(double));
        return slices;
    }
    em_run(c, o, sl);
    return 0;
}

const double* sim_last_states() { return g_dbg.data(); }

int sim_pairwise(int n, const double* lp, double* lsim_out, double* lscore_out, double* langle_out) {
    vpk_em_params p;
    memset(&p, 0, sizeof(p));
    p.use_weights = 1; p.num_init_vp = 25; p.num_iter = 1; p.split_merge_freq = 10;
    EmCtx c;
    std::vector<double> buf;
    make_ctx(c, buf, n, p, false, 0);
    c.lp = lp;
    pairwise_setup(c, true);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) lsim_out[(size_t)i * n + j] = c.lsim[(size_t)i * c.ld + j];
        lscore_out[i] = c.lscore[i];
        langle_out[i] = c.langle[i];
    }
    return 0;
}

int sim_init_vps(const float* cnn, const unsigned char* sphere, int ssize, int num_max, double* v0_out,
                 int* m0_out, float* weights_out) {
    vpk_em_params p;
    memset(&p, 0, sizeof(p));
    p.use_weights = 1; p.num_init_vp = num_max; p.num_iter = 1; p.split_merge_freq = 10;
    EmCtx c;
    std::vector<double> buf;
    make_ctx(c, buf, 8, p, false, 0);
    c.cnn = cnn; c.sphere = sphere; c.ssize = ssize;
    initial_vps(c);
    *m0_out = g_sh.M;
    for (int k = 0; k < 3 * g_sh.M; ++k) v0_out[k] = g_sh.cur[k];
    prior_setup(c);
    for (int k = 0; k < NCELL; ++k) weights_out[k] = g_sh.wts[k];
    return 0;
}

int sim_estep(int n, int m, const double* lp, const float* cnn, const double* v, double* s,
              double* p_v_out, double* lvsq_out, double* p_vl_out) {
    vpk_em_params p;
    memset(&p, 0, sizeof(p));
    p.use_weights = 1; p.num_init_vp = m; p.num_iter = 1; p.split_merge_freq = 10;
    EmCtx c;
    std::vector<double> buf;
    make_ctx(c, buf, n, p, false, 0);
    c.lp = lp; c.cnn = cnn;
    prior_setup(c);
    for (int k = 0; k < n; ++k) c.lweight[k] = 1.0;
    g_sh.M = m;
    for (int k = 0; k < 3 * m; ++k) g_sh.cur[k] = v[k];
    for (int k = 0; k < m; ++k) g_sh.s[k] = s[k];
    line_geometry_setup(c);
    estep(c, g_sh.cur);
    for (int k = 0; k < m; ++k) {
        s[k] = g_sh.s[k];
        p_v_out[k] = g_sh.pv[k];
        for (int q = 0; q < n; ++q) {
            lvsq_out[(size_t)k * n + q] = c.lvsq[(size_t)k * c.ldn + q];
            p_vl_out[(size_t)k * n + q] = c.pvl[(size_t)k * c.ldn + q];
        }
    }
    return 0;
}

int sim_weight_matrix(int n, int m, const double* p_vl, const double* lweight, const double* lsim,
                      double bias, double* w_out) {
    vpk_em_params p;
    memset(&p, 0, sizeof(p));
    p.use_weights = 1; p.num_init_vp = m; p.num_iter = 1; p.split_merge_freq = 10; p.wbias = bias;
    EmCtx c;
    std::vector<double> buf;
    make_ctx(c, buf, n, p, false, 0);
    g_sh.M = m;
    for (int i = 0; i < n; ++i) {
        c.lweight[i] = lweight[i];
        for (int j = 0; j < n; ++j) c.lsim[(size_t)i * c.ld + j] = lsim[(size_t)i * n + j];
        for (int k = 0; k < m; ++k) c.wsrc[(size_t)i * c.mcap + k] = p_vl[(size_t)k * n + i] * lweight[i];
    }
    for (int k = 0; k < n; ++k) {
        double sum = 0;
        for (int j = 0; j < n; ++j) sum += c.lsim[(size_t)j * c.ld + k];
        c.den[k] = 1 + bias * c.lweight[k] * sum;
        for (int q = m; q < c.mcap; ++q) c.wsrc[(size_t)k * c.mcap + q] = 0.0;
    }
    g_sh.ibuf[5] = 0;
    smooth(c);
    for (int k = 0; k < m; ++k)
        for (int q = 0; q < n; ++q) w_out[(size_t)k * n + q] = c.w[(size_t)k * c.ldn + q];
    return 0;
}

int sim_mstep(int n, int m, const double* l, const double* w, double* vp_out) {
    vpk_em_params p;
    memset(&p, 0, sizeof(p));
    p.use_weights = 1; p.num_init_vp = m; p.num_iter = 1; p.split_merge_freq = 10; p.s_thresh = 1e-200;
    EmCtx c;
    std::vector<double> buf;
    make_ctx(c, buf, n, p, false, 0);
    c.l = const_cast<double*>(l);
    g_sh.M = m;
    for (int k = 0; k < m; ++k)
        for (int q = 0; q < n; ++q) {
            c.w[(size_t)k * c.ldn + q] = w[(size_t)k * n + q];
            c.lvsq[(size_t)k * c.ldn + q] = 1.0;
            c.pvl[(size_t)k * c.ldn + q] = 1.0;
        }
    for (int k = 0; k < 3 * m; ++k) { g_sh.cur[k] = (k % 3 == 2) ? 1.0 : 0.0; g_sh.nxt[k] = 0.0; }
    mstep(c, 0, 1e-6);
    for (int k = 0; k < 3 * m; ++k) vp_out[k] = g_sh.nxt[k];
    return 0;
}

int sim_cluster2(int n, const double* ldist, int* labels_out, unsigned* flags_out) {
    std::vector<double> D(ldist, ldist + (size_t)n * n);
    std::vector<int> member(n), csize(n);
    g_sh.flags = 0;
    const int ld = n | 1;
    if (n <= CLUSTER_LDS_MAX && cluster_lds_doubles(n) <= WT_DOUBLES) {   // same choice as the kernel
        double* DL = WT();
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                const double v = D[(size_t)a * n + b];
                DL[a * ld + b] = (a == b || !(v + D[(size_t)b * n + a] != 0.0)) ? -1.0 : v;
            }
        cluster2_lds(n);
        const int* lmember = cluster_lds_labels(DL, n);
        for (int q = 0; q < n; ++q) member[q] = lmember[q];
    } else {
        cluster2(g_sh, n, D.data(), member.data(), csize.data());
    }
    for (int i = 0; i < n; ++i) labels_out[i] = member[i];
    *flags_out = g_sh.flags;
    return 0;
}

void sim_default_params(vpk_em_params* p) {
    p->num_iter = 100; p->do_merge = 1; p->do_split = 1; p->do_iterations = 1; p->use_weights = 1;
    p->num_init_vp = 25; p->split_merge_freq = 10; p->num_min_lines = 3; p->wbias = 1.0;
    p->merge_thresh = 1e-3; p->outlier_thresh = 1.96 * 1.96; p->final_convergence = 5e-3;
    p->s_thresh = 1e-200;
}

}  // extern "C"
