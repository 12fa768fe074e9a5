// em_layout.hpp -- per-slot HBM scratch layout of the EM workgroup (host + device).
//
// One "slot" = the private working set of one image while its workgroup runs.  Sized for the
// largest image of the batch; with 288 GB of HBM3E a few thousand slots of N = 1000 fit at once.
//   lsim   N x ld   fp64   (the dominant stream: read once per E-step)
//   pdist  N x ld   fp64   (closest distances of the line pairs; written and read once, by the setup)
//   lvsq / pvl / w  mcap x ldn fp64 ([vp][line]: lines contiguous -> coalesced per-line threads)
//   wsrc   N x mcap fp64   ([line][vp]: VPs contiguous -> wave-uniform broadcast in the smoother)
#ifndef VPK_EM_LAYOUT_HPP_
#define VPK_EM_LAYOUT_HPP_

#include <stddef.h>

namespace vpk {

struct EmLayout {
    int ldn, ld, mcap, nwaves;
    size_t lsim, pdist, den, lweight, langle, lscore, rowsum, lvsq, pvl, w, wsrc, drow, part, cl, assoc, idx;
    size_t lcopy, lpcopy, state;
    size_t total_doubles;
};

inline size_t em_align(size_t x, size_t a) { return (x + a - 1) / a * a; }

// doubles reserved per slot for the snapshot of the workgroup's LDS state (em_device.hpp: Shared) that a
// suspended image resumes from (time-sliced launches, vpk_em_set_time_slice)
constexpr size_t EM_STATE_DOUBLES = 2048;
// rows of lsim beyond the image's N (see em_layout)
constexpr size_t EM_LSIM_PAD_ROWS = 24;

inline int em_mcap(int num_init_vp, int n_init, bool has_init, bool do_split, int num_iter, int freq,
                   int maxm) {
    int m0 = has_init ? n_init : num_init_vp;
    int splits = 0;
    if (do_split && freq > 0) {
        // one split attempt at every i with i % freq == 0 and 0 < i < min(num_iter, 100)
        // (vp_localisation.py:256,262): floor(min(num_iter - 1, 99) / freq) of them -- 9 at the default
        // freq = 10, 99 at freq = 1.  The caller clamps to maxm; split_vp refuses to grow past mcap.
        int last = num_iter - 1 < 99 ? num_iter - 1 : 99;
        splits = last > 0 ? last / freq : 0;
    }
    int m = m0 + splits;
    if (m > maxm) m = maxm;
    if (m < 1) m = 1;
    return (int)em_align((size_t)m, 8);
}

inline EmLayout em_layout(int nmax, int mcap, int nwaves, bool use_weights, bool do_split) {
    EmLayout L;
    L.ldn = (int)em_align((size_t)(nmax > 0 ? nmax : 1), 8);
    L.ld = L.ldn;
    L.mcap = mcap;
    L.nwaves = nwaves;
    size_t o = 0;
    const size_t n = (size_t)L.ldn;
    // + 8 zero rows: the row-sliced smoother walks 8 ceil(N / 8) rows (zero_tail_rows).  + 16 more that are only ever READ:
    // the sparse smoother stages blocks of SP_R = 16 rows by DMA (rows up to 16 ceil(N / 16) - 1, and a 1 KB piece may run
    // past its row's ld doubles into the next row); their operand bits are zero, so the values never reach a sum, but the
    // addresses must belong to lsim whatever region follows it (static_assert in em_device.hpp ties SP_R to EM_LSIM_PAD_ROWS)
    L.lsim = o;    o += use_weights ? (n + EM_LSIM_PAD_ROWS) * n : 8;
    L.pdist = o;   o += use_weights ? n * n : 8;       // pair distances (setup only: the kNN rating reads rows of it)
    L.den = o;     o += n;
    L.lweight = o; o += n;
    L.langle = o;  o += n;
    L.lscore = o;  o += n;
    L.rowsum = o;  o += n;
    L.lvsq = o;    o += (size_t)mcap * n;
    L.pvl = o;     o += (size_t)mcap * n;
    L.w = o;       o += (size_t)mcap * n;
    L.wsrc = o;    o += n * (size_t)mcap;
    L.drow = o;    o += 6 * n;                         // E-step line constants [5][n] + p_l [n]
    L.part = o;    o += (size_t)nwaves * mcap * n;   // row-slice partials of the smoother
    L.cl = o;      o += do_split ? n * n : 8;
    L.assoc = o;   o += em_align(n, 2) / 2;            // n ints
    L.idx = o;     o += em_align(3 * n, 2) / 2;        // 3n ints
    L.lcopy = o;   o += 3 * n;                         // normalised lines: the image no longer depends on the
    L.lpcopy = o;  o += 4 * n;                         //   caller's l / lp buffers once its setup has run
    L.state = o;   o += EM_STATE_DOUBLES;
    L.total_doubles = em_align(o, 32);                 // 256-byte aligned slots
    return L;
}

}  // namespace vpk
#endif
