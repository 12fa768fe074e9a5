// vpk_pipeline.hip -- one step of the reference's per-batch pipeline as ONE host call (see include/vpk.h:
// vpk_pipeline_step).  The reference runs run_cnn over the data set and then run_em (evaluation.py:254-329); a pipeline
// built on this library runs the EM of batch k beside the CNN of batch k + 1 on another stream.  Everything such a step
// enqueues -- the CNN forward, the cross-stream dependency, the working copy of the lines, the EM launch, the gather
// records -- is enqueued here in C++, so the host spends a few tens of microseconds per step, not milliseconds of
// interpreter time, and eight ranks on one host do not compete for it.
#include "vpk_internal.hpp"

namespace {

// Result record of one image, the layout of vanishing_points_2017_amd/sharding.py (pack_records / device_records):
// [image id, status, m, (x, y, z) x 20, count x 20, horizon error (NaN here)] with the VPs in descending order of their
// line counts (calc_horizon.py:34-36), equal counts in ascending index order, at most 20 (calc_horizon's maxbest).
constexpr int REC_VPS = 20;
constexpr int REC_WIDTH = 3 + REC_VPS * 4 + 1;

__global__ void records_kernel(int batch, int max_vp, const long long* __restrict__ image_ids,
                               const double* __restrict__ vp, const double* __restrict__ counts,
                               const int* __restrict__ num_vp, const int* __restrict__ status,
                               double* __restrict__ rec) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double* r = rec + (size_t)b * REC_WIDTH;
    for (int q = 0; q < REC_WIDTH; ++q) r[q] = 0.0;
    r[0] = (double)image_ids[b];
    r[1] = (double)status[b];
    int nv = num_vp[b];
    nv = nv < 0 ? 0 : (nv > max_vp ? max_vp : nv);
    const int m = nv < REC_VPS ? nv : REC_VPS;
    r[2] = (double)m;
    const double* c = counts + (size_t)b * max_vp;
    const double* v = vp + (size_t)b * max_vp * 3;
    unsigned long long taken = 0;                       // max_vp <= 64
    for (int k = 0; k < m; ++k) {                       // selection: largest count first, first index among equals
        int best = -1;
        for (int j = 0; j < nv; ++j)
            if (!((taken >> j) & 1ull) && (best < 0 || c[j] > c[best])) best = j;
        taken |= 1ull << best;
        r[3 + 3 * k] = v[3 * best];
        r[3 + 3 * k + 1] = v[3 * best + 1];
        r[3 + 3 * k + 2] = v[3 * best + 2];
        r[3 + 3 * REC_VPS + k] = c[best];
    }
    r[REC_WIDTH - 1] = __builtin_nan("");
}

}  // namespace

extern "C" {

int vpk_record_width(void) { return REC_WIDTH; }

int vpk_build_records(vpk_handle* h, int batch, int max_vp, const int64_t* image_ids, const double* vp,
                      const double* counts, const int32_t* num_vp, const int32_t* status, double* records) {
    if (!h || batch < 1 || max_vp < 1 || max_vp > 64 || !image_ids || !vp || !counts || !num_vp || !status || !records)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_build_records: bad argument");
    VPK_HIP(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(records_kernel, dim3((batch + 63) / 64), dim3(64), 0, h->stream, batch, max_vp,
                       (const long long*)image_ids, vp, counts, num_vp, status, records);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

int vpk_pipeline_step(vpk_handle* cnn, vpk_handle* em, const vpk_step_args* a) {
    if (!cnn || !em || !a) return VPK_ERR_ARG;
    if (cnn->device != em->device) return vpk_fail(em, VPK_ERR_ARG, "vpk_pipeline_step: the two handles must be on one GPU");
    if (!a->sphere || !a->cnn_out || !a->offsets || !a->l_in || !a->l_work || a->batch < 1)
        return vpk_fail(em, VPK_ERR_ARG, "vpk_pipeline_step: null buffer or batch < 1");
    VPK_HIP(em, hipSetDevice(em->device));
    hipEvent_t* ev = (hipEvent_t*)a->events;            // optional timing events of the caller
    hipEvent_t guard = (hipEvent_t)a->reuse_event;
    // 0. these buffers' previous step (if any) has finished before the CNN overwrites the response maps
    if (guard) VPK_HIP(cnn, hipStreamWaitEvent(cnn->stream, guard, 0));
    // 1. the CNN of this batch on the CNN handle's stream
    if (ev && ev[0]) VPK_HIP(cnn, hipEventRecord(ev[0], cnn->stream));
    int rc = vpk_cnn_forward(cnn, a->sphere, a->batch, a->cnn_out);
    if (rc) return rc;
    if (ev && ev[1]) VPK_HIP(cnn, hipEventRecord(ev[1], cnn->stream));
    // 2. the EM stream waits for it (no host wait)
    if (cnn->stream != em->stream) {
        if (!em->step_event) VPK_HIP(em, hipEventCreateWithFlags(&em->step_event, hipEventDisableTiming));
        VPK_HIP(em, hipEventRecord(em->step_event, cnn->stream));
        VPK_HIP(em, hipStreamWaitEvent(em->stream, em->step_event, 0));
    }
    // 3. the EM normalises the lines in place (vp_localisation.py:185-186): it works on a copy of the resident input
    const long long total = a->offsets[a->batch];
    if (a->l_work != a->l_in && total > 0)
        VPK_HIP(em, hipMemcpyAsync(a->l_work, a->l_in, (size_t)total * 3 * sizeof(double), hipMemcpyDeviceToDevice, em->stream));
    if (ev && ev[2]) VPK_HIP(em, hipEventRecord(ev[2], em->stream));
    // 4. the EM of this batch, its prior = the CNN's response maps
    rc = vpk_em_batch(em, a->batch, a->offsets, a->l_work, a->lp, a->em_prior ? a->em_prior : a->cnn_out, a->sphere, a->sphere_size, a->init_vp,
                      a->n_init, a->params, a->max_vp, a->vp_out, a->sigma_out, a->counts_out, a->counts_w_out,
                      a->num_vp_out, a->assoc_out, a->iterations_out, a->status_out, a->flags_out, nullptr, nullptr);
    if (rc) return rc;
    if (ev && ev[3]) VPK_HIP(em, hipEventRecord(ev[3], em->stream));
    // 5. the records the ranks gather (one all_gather per step is the path's only collective)
    if (a->records) {
        if (!a->image_ids) return vpk_fail(em, VPK_ERR_ARG, "vpk_pipeline_step: records without image_ids");
        rc = vpk_build_records(em, a->batch, a->max_vp, a->image_ids, a->vp_out, a->counts_out, a->num_vp_out,
                               a->status_out, a->records);
        if (rc) return rc;
    }
    if (guard) VPK_HIP(em, hipEventRecord(guard, em->stream));
    return VPK_OK;
}

}  // extern "C"
