// vpk_internal.hpp -- host-side state shared by the translation units of libvpk.so
#ifndef VPK_INTERNAL_HPP_
#define VPK_INTERNAL_HPP_

#include <vector>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>

#include "../../include/vpk.h"
#include "em_layout.hpp"

struct vpk_cnn_state;   // vpk_cnn.hip

struct vpk_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int num_cu = 0;
    int cu_share = 0;            // CUs this handle's launches are sized for (= num_cu)
    int em_max_workgroups = 0;   // vpk_em_set_workgroups (0 = no cap)
    int em_smoother = 0;         // vpk_em_set_smoother
    int em_lds_doubles = 0;      // vpk_em_set_lds_panel (0 = whole CU)
    // capacities of the time-sliced launches' parked-image lists (vpk_em.hip); the environment variables
    // VPK_EM_WAIT_CAP / VPK_EM_STARTED_CAP shrink them at vpk_create so that tests can fill them with a few images
    int em_wait_cap = 8192;
    int em_started_cap = 1024;
    int lds_per_block = 0;
    int arch = 0;
    size_t total_mem = 0;
    // EM workspace (grown on demand, never shrunk)
    void* em_ws = nullptr;
    size_t em_ws_bytes = 0;
    void* em_hdr = nullptr;   // offsets + order + queue counter
    size_t em_hdr_bytes = 0;
    // pinned staging for the header: a ring, so that the call does not have to wait for the previous
    // batch on this handle (only for the H2D copy issued VPK_HDR_RING calls ago)
    static constexpr int VPK_HDR_RING = 4;
    void* em_hdr_host[VPK_HDR_RING] = {};
    size_t em_hdr_host_bytes[VPK_HDR_RING] = {};
    hipEvent_t em_hdr_ev[VPK_HDR_RING] = {};
    bool em_hdr_ev_valid[VPK_HDR_RING] = {};
    int em_hdr_next = 0;
    void* small_ws = nullptr; // fine-grained entry points
    size_t small_ws_bytes = 0;
    bool raster_ready = false;
    void* raster_hdr = nullptr;
    size_t raster_hdr_bytes = 0;
    std::vector<long long> raster_offsets;   // the offsets the device copy in raster_hdr holds (same batch again: no upload, no wait)
    int raster_table_size = 0;               // canvas size the sample table in raster_hdr was made for
    int raster_last_batch = 0;               // batch of the last vpk_sphere_raster call (vpk_sphere_raster_flags' layout)
    int raster_alternative = 0;              // vpk_sphere_raster_set_alternative
    bool em_ready = false;    // dynamic-LDS attribute set on the EM kernels
    // time-sliced EM launches (vpk_em_set_time_slice): images not finished within a launch's budget are parked
    // in device-side lists and resumed by the next launch; their slots outlive the launch
    double em_slice_ms = 0.0;        // 0 = off: every call runs its images to completion
    vpk_em_dist_out em_dist = {};    // vpk_em_set_distribution_out: outputs of the next vpk_em_batch call
    bool em_dist_set = false;
    int em_slice_nmax = 0;           // slots are sized for at least this many lines
    void* em_sess = nullptr;         // [counters | slot flags | parked-image list A | list B]
    size_t em_sess_bytes = 0;
    int em_sess_in = 0;              // which list the next launch drains
    bool em_unflushed = false;       // parked images may exist
    size_t em_sess_slot_bytes = 0;   // geometry the parked images were laid out with
    int em_sess_slots = 0;
    int em_sess_wgs = 0;
    int em_sess_wt_doubles = 0;
    size_t em_sess_lds = 0;
    vpk::EmLayout em_sess_layout = {};
    hipEvent_t step_event = nullptr; // vpk_pipeline_step: orders the EM stream behind the CNN stream
    vpk_cnn_state* cnn = nullptr;
};

int vpk_fail(vpk_handle* h, int code, const char* what);
int vpk_fail_hip(vpk_handle* h, hipError_t e, const char* what);
int vpk_reserve(vpk_handle* h, void** p, size_t* have, size_t want, const char* what);
void vpk_cnn_free(vpk_handle* h);

#define VPK_HIP(h, call)                                         \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return vpk_fail_hip(h, e_, #call); \
    } while (0)

#endif
