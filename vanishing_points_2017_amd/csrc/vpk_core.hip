// vpk_core.hip -- handle lifetime, stream plumbing, error reporting for libvpk.so
#include "vpk_internal.hpp"

#include <stdlib.h>
#include <string.h>

int vpk_fail(vpk_handle* h, int code, const char* what) {
    if (h) h->err = what;
    return code;
}
int vpk_fail_hip(vpk_handle* h, hipError_t e, const char* what) {
    if (h) h->err = std::string(what) + ": " + hipGetErrorString(e);
    return VPK_ERR_HIP;
}
int vpk_reserve(vpk_handle* h, void** p, size_t* have, size_t want, const char* what) {
    if (*have >= want && *p) return VPK_OK;
    if (*p) {
        VPK_HIP(h, hipStreamSynchronize(h->stream));
        VPK_HIP(h, hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    size_t grow = want + want / 8;
    hipError_t e = hipMalloc(p, grow);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        grow = want;
        e = hipMalloc(p, grow);
    }
    if (e != hipSuccess) return vpk_fail_hip(h, e, what);
    *have = grow;
    return VPK_OK;
}

extern "C" {

int vpk_version(void) { return VPK_VERSION; }

int vpk_create(int device, vpk_handle** out) {
    if (!out) return VPK_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return VPK_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return VPK_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VPK_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "libvpk: device %d is %s; this library ships gfx950 (MI355X) code objects only\n",
                device, prop.gcnArchName);
        return VPK_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return VPK_ERR_HIP;
    vpk_handle* h = new vpk_handle();
    h->device = device;
    h->num_cu = prop.multiProcessorCount;
    h->cu_share = h->num_cu;
    h->lds_per_block = (int)prop.sharedMemPerBlock;
    h->arch = 950;
    h->total_mem = prop.totalGlobalMem;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return VPK_ERR_HIP;
    }
    h->own_stream = true;
    if (const char* e = getenv("VPK_EM_WAIT_CAP")) { const int v = atoi(e); if (v >= 1 && v <= h->em_wait_cap) h->em_wait_cap = v; }
    if (const char* e = getenv("VPK_EM_STARTED_CAP")) { const int v = atoi(e); if (v >= 1 && v <= h->em_started_cap) h->em_started_cap = v; }
    *out = h;
    return VPK_OK;
}

int vpk_destroy(vpk_handle* h) {
    if (!h) return VPK_ERR_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    vpk_cnn_free(h);
    if (h->em_ws) (void)hipFree(h->em_ws);
    if (h->em_hdr) (void)hipFree(h->em_hdr);
    if (h->em_sess) (void)hipFree(h->em_sess);
    if (h->step_event) (void)hipEventDestroy(h->step_event);
    for (int i = 0; i < vpk_handle::VPK_HDR_RING; ++i) {
        if (h->em_hdr_host[i]) (void)hipHostFree(h->em_hdr_host[i]);
        if (h->em_hdr_ev[i]) (void)hipEventDestroy(h->em_hdr_ev[i]);
    }
    if (h->small_ws) (void)hipFree(h->small_ws);
    if (h->raster_hdr) (void)hipFree(h->raster_hdr);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return VPK_OK;
}

int vpk_set_stream(vpk_handle* h, void* hip_stream) {
    if (!h) return VPK_ERR_ARG;
    VPK_HIP(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) {
        VPK_HIP(h, hipStreamDestroy(h->stream));
        h->own_stream = false;
        h->stream = nullptr;
    }
    if (hip_stream == nullptr) {
        VPK_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    } else {
        h->stream = (hipStream_t)hip_stream;
    }
    return VPK_OK;
}

void* vpk_get_stream(const vpk_handle* h) { return h ? (void*)h->stream : nullptr; }

int vpk_synchronize(vpk_handle* h) {
    if (!h) return VPK_ERR_ARG;
    VPK_HIP(h, hipStreamSynchronize(h->stream));
    return VPK_OK;
}

const char* vpk_last_error(const vpk_handle* h) { return h ? h->err.c_str() : "null handle"; }

void vpk_em_default_params(vpk_em_params* p) {
    if (!p) return;
    p->num_iter = 100;          // vp_localisation.py:168
    p->do_merge = 1;
    p->do_split = 1;
    p->do_iterations = 1;
    p->use_weights = 1;
    p->num_init_vp = 25;        // :170
    p->split_merge_freq = 10;
    p->num_min_lines = 3;       // :172
    p->wbias = 1.0;
    p->merge_thresh = 1e-3;     // :171
    p->outlier_thresh = 1.96 * 1.96;
    p->final_convergence = 5e-3;
    p->s_thresh = 1e-200;
}

int vpk_device_info(const vpk_handle* h, int32_t info[4]) {
    if (!h || !info) return VPK_ERR_ARG;
    info[0] = h->num_cu;
    info[1] = h->lds_per_block;
    info[2] = h->arch;
    info[3] = (int32_t)(h->total_mem >> 30);
    return VPK_OK;
}

}  // extern "C"
