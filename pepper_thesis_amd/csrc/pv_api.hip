// pv_api.hip — context management entry points of the C-ABI (include/pepper_hip.h).
#include "pv_common.hpp"

static thread_local char g_err[512] = "";

void pv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pv_last_error(void) { return g_err; }
extern "C" int pv_version(void) { return PV_VERSION_NUM; }

extern "C" pv_ctx* pv_create(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        pv_set_error("no HIP device available (%s): the MI355X path has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        pv_set_error("device id %d out of range [0,%d)", device_id, n);
        return nullptr;
    }
    if (hipSetDevice(device_id) != hipSuccess) {
        pv_set_error("hipSetDevice(%d) failed", device_id);
        return nullptr;
    }
    pv_ctx* c = new pv_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&c->h_counts, 16 * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
        pv_set_error("stream / pinned buffer creation failed");
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void pv_destroy(pv_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    pv_rnn_free(c);
    c->arena.release();
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void* pv_stream(pv_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int pv_synchronize(pv_ctx* c) {
    PV_CHECK(c, PV_ERR_INVALID, "null context");
    PV_HIP(hipSetDevice(c->device));
    PV_HIP(hipStreamSynchronize(c->stream));
    return PV_OK;
}

extern "C" int64_t pv_workspace_bytes(pv_ctx* c) { return c ? (int64_t)c->arena.total : 0; }
