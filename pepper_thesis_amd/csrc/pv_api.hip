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
    for (auto& r : c->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->prof.pool) (void)hipEventDestroy(e);
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

extern "C" int pv_profile_begin(pv_ctx* c) {
    PV_CHECK(c, PV_ERR_INVALID, "null context");
    for (auto& r : c->prof.recs) { c->prof.pool.push_back(r.a); c->prof.pool.push_back(r.b); }
    c->prof.recs.clear();
    c->prof.only.clear();
    c->prof.on = true;
    return PV_OK;
}

extern "C" int pv_profile_begin_only(pv_ctx* c, const char* prefix) {
    int rc = pv_profile_begin(c);
    if (rc == PV_OK && prefix) c->prof.only = prefix;
    return rc;
}

// Synchronises the context's stream, then aggregates the event-bracketed launches by kernel name.
// names: '\n'-separated list written into names_buf; ms_sum[i]/counts[i] follow the same order.
extern "C" int pv_profile_end(pv_ctx* c, char* names_buf, int buf_len, float* ms_sum, int* counts, int max_kernels) {
    PV_CHECK(c && names_buf && ms_sum && counts && buf_len > 0, PV_ERR_INVALID, "null argument");
    PV_HIP(hipSetDevice(c->device));
    PV_HIP(hipDeviceSynchronize());
    c->prof.on = false;
    std::vector<std::string> names;
    for (auto& r : c->prof.recs) {
        float ms = 0.f;
        PV_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        size_t k = 0;
        for (; k < names.size(); k++) if (names[k] == r.name) break;
        if (k == names.size()) {
            if ((int)k >= max_kernels) continue;
            names.push_back(r.name);
            ms_sum[k] = 0.f;
            counts[k] = 0;
        }
        ms_sum[k] += ms;
        counts[k] += 1;
    }
    std::string joined;
    for (size_t k = 0; k < names.size(); k++) { if (k) joined += "\n"; joined += names[k]; }
    snprintf(names_buf, (size_t)buf_len, "%s", joined.c_str());
    for (auto& r : c->prof.recs) { c->prof.pool.push_back(r.a); c->prof.pool.push_back(r.b); }
    c->prof.recs.clear();
    return (int)names.size();
}

extern "C" int64_t pv_workspace_bytes(pv_ctx* c) { return c ? (int64_t)c->arena.total : 0; }

// ---- hipGraph capture of a sequence of *_dev calls -------------------------------------------------------------------
struct pv_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int device = 0;
};

extern "C" int pv_graph_begin(pv_ctx* c, void* stream) {
    PV_CHECK(c, PV_ERR_INVALID, "null context");
    PV_CHECK(!c->arena.frozen, PV_ERR_STATE, "a capture is already open on this context");
    PV_CHECK(!c->prof.on, PV_ERR_STATE, "per-kernel profiling is on: its events cannot be part of a capture");
    PV_HIP(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    PV_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    c->arena.frozen = true;
    c->capture_stream = st;
    return PV_OK;
}

extern "C" int pv_graph_end(pv_ctx* c, pv_graph** out) {
    PV_CHECK(c && out, PV_ERR_INVALID, "null argument");
    PV_CHECK(c->arena.frozen, PV_ERR_STATE, "no capture is open on this context");
    c->arena.frozen = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(c->capture_stream, &g);
    c->capture_stream = nullptr;
    if (e != hipSuccess || !g) { pv_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return PV_ERR_HIP; }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); pv_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return PV_ERR_HIP; }
    pv_graph* pg = new pv_graph();
    pg->graph = g; pg->exec = x; pg->device = c->device;
    *out = pg;
    return PV_OK;
}

extern "C" int pv_graph_launch(pv_graph* g, void* stream) {
    PV_CHECK(g && g->exec, PV_ERR_INVALID, "null graph");
    PV_CHECK(stream, PV_ERR_INVALID, "pv_graph_launch needs the stream to replay on");
    PV_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return PV_OK;
}

extern "C" void pv_graph_destroy(pv_graph* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
