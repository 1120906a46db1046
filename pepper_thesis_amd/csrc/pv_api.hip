// pv_api.hip — context management entry points of the C-ABI (include/pepper_hip.h).
#include "pv_common.hpp"

#include <cstdlib>

static thread_local char g_err[512] = "";

void pv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pv_last_error(void) { return g_err; }
extern "C" int pv_version(void) { return PV_VERSION_NUM; }

namespace {
__global__ __launch_bounds__(256) void k_zero_words(unsigned* __restrict__ p, size_t n_words) {
    const size_t n4 = n_words / 4, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const bool al = ((uintptr_t)p & 15) == 0;
    if (al) {
        for (size_t i = i0; i < n4; i += stride) reinterpret_cast<u4*>(p)[i] = u4{0u, 0u, 0u, 0u};
        for (size_t i = n4 * 4 + i0; i < n_words; i += stride) p[i] = 0u;
    } else {
        for (size_t i = i0; i < n_words; i += stride) p[i] = 0u;
    }
}
}  // namespace

int pv_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return PV_OK;
    PV_CHECK(p && bytes % 4 == 0, PV_ERR_INVALID, "pv_zero_async: bad range");
    const size_t words = bytes / 4;
    const size_t blocks = (words / 4 + 255) / 256;
    k_zero_words<<<(unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks)), 256, 0, st>>>((unsigned*)p, words);
    PV_HIP(hipGetLastError());
    return PV_OK;
}

namespace {
struct opt_desc { const char* name; const char* env; int pv_opts::*field; int lo, hi; };
const opt_desc OPTS[] = {
    {"lstm_split", "PV_LSTM_SPLIT", &pv_opts::lstm_split, 0, 1},
    {"lstm_rows", "PV_LSTM_ROWS", &pv_opts::lstm_rows, 0, 32},
    {"tail_rows", "PV_TAIL_ROWS", &pv_opts::tail_rows, 0, 32},
    {"head_splits", "PV_HEAD_SPLITS", &pv_opts::head_splits, 0, 33},
    {"head_map", "PV_HEAD_MAP", &pv_opts::head_map, 0, 1},
    {"gru_rows", "PV_GRU_ROWS", &pv_opts::gru_rows, 0, 32},
    {"gru_split", "PV_GRU_SPLIT", &pv_opts::gru_split, 0, 1},
    {"gru_usplit", "PV_GRU_USPLIT", &pv_opts::gru_usplit, 0, 1},
    {"shared_device", "PV_SHARED_DEVICE", &pv_opts::shared_device, 0, 1},
    {"exchange_spin_log2", nullptr, &pv_opts::exchange_spin_log2, 2, 22},
    {"debug_drop_part", nullptr, &pv_opts::debug_drop_part, -1, 3},
    {"p1_bf16_min_batch", nullptr, &pv_opts::p1_bf16_min_batch, 0, 1 << 20},
};
bool opt_value_ok(const opt_desc& d, int v) {
    if (v < d.lo || v > d.hi) return false;
    if (!strcmp(d.name, "lstm_rows") || !strcmp(d.name, "tail_rows") || !strcmp(d.name, "gru_rows")) return v == 0 || v == 16 || v == 32;
    if (!strcmp(d.name, "head_splits")) return v == 0 || v == 1 || v == 3 || v == 11 || v == 33;
    return true;
}
}  // namespace

extern "C" int pv_set_option(pv_ctx* c, const char* name, int value) {
    PV_CHECK(c && name, PV_ERR_INVALID, "null argument");
    for (const opt_desc& d : OPTS)
        if (!strcmp(d.name, name)) {
            PV_CHECK(opt_value_ok(d, value), PV_ERR_INVALID, "option %s: value %d not allowed", name, value);
            c->opt.*(d.field) = value;
            return PV_OK;
        }
    pv_set_error("unknown option '%s'", name);
    return PV_ERR_INVALID;
}

extern "C" int pv_get_option(pv_ctx* c, const char* name, int* value) {
    PV_CHECK(c && name && value, PV_ERR_INVALID, "null argument");
    for (const opt_desc& d : OPTS)
        if (!strcmp(d.name, name)) { *value = c->opt.*(d.field); return PV_OK; }
    pv_set_error("unknown option '%s'", name);
    return PV_ERR_INVALID;
}

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
    // the environment supplies DEFAULTS for the kernel-form options, read here and nowhere else
    for (const opt_desc& d : OPTS)
        if (d.env)
            if (const char* ev = getenv(d.env)) {
                const int v = atoi(ev);
                if (opt_value_ok(d, v)) c->opt.*(d.field) = v;
            }
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
