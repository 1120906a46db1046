// pv_common.hpp — context, workspace arena and error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pepper_hip.h"

#define PV_VERSION_NUM 100  // 0.1.0

void pv_set_error(const char* fmt, ...);

#define PV_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            pv_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PV_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define PV_CHECK(cond, code, ...)    \
    do {                             \
        if (!(cond)) {               \
            pv_set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

// A named, grow-only device buffer arena: one allocation per role, re-used across calls so that the
// steady state performs no hipMalloc/hipFree (graph-capture friendly, guide G9).
struct pv_arena {
    struct slot {
        void* p = nullptr;
        size_t bytes = 0;
    };
    std::map<std::string, slot> slots;
    size_t total = 0;
    bool frozen = false;   // a stream capture is open (pv_graph_begin): the workspace must already have its size

    int get(const char* name, size_t bytes, void** out) {
        slot& s = slots[name];
        if (bytes > s.bytes) {
            if (frozen) {
                pv_set_error("workspace %s would have to grow inside a graph capture: run the same calls once before pv_graph_begin", name);
                return PV_ERR_STATE;
            }
            if (s.p) {
                hipError_t e = hipFree(s.p);  // implicit device sync: only on growth
                if (e != hipSuccess) { pv_set_error("hipFree(%s): %s", name, hipGetErrorString(e)); return PV_ERR_HIP; }
                total -= s.bytes;
                s.p = nullptr;
                s.bytes = 0;
            }
            size_t want = bytes + bytes / 8 + 256;  // slack so that slightly larger batches do not realloc
            hipError_t e = hipMalloc(&s.p, want);
            if (e != hipSuccess) { pv_set_error("hipMalloc(%s, %zu): %s", name, want, hipGetErrorString(e)); return PV_ERR_HIP; }
            s.bytes = want;
            total += want;
        }
        *out = s.p;
        return PV_OK;
    }
    void release() {
        for (auto& kv : slots)
            if (kv.second.p) (void)hipFree(kv.second.p);
        slots.clear();
        total = 0;
    }
};

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
struct pv_prof {
    bool on = false;
    std::string only;   // non-empty: bracket only kernels whose profile name starts with this
    struct rec { const char* name; hipEvent_t a, b; };
    std::vector<rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t take() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};

struct pv_rnn_p1;  // rnn_kernels.hip
struct pv_rnn_p2;

struct pv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    pv_arena arena;
    int num_cu = 256;
    // pinned host staging for small read-backs
    int64_t* h_counts = nullptr;
    pv_rnn_p1* p1 = nullptr;
    pv_rnn_p2* p2 = nullptr;
    pv_prof prof;
    hipStream_t capture_stream = nullptr;   // pv_graph_begin .. pv_graph_end
};

// RAII bracket: { pv_prof_scope ps(ctx, "k_name", stream); kernel<<<...>>>(...); }
struct pv_prof_scope {
    pv_ctx* c; hipStream_t st; size_t idx; bool on;
    pv_prof_scope(pv_ctx* c_, const char* name, hipStream_t st_) : c(c_), st(st_), idx(0), on(c_->prof.on) {
        if (on && !c->prof.only.empty() && strncmp(name, c->prof.only.c_str(), c->prof.only.size()) != 0) on = false;
        if (!on) return;
        pv_prof::rec r; r.name = name; r.a = c->prof.take(); r.b = c->prof.take();
        (void)hipEventRecord(r.a, st);
        idx = c->prof.recs.size();
        c->prof.recs.push_back(r);
    }
    ~pv_prof_scope() { if (on) (void)hipEventRecord(c->prof.recs[idx].b, st); }
};

template <typename T>
static inline int pv_get(pv_ctx* c, const char* name, size_t n, T** out) {
    void* p = nullptr;
    int rc = c->arena.get(name, n * sizeof(T), &p);
    *out = (T*)p;
    return rc;
}

static inline hipStream_t pv_pick_stream(pv_ctx* c, void* stream) { return stream ? (hipStream_t)stream : c->stream; }

// rnn_kernels.hip
void pv_rnn_free(pv_ctx* ctx);
// rnn_gru.hip
void pv_rnn_free_p2(pv_ctx* ctx);
