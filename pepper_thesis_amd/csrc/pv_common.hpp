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

// Kernel-form choices of a context (pv_set_option / pv_get_option, include/pepper_hip.h). Defaults come from the
// environment ONCE, in pv_create (PV_LSTM_SPLIT, PV_LSTM_ROWS, PV_TAIL_ROWS, PV_HEAD_SPLITS, PV_HEAD_MAP, PV_GRU_ROWS,
// PV_GRU_SPLIT, PV_GRU_USPLIT, PV_SHARED_DEVICE); a forward call never reads the environment.
struct pv_opts {
    int lstm_split = 1;          // 0: never use the unit-split LSTM form
    int lstm_rows = 0;           // 0 auto, 16 or 32: tile form of k_lstm_layer (an explicit form also disables the unit split)
    int tail_rows = 0;           // 0 auto, 16 or 32
    int head_splits = 0;         // 0 auto, 1 / 3 / 11 / 33
    int head_map = 1;            // XCD-aware order of k_head_splitk
    int gru_rows = 0;            // 0 auto, 16 or 32
    int gru_split = 1;           // 0: neither split form of the GRU
    int gru_usplit = 1;          // 0: no unit-split form (the direction-split form stays)
    int shared_device = 0;       // 1: other work shares this GPU: no form that needs all its workgroups resident at once
    int exchange_spin_log2 = 18; // bounded polls of the split forms give up after 2^n tries (layer hand-offs: 2^(n+8))
    int debug_drop_part = -1;    // diagnostic: this part of a unit-split launch never runs (forces exchange time-outs)
    int p1_bf16_min_batch = 513; // PV_DTYPE_BF16_INPUT_GEMM, P1: calls with fewer windows run the fp32 kernels (faster there: 0.85 ms against
                                 // 1.0 for 512 windows; results then are the fp32 mode's); 0: always the bf16x3 kernels
};

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
    pv_opts opt;
    std::vector<int64_t> upload_i64;        // host staging of pv_upload_batches' rebased small arrays (alive until the next upload:
    std::vector<uint8_t> upload_u8;         // the copies out of them are asynchronous)
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

// pv_api.hip: zero `bytes` (a multiple of 4) of device memory with an ordinary KERNEL on `st`. The *_dev entry points use it
// instead of hipMemsetAsync: inside a captured hipGraph a memset becomes a memset node, and replays were seen to run the kernel
// behind such a node on a stale buffer (flags of the previous replay still set: hand-offs passed at once; ROCm 7.2, gfx950) -
// kernel nodes keep their order.
int pv_zero_async(void* p, size_t bytes, hipStream_t st);

// rnn_kernels.hip
void pv_rnn_free(pv_ctx* ctx);
// rnn_gru.hip
void pv_rnn_free_p2(pv_ctx* ctx);
