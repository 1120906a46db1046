// rnn_gru.hip — P2 (polisher bi-GRU) plan. Placeholder translation unit: entry points report
// PV_ERR_STATE until the kernels land (tracked in DESIGN.md).
#include "pv_common.hpp"

extern "C" int pv_rnn_load_p2(pv_ctx* ctx, const pv_weights_p2* w, int dtype) {
    (void)ctx; (void)w; (void)dtype;
    pv_set_error("P2 (bi-GRU) plan is not built yet");
    return PV_ERR_STATE;
}
extern "C" int pv_rnn_forward_p2(pv_ctx* ctx, const uint8_t* images, int64_t B, uint8_t* labels, float* acc) {
    (void)ctx; (void)images; (void)B; (void)labels; (void)acc;
    pv_set_error("P2 (bi-GRU) plan is not built yet");
    return PV_ERR_STATE;
}
extern "C" int pv_rnn_forward_p2_dev(pv_ctx* ctx, const uint8_t* d_images, int64_t B, uint8_t* d_labels, float* d_acc,
                                     void* stream) {
    (void)ctx; (void)d_images; (void)B; (void)d_labels; (void)d_acc; (void)stream;
    pv_set_error("P2 (bi-GRU) plan is not built yet");
    return PV_ERR_STATE;
}
