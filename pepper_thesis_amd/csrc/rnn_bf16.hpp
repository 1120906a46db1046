// rnn_bf16.hpp — pieces of PV_DTYPE_BF16_INPUT_GEMM shared by the recurrent-network translation units.
//
// In this mode every matrix product runs on the bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulation and 3-term split
// operands (x = hi + lo, w = hi + lo: x.w ~= hi.hi + hi.lo + lo.hi; byte inputs are exact in bf16, so their products need
// two terms): the input projections as one large GEMM per layer (k_gemm_bf16x3, rnn_kernels.hip), the recurrent products
// inside k_rec_bf16 (rnn_rec_bf16.hip), linear_1 as a split-K GEMM. Cell updates, the small linear layers and the softmax
// stay fp32.
#pragma once
#include "pv_common.hpp"

// C = A . W^T (+ bias) through k_gemm_bf16x3. A [M][K], W [N][K] in "split8" rows (per 8 elements: 8 bf16 hi, 8 bf16 lo);
// C [splits][M][N] fp32 row-major, or (quads) [M/4][N][4]. M % 4 == 0, N % 256 == 0, K % (32 * splits) == 0.
struct pv_gemm_desc {
    const unsigned char* A;
    const unsigned char* W;
    const float* bias;   // [N] or NULL
    float* C;
    int64_t M;
    int N, K, splits, quads;
    const char* prof_name;
};
int pv_gemm_bf16x3_async(pv_ctx* ctx, const pv_gemm_desc& g, hipStream_t st);                       // rnn_kernels.hip
int pv_gemm_bf16x3_prepare();                                                                       // function attributes (once per load)
// w [N][K] fp32 row-major (host) -> split8 rows on the device; the allocation is appended to `owned`
int pv_upload_split8(const float* w, size_t N, size_t K, unsigned char** d_split, std::vector<void*>& owned);

// ---- k_rec_bf16: one recurrent layer on pre-packed bf16 weight fragments (rnn_rec_bf16.hip) ----------------------------
// A workgroup is (batch tile of 32 * MT rows, direction); wave w owns hidden units [32w, 32w + 32) of every gate.
struct pv_rec_desc {
    int cell;                  // 4: LSTM (gates i,f,g,o, hidden 256); 3: GRU (gates r,z,n, hidden 128)
    int enc;                   // 1: byte input, its projection computed in the step; 0: input projections pre-computed in G
    const float* G;            // !enc: quads [(T * Bp) / 4][2 * cell * hidden][4], row m = t * Bp + b, biases included
    const unsigned char* wp;   // packed fragment stream (pv_pack_rec_bf16)
    const unsigned char* wx;   // GRU encoder: resident x-part fragments
    const float* bias;         // enc: [2][cell * hidden] (LSTM b_ih + b_hh; GRU r,z: b_ih + b_hh, n: b_in)
    const float* bias_hn;      // GRU: [2][hidden] b_hn
    const void* x;             // enc: bytes [B][x_row_bytes]: step t of row b reads xf bytes at b * x_row_bytes + (x_t0 + t) * xf
    int64_t x_row_bytes;
    int x_t0, xf, x_signed;
    int64_t B, Bp;             // Bp: rows of a time slab (multiple of the tile rows)
    int T;
    const float* h0;           // [Bp][2][hidden] initial hidden state or NULL (zeros)
    float* h_out;              // [Bp][2][hidden] final hidden state or NULL
    float* out_f32;            // optional fp32 [Bp][T][2 * hidden]
    unsigned char* out_tm;     // optional split8, time-major [T][Bp][2 * hidden] (A operand of the next layer's GEMM)
    unsigned char* out_bm;     // optional split8, batch-major [Bp][T * 2 * hidden] (A operand of linear_1)
    int mt;                    // M-tiles of 32 rows per workgroup: 1 or 2
    const char* prof_name;
    int tr16;                  // 1 or 2: GRU on 16-row tiles, that many per workgroup (k_gru16_bf16; wp / wx from pv_pack_gru16_bf16, Bp a
                               // multiple of 16, mt unused)
    const unsigned char* dense_w;   // GRU decoder only: fragments of a [5][2 * hidden] dense layer (pv_pack_p2_dense) and ...
    float* dense_part;              // ... its partial logits [T][Bp / 32][2 dirs][4 waves][8][32 rows] (summed by k_p2_combine), or NULL
};
int pv_rec_bf16_async(pv_ctx* ctx, const pv_rec_desc& d, hipStream_t st);
int pv_rec_bf16_prepare();
// dirs[2] (PyTorch layout) -> device fragment stream(s). kx = real input features (enc) or 0.
int pv_pack_rec_bf16(const pv_rnn_dir* dirs, int cell, int kx, unsigned char** d_wp, unsigned char** d_wx, std::vector<void*>& owned);
int pv_pack_gru16_bf16(const pv_rnn_dir* dirs, int kx, unsigned char** d_wp, unsigned char** d_wx, std::vector<void*>& owned);

// ---- k_tail_bf16: the tail of the P1 head (sum of linear_1 slabs + bias + SELU, linear_2..5 + SELU, output layer, softmax) with
// the four 512 x 512 layers as 3-term split products (rnn_rec_bf16.hip). 64 rows per workgroup.
struct pv_tail_desc {
    const float* part;         // linear_1 slabs [splits][part_rows][512]
    int splits;
    int64_t part_rows;
    const float* b1;           // [512]
    const unsigned char* wp;   // packed linear_2..5 (pv_pack_tail_bf16)
    const float* b[4];         // [512] each
    const float* wo;           // [3][512]
    const float* bo;           // [3]
    float* probs;              // [B][3]
    int64_t B;
    unsigned* epoch;           // bumped once (the forward-call counter of the unit-split LSTM form) or NULL
    const int* err;            // non-zero: the probabilities leave as NaN (see k_head_tail) or NULL
};
int pv_tail_bf16_async(pv_ctx* ctx, const pv_tail_desc& d, hipStream_t st);
int pv_tail_bf16_prepare();
// w[4]: linear_2..5, [512][512] fp32 row-major (host) -> the fragment stream of k_tail_bf16
int pv_pack_tail_bf16(const float* const* w, unsigned char** d_wp, std::vector<void*>& owned);

// ---- P2 (bi-GRU polisher model) in this mode: device weights and the layer-wise forward (rnn_rec_bf16.hip) --------------
struct pv_p2_bf16_weights {
    unsigned char* enc_wp = nullptr;   // encoder W_hh fragment stream
    unsigned char* enc_wx = nullptr;   // encoder W_ih fragments (resident in registers)
    unsigned char* dec_wp = nullptr;   // decoder W_hh fragment stream
    float* enc_bias = nullptr;         // [2][384]: b_ir + b_hr, b_iz + b_hz, b_in
    float* enc_bias_hn = nullptr;      // [2][128]
    float* dec_bias_hn = nullptr;
    unsigned char* dec_wih_s = nullptr;   // decoder W_ih of both directions [768][256], split8 rows
    unsigned char* dense_frag = nullptr;  // dense1 as MFMA fragments of the decoder kernel (pv_pack_p2_dense)
    unsigned char* dense_frag16 = nullptr;   // ... of the 16-row decoder kernel (pv_pack_p2_dense16)
    unsigned char* enc16_wp = nullptr;    // the same layers packed for the 16-row form (pv_pack_gru16_bf16)
    unsigned char* enc16_wx = nullptr;
    unsigned char* dec16_wp = nullptr;
    float* dec_bias_cat = nullptr;        // [768]: per direction b_ir + b_hr, b_iz + b_hz, b_in
    const float* dense_w = nullptr;       // [5][256], [5] (owned by pv_rnn_p2)
    const float* dense_b = nullptr;
};
// dense_w [5][256] (host) -> [2 dirs][4 waves][2 k-steps][hi 1 KB | lo 1 KB]
int pv_pack_p2_dense(const float* dense_w, unsigned char** d_frag, std::vector<void*>& owned);
int pv_pack_p2_dense16(const float* dense_w, unsigned char** d_frag, std::vector<void*>& owned);
int pv_p2_bf16_forward(pv_ctx* ctx, const pv_p2_bf16_weights& w, const uint8_t* d_images, int64_t B, uint8_t* d_labels, float* d_acc,
                       hipStream_t st, int seq, int nwin, const float* d_hidden_in, float* d_hidden_out, float* d_logits);
