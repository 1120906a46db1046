/*
 * pepper_hip.h — C-ABI of the MI355X-native PEPPER hot path.
 *
 * Two operator families live behind this boundary:
 *
 *   1. the pileup summary-image builder ("make_images" hot loop), replacing the pybind11 class
 *      PEPPER_VARIANT.RegionalSummaryGenerator
 *        reference: pepper_variant/modules/cpp/pybind_api.h:55-62 (binding),
 *                   pepper_variant/modules/cpp/region_summary.cpp:568-916 (generate_summary),
 *                   pepper_variant/modules/cpp/region_summary.cpp:337-566 (populate_summary_matrix)
 *   2. the recurrent-network inference step ("run_inference" hot loop), replacing
 *        transducer_model(images, False)   pepper_variant/modules/python/models/predict_distributed_gpu.py:65
 *        ort_session.run(...)              pepper_variant/modules/python/models/predict_distributed_cpu.py:85-88
 *      for plan P1 (pepper_variant 2x bi-LSTM + MLP head, models/simple_model.py:48-82) and
 *        transducer_model(image_chunk, hidden) in the sliding loop
 *                                          pepper/modules/python/models/predict.py:47-97
 *      for plan P2 (pepper polisher bi-GRU encoder/decoder, pepper/modules/python/models/simple_model.py:27-42).
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative PV_ERR_* code;
 *     pv_last_error() returns a thread-local message for the last failure on this thread.
 *   - the caller owns every buffer. Entry points ending in _dev take DEVICE pointers and a HIP stream
 *     (hipStream_t passed as void*, NULL = the context's own stream) and never synchronise with the host;
 *     the others take HOST pointers, stage through the context's workspace and return when results are
 *     in the caller's host buffers.
 *   - one pv_ctx = one HIP device + one stream + one workspace. Calls on distinct contexts are
 *     independent; calls on one context must not overlap.
 *   - there is NO CPU fallback: if no HIP device is present pv_create fails with PV_ERR_NO_DEVICE.
 */
#ifndef PEPPER_HIP_H
#define PEPPER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_OK 0
#define PV_ERR_INVALID (-1)      /* bad argument / malformed input            */
#define PV_ERR_NO_DEVICE (-2)    /* no HIP device or device id out of range   */
#define PV_ERR_HIP (-3)          /* a HIP runtime call failed                 */
#define PV_ERR_CAPACITY (-4)     /* caller output buffers too small (n_out / str_bytes hold the need) */
#define PV_ERR_LIMIT (-5)        /* an internal fixed limit was exceeded (message says which) */
#define PV_ERR_STATE (-6)        /* call order problem (e.g. forward before load) */

/* window geometry of the reference (pepper_variant/modules/python/Options.py:5-8,
 * region_summary.cpp:831 "candidate_window_size + 1") */
#define PV_WINDOW_ROWS 33
#define PV_FEATURES 26
#define PV_WINDOW_BYTES (PV_WINDOW_ROWS * PV_FEATURES)
#define PV_MAX_COLOR 125         /* region_summary.h:15-16 */
/* haplotag-aware variant (`-hp`): ImageSizeOptionsHP (Options.py:17-22): 48 planes, window 20 -> 21 rows
 * (region_summary_hp.cpp:946 "candidate_window_size + 1") */
#define PV_HP_WINDOW_ROWS 21
#define PV_HP_FEATURES 48
#define PV_HP_WINDOW_BYTES (PV_HP_WINDOW_ROWS * PV_HP_FEATURES)
#define PV_MAX_ALLELE_KEY 61     /* region_summary.cpp:461,511 "candidate_string.length() <= 61" */

/* CIGAR op codes = BAM codes = CIGAR_OPERATIONS (pepper_variant/modules/cpp/cigar.h:15-27) */
#define PV_CIGAR_MATCH 0
#define PV_CIGAR_IN 1
#define PV_CIGAR_DEL 2
#define PV_CIGAR_REF_SKIP 3
#define PV_CIGAR_SOFT_CLIP 4
#define PV_CIGAR_HARD_CLIP 5
#define PV_CIGAR_PAD 6
#define PV_CIGAR_EQUAL 7
#define PV_CIGAR_DIFF 8
#define PV_CIGAR_BACK 9

typedef struct pv_ctx pv_ctx;

/* ---- image builder ------------------------------------------------------------------------- */

/* The scalar arguments of RegionalSummaryGenerator::generate_summary
 * (region_summary.h:191-206; call site AlignmentSummarizer.py:223-238), same order, same types. */
typedef struct pv_params {
    double min_snp_baseq;
    double min_indel_baseq;
    double snp_freq_threshold;
    double insert_freq_threshold;
    double delete_freq_threshold;
    double min_coverage_threshold;
    double snp_candidate_freq_threshold;
    double indel_candidate_freq_threshold;
    double candidate_support_threshold;
    int32_t skip_indels;
    int32_t candidate_window_size; /* must be 32 */
    int32_t feature_size;          /* must be 26 */
    int32_t reserved;
} pv_params;

/* A batch of regions in flat SoA form. Region g owns reads [read_off[g], read_off[g+1]) and reference
 * bytes ref[ref_off[g] .. ref_off[g+1]). This replaces the by-value `vector<type_read>` argument
 * (read.h:60-108: pos, flags.is_reverse, mapping_quality, sequence, base_qualities, cigar_tuples) and
 * the constructor arguments (region_summary.cpp:9-17: region_start, region_end, reference_sequence).
 *   ref_end is INCLUSIVE; R = ref_end - ref_start + 1; ref_off[g+1]-ref_off[g] must be >= R.
 *   cigar words use BAM packing: (length << 4) | op.
 *   bases are normally the upper-case symbols bam_handler.cpp emits (seq_nt16_str "=ACMGRSVTWYHKDBN"),
 *   but ANY byte is handled exactly as the reference would (raw-byte SNP keys, toupper for planes). */
/* Limits: a region may hold at most 32767 reads (the per-column counters are 16-bit; the reference's caller keeps at most
 * MAX_READS_IN_REGION = 5000, pepper_variant/modules/python/Options.py:98, AlignmentSummarizer.py:191-208). Beyond that the
 * host-buffer forms return PV_ERR_LIMIT and the device-resident forms report status PV_ERR_LIMIT in d_counts[2]. */
typedef struct pv_batch_in {
    int32_t n_regions;
    int32_t reserved;
    const int64_t* ref_start;   /* [n_regions] */
    const int64_t* ref_end;     /* [n_regions] inclusive */
    const int64_t* cand_start;  /* [n_regions] candidate_region_start */
    const int64_t* cand_end;    /* [n_regions] candidate_region_end (inclusive) */
    const int64_t* ref_off;     /* [n_regions+1] */
    const uint8_t* ref;         /* reference bytes */
    const int64_t* read_off;    /* [n_regions+1] */
    const int64_t* read_pos;    /* [n_reads] type_read::pos */
    const uint8_t* read_flags;  /* [n_reads] bit0 = flags.is_reverse */
    const uint8_t* read_mapq;   /* [n_reads] type_read::mapping_quality */
    const int64_t* base_off;    /* [n_reads+1] into bases/quals */
    const uint8_t* bases;       /* type_read::sequence */
    const uint8_t* quals;       /* type_read::base_qualities (raw phred) */
    const int64_t* cigar_off;   /* [n_reads+1] into cigar */
    const uint32_t* cigar;      /* type_read::cigar_tuples */
} pv_batch_in;

/* One output record per surviving candidate allele = one CandidateImageSummary
 * (region_summary.h:88-111), in the reference's order (regions in batch order, sites ascending,
 * alleles in std::set<std::string> order). images are already cast to int8 with wrap-around as
 * DataStore.write_summary does (pepper_variant/modules/python/DataStore.py:68). */
typedef struct pv_batch_out {
    int64_t capacity;      /* in: number of windows the arrays below can hold */
    int64_t str_capacity;  /* in: bytes cand_str can hold */
    int32_t* region;       /* [capacity] index of the region in the batch */
    int64_t* position;     /* [capacity] CandidateImageSummary::position */
    uint8_t* depth;        /* [capacity] min(coverage,125) */
    uint8_t* cand_freq;    /* [capacity] candidate_frequency[0] = min(allele_depth,125) */
    int8_t* images;        /* [capacity][33][26] */
    int32_t* images_i32;   /* optional (may be NULL): the un-cast int values of image_matrix */
    char* cand_str;        /* allele keys, concatenated: "1T", "2AGG", "3CAA" ... (candidates[0]) */
    int64_t* cand_off;     /* [capacity+1] */
    int64_t n_out;         /* out: number of windows produced (or needed on PV_ERR_CAPACITY) */
    int64_t str_bytes;     /* out: bytes of cand_str produced (or needed) */
} pv_batch_out;

pv_ctx* pv_create(int device_id);
void pv_destroy(pv_ctx* ctx);
const char* pv_last_error(void);
/* returns the HIP stream (hipStream_t) the context launches on */
void* pv_stream(pv_ctx* ctx);
int pv_synchronize(pv_ctx* ctx);

/* generate_summary for a batch of regions, HOST buffers in and out. */
int pv_summarize_regions(pv_ctx* ctx, const pv_batch_in* in, const pv_params* params, pv_batch_out* out);

/* Device-resident form used by the fused pipeline and the benchmark: every pointer inside `in` and
 * `out` (the arrays, not the structs) is a DEVICE pointer; totals that the host form derives by
 * reading the offset arrays are passed explicitly. Asynchronous on `stream`; results counters are
 * written to the four-element DEVICE array `d_counts` = {n_out, str_bytes, status, reserved}
 * (out->n_out etc. are not touched). status is PV_OK or a PV_ERR_* code detected on the device
 * (malformed read, workspace limit). Windows beyond out->capacity are dropped (d_counts[0] still
 * holds the number needed). */
int pv_summarize_regions_dev(pv_ctx* ctx, const pv_batch_in* in, const pv_params* params,
                             int64_t n_reads, int64_t n_bases, int64_t n_cigar, int64_t n_ref_bytes,
                             int64_t max_region_len, pv_batch_out* out, int64_t* d_counts, void* stream);

/* Stage a HOST batch for the device-resident form: copies the arrays of `host` into the context's workspace (asynchronously
 * on `stream`, after validating the offset arrays on the host) and fills `dev` with the same struct holding DEVICE pointers,
 * valid until the next pv_upload_batch on this context; totals4 = {n_reads, n_bases, n_cigar, n_ref_bytes}, the totals
 * pv_summarize_regions_dev takes. With pv_summarize_regions_dev and pv_rnn_forward_p1_dev behind it this is the fused
 * call_variant step: reads in, probabilities out, windows never on the host. */
int pv_upload_batch(pv_ctx* ctx, const pv_batch_in* host, pv_batch_in* dev, int64_t* totals4, void* stream);
/* The same for a batch that arrives in n_parts host batches (e.g. one per interval from reader threads): their regions are laid
 * end to end in part order; the large arrays are copied part by part to their offsets on the device, so the caller never
 * concatenates them on the host. The previous upload's copies must have completed (synchronise the stream) before the next
 * call on the same context. */
int pv_upload_batches(pv_ctx* ctx, int n_parts, const pv_batch_in* const* parts, pv_batch_in* dev, int64_t* totals4, void* stream);

/* ---- haplotag-aware image builder (`make_images -hp`) --------------------------------------------
 * Replaces PEPPER_VARIANT.RegionalSummaryGeneratorHP (pybind_api.h:64-71; region_summary_hp.cpp:350-663 populate_summary_matrix,
 * :665-1012 generate_summary; call site AlignmentSummarizerHP.py:215-233). Same flat batch and scalar struct as
 * pv_summarize_regions plus one int per read, `read_hp` = type_read::hp_tag (the HP aux tag, 0 when absent; NULL = all 0):
 *   48 planes = {REF, SNP, INS, DEL overlays 0-3} + 4 x {REF count, A, C, G, T, I, D, *} for (HP1 fwd, HP1 rev, HP2 fwd,
 *   HP2 rev) with 3 overlay planes in front of each group; an untagged read (hp 0) counts in both haplotypes;
 *   params->candidate_window_size must be 20 and params->feature_size 48; out->images is [capacity][21][48]
 *   (images_i32 likewise); every plane is clamped to +-125 (region_summary_hp.cpp:762-767).
 * Results are bit-identical with the reference class on the same reads. */
int pv_summarize_regions_hp(pv_ctx* ctx, const pv_batch_in* in, const int32_t* read_hp, const pv_params* params,
                            pv_batch_out* out);
/* device-resident, asynchronous form: see pv_summarize_regions_dev; read_hp is a DEVICE pointer (or NULL) */
int pv_summarize_regions_hp_dev(pv_ctx* ctx, const pv_batch_in* in, const int32_t* read_hp, const pv_params* params,
                                int64_t n_reads, int64_t n_bases, int64_t n_cigar, int64_t n_ref_bytes,
                                pv_batch_out* out, int64_t* d_counts, void* stream);

/* ---- P2 (polisher) summary images -------------------------------------------------------------
 * Replaces SummaryGenerator::generate_summary + generate_image
 * (pepper/modules/src/pileup_summary/summary_generator.cpp:47-121, 274-304, 371-392) and
 * AlignmentSummarizer.chunk_images (pepper/modules/python/AlignmentSummarizer.py:19-56) for a batch of
 * regions. Input is the same pv_batch_in struct; quals, cand_start and cand_end are not read, and the reference
 * bytes are only needed for their length, as in the reference. Every reference position of a region
 * gives one image row followed by `longest insert anchored there` insert rows; a row is 10 uint8:
 *   0-3 A,C,G,T reverse  4-7 A,C,G,T forward  8 other/deleted reverse  9 other/deleted forward,
 *   value = (uint8) (count / max(1, coverage[position]) * 254), the double->uint8 conversion taken as
 *   truncation to int32 followed by the low byte (what the x86-64 build of the reference does; only
 *   reachable where deletions cover a column no read base covers).
 * The rows of a region are cut into chunks of seq_length rows that overlap by seq_overlap rows; the
 * last chunk is padded with zero rows whose position/index are -1. */
typedef struct pv_polish_out {
    int64_t chunk_capacity; /* in: chunks the chunk arrays can hold */
    int64_t row_capacity;   /* in: rows the flat arrays can hold (0 when the flat arrays are NULL) */
    uint8_t* images;        /* [chunk_capacity][seq_length][10] */
    int64_t* position;      /* [chunk_capacity][seq_length] genomic_pos.first */
    int32_t* index;         /* [chunk_capacity][seq_length] genomic_pos.second (0 = base row, k = k-th insert row) */
    int32_t* region;        /* [chunk_capacity] region of the batch */
    int32_t* chunk_id;      /* [chunk_capacity] chunk number inside its region */
    uint8_t* flat_images;   /* optional [row_capacity][10]: SummaryGenerator::image of all regions, concatenated */
    int64_t* flat_position; /* optional [row_capacity] */
    int32_t* flat_index;    /* optional [row_capacity] */
    int64_t* region_row_off;/* optional [n_regions+1] first flat row of every region */
    int64_t n_chunks;       /* out: chunks produced (or needed on PV_ERR_CAPACITY) */
    int64_t n_rows;         /* out: flat rows produced (or needed) */
} pv_polish_out;

/* HOST buffers in and out. */
int pv_polish_summarize_regions(pv_ctx* ctx, const pv_batch_in* in, int seq_length, int seq_overlap, pv_polish_out* out);
/* Device-resident, asynchronous form (see pv_summarize_regions_dev): d_counts = {n_chunks, n_rows, status, insert rows}. */
int pv_polish_summarize_regions_dev(pv_ctx* ctx, const pv_batch_in* in, int64_t n_reads, int64_t n_bases, int64_t n_cigar,
                                    int64_t n_ref_bytes, int seq_length, int seq_overlap, pv_polish_out* out,
                                    int64_t* d_counts, void* stream);

/* ---- recurrent-network inference ------------------------------------------------------------ */

#define PV_PLAN_P1_LSTM 1 /* pepper_variant: 2x bi-LSTM(256) + 5xLinear(512)/SELU + Linear(3) + softmax */
#define PV_PLAN_P2_GRU 2  /* pepper polisher: bi-GRU(128) encoder+decoder + Linear(256->5), sliding 100/50 over 1000 */

#define PV_DTYPE_F32 0            /* every product in fp32 (f32 MFMA == fmaf chain) */
#define PV_DTYPE_BF16_INPUT_GEMM 1 /* bf16 operands (fp32 accumulate) for the input-projection GEMMs only */

/* Weights in PyTorch state_dict layout (row-major, fp32), i.e. exactly the tensors
 * ModelHander.load_simple_model_for_training (pepper_variant/modules/python/models/ModelHander.py:18-44)
 * obtains from the checkpoint. Index [0] = forward direction, [1] = "_reverse". HOST pointers. */
typedef struct pv_rnn_dir {
    const float* w_ih; /* [G*H, K]  (G = 4 for LSTM gates i,f,g,o; 3 for GRU gates r,z,n) */
    const float* w_hh; /* [G*H, H] */
    const float* b_ih; /* [G*H] */
    const float* b_hh; /* [G*H] */
} pv_rnn_dir;

typedef struct pv_weights_p1 {
    pv_rnn_dir encoder[2]; /* LSTM(26 -> 256)  simple_model.py:23-27 */
    pv_rnn_dir decoder[2]; /* LSTM(512 -> 256) simple_model.py:28-32 */
    const float* linear_w[5]; /* linear_1 [512,16896], linear_2..5 [512,512]  simple_model.py:35-44 */
    const float* linear_b[5]; /* [512] each */
    const float* out_w;       /* output_layer_type [3,512] simple_model.py:46 */
    const float* out_b;       /* [3] */
} pv_weights_p1;

typedef struct pv_weights_p2 {
    pv_rnn_dir encoder[2]; /* GRU(10 -> 128)  pepper/modules/python/models/simple_model.py:12-16 */
    pv_rnn_dir decoder[2]; /* GRU(256 -> 128) :17-21 */
    const float* dense_w;  /* dense1 [5,256] :24 */
    const float* dense_b;  /* [5] */
} pv_weights_p2;

int pv_rnn_load_p1(pv_ctx* ctx, const pv_weights_p1* w, int dtype);
int pv_rnn_load_p2(pv_ctx* ctx, const pv_weights_p2* w, int dtype);

/* P1: images int8 [B,33,26] -> probs float [B,3] (softmax over {hom-ref, het, hom-alt}).
 * Equivalent to TransducerGRU.forward(images.float(), train_mode=False) in eval mode. */
int pv_rnn_forward_p1(pv_ctx* ctx, const int8_t* images, int64_t B, float* probs);
int pv_rnn_forward_p1_dev(pv_ctx* ctx, const int8_t* d_images, int64_t B, float* d_probs, void* stream);
/* optional taps for parity tests (device or host per the variant called): encoder/decoder outputs
 * [B,33,512]; either may be NULL */
int pv_rnn_forward_p1_debug(pv_ctx* ctx, const int8_t* images, int64_t B, float* probs,
                            float* enc_out, float* dec_out);

/* P2: images uint8 [B,1000,10] -> labels uint8 [B,1000] (argmax of the accumulated softmax) and,
 * optionally, the accumulated softmax acc float [B,1000,5] (may be NULL). Reproduces the 19-window
 * sliding loop with hidden carry of pepper/modules/python/models/predict.py:47-97. */
int pv_rnn_forward_p2(pv_ctx* ctx, const uint8_t* images, int64_t B, uint8_t* labels, float* acc);
int pv_rnn_forward_p2_dev(pv_ctx* ctx, const uint8_t* d_images, int64_t B, uint8_t* d_labels, float* d_acc,
                          void* stream);

/* P2, one model call: TransducerGRU.forward(x, hidden) (pepper/modules/python/models/simple_model.py:27-42,
 * called once per window by predict.py:65). images uint8 [B,100,10], hidden_in float [B,2,128] (NULL = zeros)
 * -> logits float [B,100,5] (before softmax), hidden_out float [B,2,128] (may be NULL). HOST pointers. */
int pv_rnn_forward_p2_window(pv_ctx* ctx, const uint8_t* images, const float* hidden_in, int64_t B, float* logits,
                             float* hidden_out);

/* ---- multi-GPU: the one exchange step ------------------------------------------------------------------------------
 * Regions shard across ranks (interval i -> rank i % world, pepper_variant/modules/python/ImageGenerationUI.py:211) with no
 * data-path collective; pv_gather moves every rank's per-window rows (probabilities, and whatever keys the caller packs next
 * to them) to ONE rank over RCCL: an all-gather of the row counts, then grouped point-to-point sends to `dst` (xGMI is
 * point-to-point: the sends of the other ranks run on different links). The reference has no counterpart (each caller process
 * writes its own prediction file, RunInference.py:101-106; its only process-group site is
 * pepper/modules/python/models/predict_distributed_gpu.py:124-129). RCCL (librccl.so.1) is opened with dlopen on first use.
 * One communicator per context; the 128-byte id is made by one rank (pv_comm_unique_id) and handed to the others by the
 * launcher (environment, file, TCP store, torch.distributed broadcast). */
typedef struct pv_comm pv_comm;
#define PV_COMM_ID_BYTES 128
int pv_comm_unique_id(pv_ctx* ctx, char* id128);
int pv_comm_create(pv_ctx* ctx, const char* id128, int rank, int world, pv_comm** out);
void pv_comm_destroy(pv_comm* comm);
/* Every rank passes its n_rows rows of row_bytes bytes (DEVICE memory). counts_out [world] (host, optional) receives the
 * per-rank row counts on every rank when the call returns; on `dst` the rows arrive rank-major in d_recv (DEVICE, capacity
 * recv_capacity_rows rows) asynchronously on `stream`; other ranks may pass NULL for d_recv.
 * The capacity test is COLLECTIVE: the counts are all-gathered together with the destination's capacity, so when the rows do
 * not fit EVERY rank returns PV_ERR_CAPACITY (counts_out filled) before any send or receive is posted - no rank is left
 * waiting for a partner that gave up. A destination that passes NULL for d_recv announces capacity 0. */
int pv_gather(pv_ctx* ctx, pv_comm* comm, const void* d_send, int64_t n_rows, int row_bytes, void* d_recv,
              int64_t recv_capacity_rows, int64_t* counts_out, int dst, void* stream);

/* The count exchange alone (one all-gather of an int64 per rank): lets the destination size its receive buffer for ragged
 * ranks before pv_gather. counts_out [world] on every rank. */
int pv_gather_counts(pv_ctx* ctx, pv_comm* comm, int64_t n_rows, int64_t* counts_out, void* stream);

/* Diagnostic (tests, tuning): C = A . W^T + bias through the 3-term split-bf16 MFMA GEMM of PV_DTYPE_BF16_INPUT_GEMM alone.
 * HOST pointers, fp32 row-major A [M,K], W [N,K], bias [N] or NULL; C [splits][M][N] row-major (quads = 0) or [M/4][N][4]
 * (quads = 1: four consecutive rows of a column adjacent, splits = 1). M % 4 == 0, N % 256 == 0, K % (32 * splits) == 0.
 * *ms (optional) receives the kernel's duration. Has no counterpart in the reference. */
int pv_debug_gemm_bf16x3(pv_ctx* ctx, const float* A, const float* W, const float* bias, int64_t M, int N, int K,
                         int splits, int quads, float* C, float* ms);

/* Per-kernel timing for the benchmark's roofline leg: between pv_profile_begin and pv_profile_end every
 * kernel the context launches is bracketed by HIP events on its launch stream. pv_profile_end
 * synchronises the device and returns the number of distinct kernels; names_buf receives their names
 * ('\n'-separated), ms_sum[i] / counts[i] the summed duration and launch count of kernel i. */
int pv_profile_begin(pv_ctx* ctx);
/* The same, for the kernels whose profile name starts with `prefix` only (NULL or "": all). Two events per launch put a few
 * microseconds between kernels, so a timed region that needs one kernel's launch durations brackets that kernel alone. */
int pv_profile_begin_only(pv_ctx* ctx, const char* prefix);
int pv_profile_end(pv_ctx* ctx, char* names_buf, int buf_len, float* ms_sum, int* counts, int max_kernels);

/* Small batches run in "split" kernel forms whose workgroups swap hidden state every time step (pv_rnn_forward_p1* up to
 * 1024 windows, pv_rnn_forward_p2* up to 2048 chunks); a launch needs all its workgroups resident at once, which holds
 * whenever it is chosen, except on a GPU that other work keeps busy for long stretches: a poll then gives up after a bounded
 * wait. A call in which that happened NEVER returns numbers that look like results: its last kernel overwrites the outputs
 * (P1 probabilities NaN; P2 labels 255, accumulated softmax / logits / hidden state NaN), and so does every later call on the
 * context until the host acknowledges the condition. The host-buffer entry points check for it themselves (PV_ERR_STATE, and
 * acknowledge); callers of the asynchronous *_dev forms call this at their own synchronisation points: it synchronises the
 * context's stream, acknowledges, and returns the number of polls that gave up since the last call (0 = every result is
 * good), or a negative PV_ERR_* code. A context that shares its GPU with other work sets option shared_device = 1. */
int pv_rnn_exchange_timeouts(pv_ctx* ctx);

/* Kernel-form options of a context. They replace process-environment lookups at call time: the environment only supplies
 * DEFAULTS, read once in pv_create (variable in brackets); a forward call reads the context's options and nothing else.
 *   lstm_split    [PV_LSTM_SPLIT]   1 (default) / 0: allow / never use the unit-split LSTM form (<= 1024 windows per call)
 *   lstm_rows     [PV_LSTM_ROWS]    0 auto / 16 / 32: tile form of the one-workgroup LSTM kernel (explicit: no unit split)
 *   tail_rows     [PV_TAIL_ROWS]    0 auto / 16 / 32;   head_splits [PV_HEAD_SPLITS] 0 auto / 1 / 3 / 11 / 33;   head_map [PV_HEAD_MAP] 1 / 0
 *   gru_rows      [PV_GRU_ROWS]     0 auto / 16 / 32
 *   gru_split     [PV_GRU_SPLIT]    1 / 0: allow the split GRU forms at all;   gru_usplit [PV_GRU_USPLIT] 1 / 0: the unit-split one
 *   p1_bf16_min_batch                 P1 in the PV_DTYPE_BF16_INPUT_GEMM mode: calls with fewer windows than this (default 513) run the
 *                                   fp32 kernels, which are faster there; 0 = always the bf16x3 kernels
 *   shared_device [PV_SHARED_DEVICE] 0 / 1: other streams or processes keep this GPU busy (e.g. several un-fused callers per
 *                                   GPU, RunInferenceArguments.py:67-74): never choose a form that needs co-resident workgroups
 *   exchange_spin_log2              2..22 (default 18): bounded polls give up after 2^n tries
 *   debug_drop_part                 -1 (off) / 0..3: diagnostic, one part of every unit-split group never runs (tests force a
 *                                   time-out with it and see the poison)
 * Unknown names and values outside these sets return PV_ERR_INVALID. */
int pv_set_option(pv_ctx* ctx, const char* name, int value);
int pv_get_option(pv_ctx* ctx, const char* name, int* value);

/* ---- hipGraph capture of a launch sequence ------------------------------------------------------------------------
 * Everything the *_dev entry points do is stream work with device-resident state (no host read-back, tags / counters of
 * the split forms kept on the device), so a sequence of them can be captured once and replayed:
 *     run the calls once (sizes the workspace)          pv_summarize_regions_dev(...); pv_rnn_forward_p1_dev(...);
 *     pv_graph_begin(ctx, stream);                       same calls, same pointers: recorded, not run
 *     pv_graph_end(ctx, &graph);
 *     per batch: refill the SAME input buffers, then     pv_graph_launch(graph, stream);
 * `stream` must be a created stream (NULL = the context's own; the legacy null stream cannot capture) and the one the calls
 * in between are given. A call that would have to grow the workspace inside a capture fails with PV_ERR_STATE. The
 * reference has no counterpart (its loop is eager PyTorch); BASELINE configs[4] names the technique. */
typedef struct pv_graph pv_graph;
int pv_graph_begin(pv_ctx* ctx, void* stream);
int pv_graph_end(pv_ctx* ctx, pv_graph** graph);
int pv_graph_launch(pv_graph* graph, void* stream);
void pv_graph_destroy(pv_graph* graph);

/* bytes of device workspace the context currently holds (diagnostics) */
int64_t pv_workspace_bytes(pv_ctx* ctx);
/* library/ABI version: major*10000 + minor*100 + patch */
int pv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PEPPER_HIP_H */
