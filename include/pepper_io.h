/*
 * pepper_io.h — C-ABI of the native BAM/BAI + FASTA/FAI readers that feed the image builder
 * (SURVEY 8f-1). Replaces the pybind11 classes PEPPER_VARIANT.BAM_handler / FASTA_handler
 *   reference: pepper_variant/modules/cpp/pybind_api.h:224-235 (bindings),
 *              pepper_variant/modules/cpp/bam_handler.cpp:115-451 (get_reads, region clipping),
 *              pepper_variant/modules/cpp/fasta_handler.cpp:18-56.
 *              pepper_variant/modules/python/AlignmentSummarizer.py:180-218 (per-interval fetch + reservoir sampling),
 *              pepper_variant/modules/python/VcfWriter.py:21-46 (bgzip + tabix outputs through pysam).
 * CPU-side library (libpepper_io.so: zlib, and libdeflate.so.0 through dlopen when the host has it); buffers returned through pvio_reads are owned
 * by the handle and stay valid until the next pvio_bam_get_reads / pvio_bam_close on it; a pvio_batch owns its
 * arrays until pvio_batch_free. Handles are not thread-safe: one (pv_bam, pv_fasta) pair per reader thread;
 * the calls themselves hold no global state (errors are thread-local) and may run concurrently on distinct handles.
 * Every size read from a file is validated before use and every BGZF block is checked against its CRC32 trailer: corrupt or
 * truncated inputs fail with a message - also in the middle of a region query (only a clean end of file ends one quietly).
 */
#ifndef PEPPER_IO_H
#define PEPPER_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pv_bam pv_bam;
typedef struct pv_fasta pv_fasta;

/* BGZF blocks are inflated with libdeflate when libdeflate.so.0 can be opened (PEPPER_INFLATE=zlib forces zlib), else with
 * zlib: byte-identical results, about twice the rate. pvio_inflate_backend names the one in use ("libdeflate" / "zlib");
 * pvio_set_inflate_backend(0 / 1) switches at run time (tests) and returns 1 when libdeflate is then in use. */
const char* pvio_inflate_backend(void);
int pvio_set_inflate_backend(int use_libdeflate);

/* the reads of ONE region after the reference's clipping, in the flat layout of pv_batch_in */
typedef struct pvio_reads {
    int64_t n_reads, n_bases, n_cigar;
    const int64_t* pos;        /* type_read::pos (first kept reference position) */
    const int64_t* pos_end;    /* type_read::pos_end */
    const uint16_t* flag;      /* raw BAM flag */
    const uint8_t* is_reverse; /* flags.is_reverse */
    const uint8_t* mapq;
    const int32_t* hp_tag;     /* HP aux tag or 0 */
    const int64_t* base_off;   /* [n_reads+1] */
    const uint8_t* bases;      /* upper-case IUPAC symbols */
    const uint8_t* quals;      /* raw phred */
    const int64_t* cigar_off;  /* [n_reads+1] */
    const uint32_t* cigar;     /* (len << 4) | op, op codes of the ORIGINAL ops */
    const int64_t* name_off;   /* [n_reads+1] */
    const char* names;         /* query names, concatenated */
} pvio_reads;

const char* pvio_last_error(void);

pv_bam* pvio_bam_open(const char* path); /* needs <path>.bai or <stem>.bai; NULL on failure */
void pvio_bam_close(pv_bam* bam);
/* n helper threads of this handle inflate BGZF blocks AHEAD of the thread that reads records from it (what hts_set_threads
 * does for htslib's bam_handler); 0 (the default) = none, nothing is read ahead. Read-ahead stops at the end of the index
 * chunk being walked. Returns the number of helpers now running, -1 on a null handle. Call it between queries only. */
int pvio_bam_set_threads(pv_bam* bam, int n_helpers);
int pvio_bam_nref(pv_bam* bam);
const char* pvio_bam_ref_name(pv_bam* bam, int i);
int64_t pvio_bam_ref_len(pv_bam* bam, int i);
/* BAM_handler::get_reads(chromosome, start, stop, include_supplementary, min_mapq, min_baseq); 0 on success */
int pvio_bam_get_reads(pv_bam* bam, const char* contig, int64_t start, int64_t stop, int include_supplementary,
                       int min_mapq, int min_baseq, pvio_reads* out);

/* A batch of intervals read straight into the flat SoA layout of pv_batch_in (include/pepper_hip.h): for interval i the
 * region [max(0, start_i - safe_bases), end_i + safe_bases] is fetched with get_reads' clipping, reservoir-sampled when it
 * holds more than min(max_reads, downsample_rate * n) reads (NumPy legacy RandomState(seed).randint stream, fresh per
 * interval), its reference bases fetched (clamped at the contig end) and its candidate range set to [start_i, min(end_i,
 * region end)] — AlignmentSummarizer.py:180-218. Intervals without reads are left out ("no group when no reads");
 * interval_index[g] names the input interval of batch region g. */
typedef struct pvio_batch {
    void* owner;
    int32_t n_regions;
    int32_t reserved;
    int64_t n_reads, n_bases, n_cigar, n_ref_bytes, max_region_len;
    const int64_t *ref_start, *ref_end, *cand_start, *cand_end, *ref_off; /* as in pv_batch_in */
    const uint8_t* ref;
    const int64_t *read_off, *read_pos;
    const uint8_t *read_flags, *read_mapq;
    const int64_t* base_off;
    const uint8_t *bases, *quals;
    const int64_t* cigar_off;
    const uint32_t* cigar;
    const int64_t* interval_index; /* [n_regions] */
    const int64_t* reads_seen;     /* [n_regions] reads before down-sampling */
    double t_inflate;              /* stage timers (seconds): BGZF read + inflate ... */
    double t_total;                /* ... and the whole call (record decode + clip + FASTA = t_total - t_inflate) */
    int64_t bytes_inflated;
    const int32_t* read_hp;        /* [n_reads] HP aux tag or 0: the `read_hp` argument of pv_summarize_regions_hp */
    double t_helpers;              /* seconds the handle's helper threads spent inflating during the call (pvio_bam_set_threads) */
} pvio_batch;
int pvio_fill_batch(pv_bam* bam, pv_fasta* fa, int n_intervals, const char* const* contigs, const int64_t* starts,
                    const int64_t* ends, int safe_bases, int include_supplementary, int min_mapq, double downsample_rate,
                    int64_t max_reads, uint32_t seed, pvio_batch** out);
void pvio_batch_free(pvio_batch* batch);
/* kept read indices, in output order, of the reservoir sampling above; returns their number (out holds n_reads slots) */
int64_t pvio_reservoir_indices(int64_t n_reads, double downsample_rate, int64_t max_reads, uint32_t seed, int64_t* out);

pv_fasta* pvio_fasta_open(const char* path); /* needs <path>.fai */
void pvio_fasta_close(pv_fasta* fa);
int pvio_fasta_nseq(pv_fasta* fa);
const char* pvio_fasta_name(pv_fasta* fa, int i);
int64_t pvio_fasta_len(pv_fasta* fa, const char* contig); /* -2 if the contig is unknown */
/* FASTA_handler::get_reference_sequence(contig, start, stop): upper-cased bases [start, stop-1] clamped to the
 * sequence, written to out (capacity stop-start); returns the number of bases, -2 unknown contig, -1 error */
int64_t pvio_fasta_fetch(pv_fasta* fa, const char* contig, int64_t start, int64_t stop, char* out);

/* ---- writers -------------------------------------------------------------------------------------------------- */
/* coordinate-sorted BAM + <path>.bai from flat arrays (reads sorted by (tid, pos); read_flags bit0 = reverse strand):
 * synthetic inputs for the file-path benchmark and tests */
int pvio_write_bam(const char* path, int n_ref, const char* const* ref_names, const int64_t* ref_lens, int64_t n_reads,
                   const int32_t* read_tid, const int64_t* read_pos, const uint8_t* read_flags, const uint8_t* read_mapq,
                   const int64_t* base_off, const uint8_t* bases, const uint8_t* quals, const int64_t* cigar_off,
                   const uint32_t* cigar, int level);
/* bgzip `text` (a whole VCF: '#' header lines, then records sorted by contig and position) to `path` and write the tabix
 * index `path`.tbi — what pysam.VariantFile(*.vcf.gz, 'w') + pysam.tabix_index produce (VcfWriter.py:21-46) */
int pvio_write_vcf_gz(const char* path, const char* text, int64_t n_bytes);
/* inflate a whole BGZF file; out == NULL only counts; returns bytes or -1 */
int64_t pvio_bgzf_read_all(const char* path, char* out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif
