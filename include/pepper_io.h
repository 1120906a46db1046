/*
 * pepper_io.h — C-ABI of the native BAM/BAI + FASTA/FAI readers that feed the image builder
 * (SURVEY 8f-1). Replaces the pybind11 classes PEPPER_VARIANT.BAM_handler / FASTA_handler
 *   reference: pepper_variant/modules/cpp/pybind_api.h:224-235 (bindings),
 *              pepper_variant/modules/cpp/bam_handler.cpp:115-451 (get_reads, region clipping),
 *              pepper_variant/modules/cpp/fasta_handler.cpp:18-56.
 * CPU-side library (libpepper_io.so, links zlib only); buffers returned through pvio_reads are owned
 * by the handle and stay valid until the next pvio_bam_get_reads / pvio_bam_close on it.
 */
#ifndef PEPPER_IO_H
#define PEPPER_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pv_bam pv_bam;
typedef struct pv_fasta pv_fasta;

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
int pvio_bam_nref(pv_bam* bam);
const char* pvio_bam_ref_name(pv_bam* bam, int i);
int64_t pvio_bam_ref_len(pv_bam* bam, int i);
/* BAM_handler::get_reads(chromosome, start, stop, include_supplementary, min_mapq, min_baseq); 0 on success */
int pvio_bam_get_reads(pv_bam* bam, const char* contig, int64_t start, int64_t stop, int include_supplementary,
                       int min_mapq, int min_baseq, pvio_reads* out);

pv_fasta* pvio_fasta_open(const char* path); /* needs <path>.fai */
void pvio_fasta_close(pv_fasta* fa);
int pvio_fasta_nseq(pv_fasta* fa);
const char* pvio_fasta_name(pv_fasta* fa, int i);
int64_t pvio_fasta_len(pv_fasta* fa, const char* contig); /* -2 if the contig is unknown */
/* FASTA_handler::get_reference_sequence(contig, start, stop): upper-cased bases [start, stop-1] clamped to the
 * sequence, written to out (capacity stop-start); returns the number of bases, -2 unknown contig, -1 error */
int64_t pvio_fasta_fetch(pv_fasta* fa, const char* contig, int64_t start, int64_t stop, char* out);

#ifdef __cplusplus
}
#endif
#endif
