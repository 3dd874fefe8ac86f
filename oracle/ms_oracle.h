/*
 * ms_oracle.h -- CPU ORACLE for the MuCHSALSA overlap hot path (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * This is a plain-C restatement of the reference algorithm for the path
 *   PAF rows -> VertexMatch store -> MatchMap::calculateEdges -> chainingAndOverlaps
 * (reference: libms/src/BlastFileReader.cpp:72-130, libms/src/matching/MatchMap.cpp:52-224,
 *  libms/src/kernel/mpp.cpp:38-305, libms/src/kernel/ol.cpp:31-101, src/main.cpp:328-414).
 *
 * PARITY UNPINNED: the reference's own tests hold no fixture for this path and the
 * reference cannot be built in this image without writing a stand-in for the un-vendored microsoft/GSL v3.0.1
 * headers (libms/CMakeLists.txt:5-14), which the build rules forbid.  The oracle is therefore pinned only by
 * (a) hand-derived edge-case vectors in tests/golden/ and (b) an independent pure-Python restatement
 * (oracle/ms_oracle_py.py) that must agree with it bit for bit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (muchsalsa_amd/, include/msgpu.h) never links, imports or calls it.
 *
 * Record layouts are byte-identical to include/msgpu.h so tests can compare tables directly.
 */
#ifndef MS_ORACLE_H
#define MS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One ACCEPTED PAF line (output of BlastFileReader::parseLine, BlastFileReader.cpp:101-126). 40 bytes. */
typedef struct ms_row {
  uint32_t anchor_id; /* illumina (unitig) registry id, first-seen order (Registry.cpp:36-45)           */
  uint32_t read_id;   /* nanopore registry id, first-seen order; registered before the anchor id        */
  int32_t  read_len;  /* PAF col 6 (std::stoi)                                                           */
  int32_t  i_lo, i_hi; /* illuminaRange  = (col2, col3-1), inclusive                                     */
  int32_t  n_lo, n_hi; /* nanoporeRange  = (col7, col8-1), inclusive                                     */
  uint32_t score;     /* col 9, number of matches                                                        */
  uint32_t line;      /* 0-based line index in the PAF file                                              */
  uint32_t flags;     /* bit0: direction (col4 == "+"), bit1: isPrimary                                  */
} ms_row;

#define MS_ROW_DIR 1u
#define MS_ROW_PRIMARY 2u

/* Graph edge between two reads. 32 bytes. */
typedef struct ms_edge {
  uint32_t v1, v2;      /* read ids; v1 is the read whose first accepted line is lower (MatchMap.cpp:204-213) */
  uint64_t em_off;      /* first EdgeMatch of this edge in the EdgeMatch table                             */
  uint64_t order_off;   /* first EdgeOrder of this edge in the order table                                 */
  uint32_t em_cnt;
  uint16_t order_cnt;
  uint8_t  shadow;      /* Edge::isShadow after chainingAndOverlaps (main.cpp:389-395)                     */
  uint8_t  pad;
} ms_edge;

/* EdgeMatch (MatchMap.h:68-74) + its keys. 32 bytes. */
typedef struct ms_edgematch {
  int32_t  ov_lo, ov_hi; /* overlap on the anchor (inclusive)                                              */
  double   score;        /* sumScore (MatchMap.cpp:200-202)                                                */
  uint32_t anchor_id;
  uint32_t line;         /* outerMatch->lineNumber (MatchMap.cpp:218)                                      */
  uint32_t flags;        /* bit0: direction, bit1: isPrimary                                               */
  uint32_t edge_idx;
} ms_edgematch;

/* EdgeOrder (Edge.h:49-60). 64 bytes. */
typedef struct ms_order {
  uint32_t edge_idx;
  uint32_t flags;        /* bit0: start==v1 (else start==v2,end==v1), bit1: isContained, bit2: direction, bit3: isPrimary */
  double   left_offset;
  double   right_offset;
  uint64_t score;        /* path score truncated to size_t (mpp.cpp:34,221,244)                            */
  uint64_t ids_off;      /* into the ids pool                                                              */
  uint32_t ids_cnt;
  uint32_t start, end, base; /* read ids; base is always v1 (ol.cpp:81-96)                                 */
  uint32_t pad[2];
} ms_order;

#define MS_ORD_START_V1 1u
#define MS_ORD_CONTAINED 2u
#define MS_ORD_DIR 4u
#define MS_ORD_PRIMARY 8u

typedef struct ms_params {
  uint32_t min_matches; /* 400  BlastFileReader.cpp:48 */
  uint32_t th_length;   /* 500  BlastFileReader.cpp:49 */
  uint32_t th_matches;  /* 500  BlastFileReader.cpp:50 */
  uint32_t th_overlap;  /* 100  MatchMap.cpp:41        */
  uint64_t wiggle_room; /* 300  Application.h:132      */
  double   ratio_pct;   /* 15   mpp.cpp:136            */
  double   alt_frac;    /* 0.75 mpp.cpp:223            */
} ms_params;

typedef struct ms_rows {
  ms_row  *rows;
  size_t   n_rows;       /* accepted rows, in line order */
  size_t   n_lines;      /* lines indexed in the file (the last one is never parsed) */
  uint32_t n_reads, n_anchors;
  char    *read_names;   /* NUL-separated, registry order */
  char    *anchor_names;
  size_t   read_names_len, anchor_names_len;
} ms_rows;

typedef struct ms_tables {
  ms_edge      *edges;   size_t n_edges;   /* sorted by (v1, v2)                          */
  ms_edgematch *ems;     size_t n_ems;     /* grouped by edge, then vStart order (mpp.cpp:164-172) */
  ms_order     *orders;  size_t n_orders;  /* grouped by edge, reference emission order   */
  uint32_t     *ids;     size_t n_ids;
  /* per-read facts (Vertex): indexed by read id, n_reads = max id + 1 */
  int32_t      *read_len; uint32_t *read_first_line; uint32_t n_reads;
  /* counters (SURVEY.md section 8 notation) */
  uint64_t rows_alive;   /* rows after the (read,anchor) lowest-line rule                                  */
  uint64_t n_anchors;    /* anchors with >= 1 alive row                                                    */
  uint64_t p_eval;       /* scaffold pairs tested                                                          */
  uint64_t compat_checks;/* checkCompatibility calls                                                       */
  uint64_t shadow_edges;
} ms_tables;

void ms_oracle_default_params(ms_params *p);

/* A1: index lines, parse all but the last, filter, assign registry ids. Returns 0 or a negative error. */
int ms_oracle_parse_paf(const char *path, const ms_params *p, ms_rows *out);
void ms_oracle_free_rows(ms_rows *r);

/* A1 tail + A2..A7 on accepted rows (any order; duplicates allowed). Returns 0 or a negative error. */
int ms_oracle_overlap(const ms_row *rows, size_t n_rows, const ms_params *p, ms_tables *out);
void ms_oracle_free_tables(ms_tables *t);

const char *ms_oracle_strerror(int code);

/* ---- sequence side of the "consensus" stage (data-parallel half) ------------------------
 * PINNED by the reference's own fixtures: libms/tests/SA_test.cpp:11-136 (whole-record fetch of test_data/fasta.fa
 * and test_data/fastq.fq), copied as data into tests/golden/. */

typedef struct ms_seqs {
  uint32_t  n;        /* records, in file order (duplicate names keep the first: unordered_map::emplace)          */
  char     *names;    /* NUL-separated cleaned ids (cut at the first whitespace, SequenceAccessor.cpp:82-87)      */
  size_t    names_len;
  char     *bases;    /* concatenated sequences, whitespace stripped (SequenceAccessor.cpp:54-69)                 */
  uint64_t *off;      /* n + 1 offsets into bases                                                                 */
} ms_seqs;

/* SequenceAccessor::_buildNanoporeIdx / _buildIlluminaIdx + getSequenceFromFile (SequenceAccessor.cpp:54-69,143-231).
 * is_fastq < 0: decide from the file extension like isFastQ (SequenceAccessor.cpp:71-80). */
int  ms_oracle_seq_load(const char *path, int is_fastq, ms_seqs *out);
void ms_oracle_seq_free(ms_seqs *s);

/* strSlice (SequenceUtils.cpp:27-38): Python-like indices, INCLUSIVE end, clipped.  Returns start, writes length. */
size_t ms_oracle_str_slice(size_t size, int start, int end, size_t *len);
/* getReverseComplement (SequenceUtils.cpp:41-61): A<->T, C<->G, everything else unchanged. */
void ms_oracle_revcomp(const char *in, size_t n, char *out);
/* getIlluminaSequence / getNanoporeSequence (SequenceUtils.cpp:63-85) = strSlice(seq, l, r + 1), reverse complement
 * when !direction.  out must hold ms_oracle_get_sequence_len bytes; returns the length. */
size_t ms_oracle_get_sequence(const char *seq, size_t size, int left, int right, int direction, char *out);

/* ---- segment builders of assemblePath (ap.cpp:191-203, 352-579): which pieces of which sequence, in which
 * orientation, make up an anchor / the flank left or right of an anchor / the stretch between two anchors.
 * `m*` = the VertexMatch of the read on the anchor(s); nano / illu* = the whole sequences; ov* = the anchor's (cluster)
 * overlap from Id2OverlapMap.  `out` must hold nano_len + illu lengths + 8 bytes.  Returns the length written. */
size_t ms_oracle_anchor_sequence(const ms_row *m, const char *illu, size_t illu_len, int ov_lo, int ov_hi,
                                 int direction, char *out);                                    /* ap.cpp:424-433 */
size_t ms_oracle_left_of_anchor(const ms_row *m, const char *nano, size_t nano_len, const char *illu, size_t illu_len,
                                size_t nanopore_length, int ov_lo, int ov_hi, int direction, char *out); /* :352-386 */
size_t ms_oracle_right_of_anchor(const ms_row *m, const char *nano, size_t nano_len, const char *illu, size_t illu_len,
                                 size_t nanopore_length, int ov_lo, int ov_hi, int direction, char *out); /* :388-422 */
/* getSequenceBetweenAnchors (ap.cpp:435-579): returns 1 and the sequence, or 0 (std::nullopt) when the corrected
 * ranges overlap; *distance = std::get<0> of the result. */
int ms_oracle_between_anchors(const ms_row *ml, const ms_row *mr, const char *nano, size_t nano_len,
                              const char *illu_l, size_t illu_l_len, const char *illu_r, size_t illu_r_len, int ovl_lo,
                              int ovl_hi, int ovr_lo, int ovr_hi, int direction, int *distance, char *out,
                              size_t *out_len);

/* ---- graph clean-up, first step: findContractionEdges (src/main.cpp:416-463) with
 * sanityCheck (libms/src/kernel/sc.cpp:29-90) over the result tables.  out[e] (n_edges entries) = index in the order
 * table of the first contained & primary EdgeOrder of edge e that is sane against every non-shadow neighbour of its
 * start vertex, or -1.  Defined in ms_oracle_graph.c. */
int ms_oracle_find_contraction_edges(const ms_tables *t, uint64_t wiggle, int64_t *out);

/* Levenshtein distance (unit costs, global alignment), full O(n*m) DP -- the checker of the banded GPU kernel
 * (the reference has no counterpart).  Returns min(distance, band + 1). */
uint32_t ms_oracle_edit_distance(const char *a, size_t n, const char *b, size_t m, uint32_t band);
/* the same restricted to |i - j| <= band: exact when the distance is <= band, O(n * band) */
uint32_t ms_oracle_edit_distance_banded(const char *a, size_t n, const char *b, size_t m, uint32_t band);

#ifdef __cplusplus
}
#endif
#endif
