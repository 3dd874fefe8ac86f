/*
 * ms_oracle.c -- CPU ORACLE (test infrastructure, NOT product; see ms_oracle.h for the rules of use).
 *
 * Plain-C restatement of the reference overlap path.  Every function cites the reference lines it follows
 * (paths relative to the reference tree).  Single-thread reference semantics.
 * PARITY UNPINNED by the reference's own tests (it holds no fixture for this path); see ms_oracle.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (IEEE fp64, no FMA contraction, matching the reference's
 * plain x86-64 build) -- see oracle/Makefile.
 */
#include "ms_oracle.h"

#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
  MS_OK          = 0,
  MS_E_IO        = -1, /* "Can't open blast file."  BlastFileAccessor.cpp:43-45 */
  MS_E_FORMAT    = -2, /* "Invalid BLAST file."     BlastFileReader.cpp:97-99   */
  MS_E_NUMBER    = -3, /* std::stoi would throw                                  */
  MS_E_NOMEM     = -4,
  MS_E_ARG       = -5
};

const char *ms_oracle_strerror(int code) {
  switch (code) {
  case MS_OK: return "ok";
  case MS_E_IO: return "can't open blast file";
  case MS_E_FORMAT: return "invalid BLAST file (fewer than 10 tab-separated columns)";
  case MS_E_NUMBER: return "invalid integer field";
  case MS_E_NOMEM: return "out of memory";
  case MS_E_ARG: return "invalid argument";
  default: return "unknown error";
  }
}

void ms_oracle_default_params(ms_params *p) {
  p->min_matches = 400; /* BlastFileReader.cpp:48 */
  p->th_length   = 500; /* BlastFileReader.cpp:49 */
  p->th_matches  = 500; /* BlastFileReader.cpp:50 */
  p->th_overlap  = 100; /* MatchMap.cpp:41        */
  p->wiggle_room = 300; /* Application.h:132      */
  p->ratio_pct   = 15;  /* mpp.cpp:136            */
  p->alt_frac    = 0.75; /* mpp.cpp:223           */
}

/* ------------------------------------------------------------------------------------------------------------------
 * small containers
 * ---------------------------------------------------------------------------------------------------------------- */

static void *xrealloc(void *p, size_t n) {
  void *q = realloc(p, n ? n : 1);
  if (!q) {
    fprintf(stderr, "ms_oracle: out of memory\n");
    abort();
  }
  return q;
}

/* open-addressing map u64 -> u32 */
typedef struct {
  uint64_t *keys;
  uint32_t *vals;
  size_t    cap, n;
} map64;

#define MAP_EMPTY UINT64_MAX

static uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

static void map_init(map64 *m, size_t expect) {
  size_t cap = 16;
  while (cap < expect * 2) cap <<= 1;
  m->cap  = cap;
  m->n    = 0;
  m->keys = (uint64_t *)xrealloc(NULL, cap * sizeof(uint64_t));
  m->vals = (uint32_t *)xrealloc(NULL, cap * sizeof(uint32_t));
  for (size_t i = 0; i < cap; ++i) m->keys[i] = MAP_EMPTY;
}
static void map_free(map64 *m) {
  free(m->keys);
  free(m->vals);
  m->keys = NULL;
  m->vals = NULL;
}
static uint32_t *map_find(const map64 *m, uint64_t k) {
  size_t i = mix64(k) & (m->cap - 1);
  while (m->keys[i] != MAP_EMPTY) {
    if (m->keys[i] == k) return &m->vals[i];
    i = (i + 1) & (m->cap - 1);
  }
  return NULL;
}
static void map_grow(map64 *m);
static uint32_t *map_insert(map64 *m, uint64_t k, uint32_t v, int *inserted) {
  if ((m->n + 1) * 2 > m->cap) map_grow(m);
  size_t i = mix64(k) & (m->cap - 1);
  while (m->keys[i] != MAP_EMPTY) {
    if (m->keys[i] == k) {
      if (inserted) *inserted = 0;
      return &m->vals[i];
    }
    i = (i + 1) & (m->cap - 1);
  }
  m->keys[i] = k;
  m->vals[i] = v;
  m->n++;
  if (inserted) *inserted = 1;
  return &m->vals[i];
}
static void map_grow(map64 *m) {
  map64 o = *m;
  m->cap  = o.cap * 2;
  m->n    = 0;
  m->keys = (uint64_t *)xrealloc(NULL, m->cap * sizeof(uint64_t));
  m->vals = (uint32_t *)xrealloc(NULL, m->cap * sizeof(uint32_t));
  for (size_t i = 0; i < m->cap; ++i) m->keys[i] = MAP_EMPTY;
  for (size_t i = 0; i < o.cap; ++i)
    if (o.keys[i] != MAP_EMPTY) map_insert(m, o.keys[i], o.vals[i], NULL);
  free(o.keys);
  free(o.vals);
}

/* string registry: name -> dense id in first-seen order (Registry.cpp:36-45) */
typedef struct {
  char     *blob;
  size_t    blob_len, blob_cap;
  size_t   *offs; /* offset of name i in blob */
  uint32_t  n, offs_cap;
  uint32_t *slots; /* id+1, 0 = empty */
  size_t    cap;
} registry;

static uint64_t fnv1a(const char *s, size_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < n; ++i) {
    h ^= (unsigned char)s[i];
    h *= 1099511628211ULL;
  }
  return h;
}
static void reg_init(registry *r) {
  memset(r, 0, sizeof(*r));
  r->cap   = 1024;
  r->slots = (uint32_t *)calloc(r->cap, sizeof(uint32_t));
}
static void reg_free(registry *r) {
  free(r->blob);
  free(r->offs);
  free(r->slots);
}
static uint32_t reg_get(registry *r, const char *s, size_t n) {
  if ((size_t)(r->n + 1) * 2 > r->cap) {
    size_t    ncap = r->cap * 2;
    uint32_t *ns   = (uint32_t *)calloc(ncap, sizeof(uint32_t));
    for (uint32_t id = 0; id < r->n; ++id) {
      const char *nm = r->blob + r->offs[id];
      size_t      i  = fnv1a(nm, strlen(nm)) & (ncap - 1);
      while (ns[i]) i = (i + 1) & (ncap - 1);
      ns[i] = id + 1;
    }
    free(r->slots);
    r->slots = ns;
    r->cap   = ncap;
  }
  size_t i = fnv1a(s, n) & (r->cap - 1);
  while (r->slots[i]) {
    const char *nm = r->blob + r->offs[r->slots[i] - 1];
    if (strlen(nm) == n && memcmp(nm, s, n) == 0) return r->slots[i] - 1;
    i = (i + 1) & (r->cap - 1);
  }
  if (r->blob_len + n + 1 > r->blob_cap) {
    r->blob_cap = (r->blob_len + n + 1) * 2;
    r->blob     = (char *)xrealloc(r->blob, r->blob_cap);
  }
  if (r->n + 1 > r->offs_cap) {
    r->offs_cap = r->offs_cap ? r->offs_cap * 2 : 1024;
    r->offs     = (size_t *)xrealloc(r->offs, r->offs_cap * sizeof(size_t));
  }
  r->offs[r->n] = r->blob_len;
  memcpy(r->blob + r->blob_len, s, n);
  r->blob[r->blob_len + n] = '\0';
  r->blob_len += n + 1;
  r->slots[i] = r->n + 1;
  return r->n++;
}

/* ------------------------------------------------------------------------------------------------------------------
 * A1: PAF parse (BlastFileAccessor.cpp:77-91 line index; BlastFileReader.cpp:72-130 parse + filter)
 * ---------------------------------------------------------------------------------------------------------------- */

/* std::stoi semantics: skip whitespace, optional sign, >= 1 digit, must fit int. */
static int stoi_like(const char *s, size_t n, int *out) {
  size_t i = 0;
  while (i < n && (s[i] == ' ' || (s[i] >= '\t' && s[i] <= '\r'))) ++i;
  int neg = 0;
  if (i < n && (s[i] == '+' || s[i] == '-')) {
    neg = s[i] == '-';
    ++i;
  }
  if (i >= n || s[i] < '0' || s[i] > '9') return MS_E_NUMBER;
  long long v = 0;
  while (i < n && s[i] >= '0' && s[i] <= '9') {
    v = v * 10 + (s[i] - '0');
    if (v > (long long)INT_MAX + 1) return MS_E_NUMBER;
    ++i;
  }
  if (neg) v = -v;
  if (v > INT_MAX || v < INT_MIN) return MS_E_NUMBER;
  *out = (int)v;
  return MS_OK;
}

int ms_oracle_parse_paf(const char *path, const ms_params *p, ms_rows *out) {
  memset(out, 0, sizeof(*out));
  FILE *f = fopen(path, "rb");
  if (!f) return MS_E_IO;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *buf = (char *)xrealloc(NULL, (size_t)sz + 1);
  if (sz > 0 && fread(buf, 1, (size_t)sz, f) != (size_t)sz) {
    fclose(f);
    free(buf);
    return MS_E_IO;
  }
  fclose(f);
  size_t len = (size_t)sz;

  /* line index: a line ends at '\n'; a non-empty unterminated tail is a line too (IO.cpp:54-97). */
  size_t  n_lines = 0, cap_lines = 1024;
  size_t *loff = (size_t *)xrealloc(NULL, cap_lines * sizeof(size_t));
  for (size_t pos = 0; pos < len;) {
    if (n_lines + 1 >= cap_lines) {
      cap_lines *= 2;
      loff = (size_t *)xrealloc(loff, cap_lines * sizeof(size_t));
    }
    loff[n_lines++] = pos;
    const char *nl = (const char *)memchr(buf + pos, '\n', len - pos);
    pos = nl ? (size_t)(nl - buf) + 1 : len;
  }
  loff[n_lines] = len;
  out->n_lines  = n_lines;

  registry rn, ri;
  reg_init(&rn);
  reg_init(&ri);
  size_t  cap_rows = 1024, n_rows = 0;
  ms_row *rows = (ms_row *)xrealloc(NULL, cap_rows * sizeof(ms_row));
  int     rc = MS_OK;

  /* BlastFileReader.cpp:76: lineIdx < getLineCount() - 1 -- the last line is never parsed.
   * (getLineCount()==0 underflows in the reference; an empty file is treated as "nothing to parse".) */
  size_t n_parse = n_lines ? n_lines - 1 : 0;
  for (size_t li = 0; li < n_parse; ++li) {
    const char *s = buf + loff[li];
    size_t      n = loff[li + 1] - loff[li];
    if (n && s[n - 1] == '\n') --n; /* getLine pops the newline (BlastFileAccessor.cpp:60-63) */
    /* tokenise on '\t' with std::getline semantics (BlastFileReader.cpp:91-95) */
    const char *tok[12];
    size_t      tlen[12];
    size_t      nt = 0, st = 0;
    for (size_t i = 0; i <= n; ++i) {
      if (i == n || s[i] == '\t') {
        if (i == n && st == n) break; /* nothing left after the last delimiter: no token */
        if (nt < 12) {
          tok[nt]  = s + st;
          tlen[nt] = i - st;
        }
        ++nt;
        st = i + 1;
      }
    }
    /* reference requires >= 9 tokens and then reads tokens[9] (UB at exactly 9); we require 10. */
    if (nt < 10) {
      rc = MS_E_FORMAT;
      break;
    }
    int irs, ire, nom, nle;
    if ((rc = stoi_like(tok[2], tlen[2], &irs)) || (rc = stoi_like(tok[3], tlen[3], &ire)) ||
        (rc = stoi_like(tok[9], tlen[9], &nom)) || (rc = stoi_like(tok[6], tlen[6], &nle)))
      break;
    if (nom < 0) { /* reference casts to size_t (huge); we refuse */
      rc = MS_E_NUMBER;
      break;
    }
    int i_lo = irs, i_hi = ire - 1;
    int add = (uint32_t)nom >= p->min_matches;           /* :106 */
    add &= (i_hi - i_lo + 1) >= (int)p->min_matches;     /* :107 */
    if (!add) continue;
    uint32_t rid = reg_get(&rn, tok[5], tlen[5]);        /* :110 nanopore first */
    uint32_t aid = reg_get(&ri, tok[0], tlen[0]);        /* :111 */
    int nrs, nre;
    if ((rc = stoi_like(tok[7], tlen[7], &nrs)) || (rc = stoi_like(tok[8], tlen[8], &nre))) break;
    int dir  = (tlen[4] == 1 && tok[4][0] == '+');       /* :117 */
    int prim = (i_hi - i_lo + 1) >= (int)p->th_length;   /* :121 */
    prim &= (uint32_t)nom >= p->th_matches;              /* :122 */
    if (n_rows == cap_rows) {
      cap_rows *= 2;
      rows = (ms_row *)xrealloc(rows, cap_rows * sizeof(ms_row));
    }
    ms_row *r   = &rows[n_rows++];
    r->anchor_id = aid;
    r->read_id   = rid;
    r->read_len  = nle;
    r->i_lo      = i_lo;
    r->i_hi      = i_hi;
    r->n_lo      = nrs;
    r->n_hi      = nre - 1;
    r->score     = (uint32_t)nom;
    r->line      = (uint32_t)li;
    r->flags     = (dir ? MS_ROW_DIR : 0u) | (prim ? MS_ROW_PRIMARY : 0u);
  }
  free(buf);
  free(loff);
  if (rc != MS_OK) {
    free(rows);
    reg_free(&rn);
    reg_free(&ri);
    return rc;
  }
  out->rows             = rows;
  out->n_rows           = n_rows;
  out->n_reads          = rn.n;
  out->n_anchors        = ri.n;
  out->read_names       = rn.blob;
  out->read_names_len   = rn.blob_len;
  out->anchor_names     = ri.blob;
  out->anchor_names_len = ri.blob_len;
  rn.blob = NULL;
  ri.blob = NULL;
  reg_free(&rn);
  reg_free(&ri);
  return MS_OK;
}

void ms_oracle_free_rows(ms_rows *r) {
  free(r->rows);
  free(r->read_names);
  free(r->anchor_names);
  memset(r, 0, sizeof(*r));
}

/* ------------------------------------------------------------------------------------------------------------------
 * MatchMap store (MatchMap.cpp:52-81) and helpers
 * ---------------------------------------------------------------------------------------------------------------- */

typedef struct {
  const ms_row *rows; /* copy sorted by line */
  size_t        n_rows;
  map64         vm;   /* (read_id<<32 | anchor_id) -> row index : m_vertexMatches */
  uint32_t      n_reads, n_anchor_ids;
  int32_t      *read_len;
  uint32_t     *read_first_line;
  uint8_t      *read_exists;
  /* per-read alive rows sorted by (n_lo, n_hi, anchor_id): mpp.cpp:259-267 sorts exactly this */
  size_t   *rd_off;
  uint32_t *rd_rows;
  const ms_params *p;
  uint64_t  compat_checks;
} store;

static inline uint64_t vmkey(uint32_t read, uint32_t anchor) { return ((uint64_t)read << 32) | anchor; }

/* MatchMap::getVertexMatch, MatchMap.cpp:83-95 */
static const ms_row *get_vertex_match(const store *s, uint32_t read, uint32_t anchor) {
  uint32_t *v = map_find(&s->vm, vmkey(read, anchor));
  return v ? &s->rows[*v] : NULL;
}

/* VertexMatch::rRatio, BlastFileReader.cpp:118-119 */
static inline double rratio(const ms_row *m) {
  return (double)(m->i_hi - m->i_lo + 1) / (double)(m->n_hi - m->n_lo + 1);
}

/* std::max / std::min on doubles, with the library's exact tie/NaN behaviour */
static inline double std_max(double a, double b) { return (a < b) ? b : a; }
static inline double std_min(double a, double b) { return (b < a) ? b : a; }

typedef struct {
  const ms_edgematch *em;
  const ms_row       *m1, *m2; /* VertexMatch of this anchor on v1 / v2 */
} elem;

/* nanoCheck lambda, mpp.cpp:40-118 */
static int nano_check(int *pOrientation, double *pDiff, const ms_edgematch *em1, const ms_edgematch *em2,
                      const ms_row *vm1, const ms_row *vm2) {
  double ncl1 = (double)(em1->ov_lo - vm1->i_lo) / rratio(vm1);
  double ncr1 = (double)(vm1->i_hi - em1->ov_hi) / rratio(vm1);
  if (!(vm1->flags & MS_ROW_DIR)) {
    double t = ncl1;
    ncl1 = ncr1;
    ncr1 = t;
  }
  double ncl2 = (double)(em2->ov_lo - vm2->i_lo) / rratio(vm2);
  double ncr2 = (double)(vm2->i_hi - em2->ov_hi) / rratio(vm2);
  if (!(vm2->flags & MS_ROW_DIR)) {
    double t = ncl2;
    ncl2 = ncr2;
    ncr2 = t;
  }
  double c1f = (double)vm1->n_lo + ncl1, c1s = (double)vm1->n_hi - ncr1;
  double c2f = (double)vm2->n_lo + ncl2, c2s = (double)vm2->n_hi - ncr2;

  *pOrientation = 0;
  *pDiff        = 0;
  if (c1f <= c2s && c2f <= c1s) {
    if (c1f < c2f && c1s < c2s) {
      *pOrientation = 2;
      *pDiff        = c1s - c2f + 1;
    }
    if (c1f > c2f && c1s > c2s) {
      *pOrientation = -2;
      *pDiff        = c2s - c1f + 1;
    }
  } else if (c1f < c2f) {
    *pOrientation = 1;
    *pDiff        = c2f - c1s + 1;
  } else {
    *pOrientation = -1;
    *pDiff        = c1f - c2s + 1;
  }

  int uco = 0;
  if (vm1->n_lo <= vm2->n_hi && vm2->n_lo <= vm1->n_hi) {
    if (vm1->n_lo < vm2->n_lo && vm1->n_hi < vm2->n_hi) uco = 2;
    if (vm1->n_lo > vm2->n_lo && vm1->n_hi > vm2->n_hi) uco = -2;
    if ((*pOrientation < 0 && uco >= 0) || (*pOrientation > 0 && uco <= 0)) return 1;
  }
  return 0;
}

/* checkCompatibility, mpp.cpp:38-142 */
static int check_compat(store *s, const elem *a, const elem *b) {
  s->compat_checks++;
  int    o1 = 0, o2 = 0;
  double d1 = 0.0, d2 = 0.0;
  int    abort_ = 0;
  abort_ |= nano_check(&o1, &d1, a->em, b->em, a->m1, b->m1); /* vertices.first  */
  abort_ |= nano_check(&o2, &d2, a->em, b->em, a->m2, b->m2); /* vertices.second */
  if (abort_) return 0;
  if (!(a->em->flags & 1u)) o2 = o2 * -1; /* :131 */
  int matching = 0;
  if (o1 == o2 && o1 != 0) {
    double diff = std_max(d1, d2) - std_min(d1, d2);
    matching    = (diff <= (double)s->p->wiggle_room) || (diff * 100 / std_max(d1, d2) <= s->p->ratio_pct);
  } else if ((o1 < 0 && o2 < 0) || (o1 > 0 && o2 > 0)) {
    matching = d1 + d2 <= (double)s->p->wiggle_room;
  }
  return matching;
}

typedef struct {
  uint32_t *ids; /* anchor ids */
  size_t    n;
  uint64_t  score; /* size_t in path_t, mpp.cpp:34 */
  int       primary;
} path_t;

typedef struct {
  path_t *v;
  size_t  n, cap;
} pathvec;

static void pv_push(pathvec *pv, const uint32_t *ids, size_t n, uint64_t score, int primary) {
  if (pv->n == pv->cap) {
    pv->cap = pv->cap ? pv->cap * 2 : 4;
    pv->v   = (path_t *)xrealloc(pv->v, pv->cap * sizeof(path_t));
  }
  path_t *p  = &pv->v[pv->n++];
  p->ids     = (uint32_t *)xrealloc(NULL, (n ? n : 1) * sizeof(uint32_t));
  memcpy(p->ids, ids, n * sizeof(uint32_t));
  p->n       = n;
  p->score   = score;
  p->primary = primary;
}
static void pv_free(pathvec *pv) {
  for (size_t i = 0; i < pv->n; ++i) free(pv->v[i].ids);
  free(pv->v);
  memset(pv, 0, sizeof(*pv));
}

static int cmp_elem(const void *pa, const void *pb) { /* std::sort(vStart), mpp.cpp:172: (nanoporeRange, id) */
  const elem *a = (const elem *)pa, *b = (const elem *)pb;
  if (a->m1->n_lo != b->m1->n_lo) return a->m1->n_lo < b->m1->n_lo ? -1 : 1;
  if (a->m1->n_hi != b->m1->n_hi) return a->m1->n_hi < b->m1->n_hi ? -1 : 1;
  if (a->em->anchor_id != b->em->anchor_id) return a->em->anchor_id < b->em->anchor_id ? -1 : 1;
  return 0;
}

typedef struct {
  size_t *path; /* indices into vStart */
  size_t  n;
  double  score;
} pop_t;

/* position of anchor `id` in read `rd`'s sorted match list, searching from `from`; returns list size if absent
 * (std::find_if; an out-of-range start behaves like libstdc++: returns end()) */
static long find_from(const store *s, uint32_t rd, int reversed, long from, uint32_t id) {
  long n = (long)(s->rd_off[rd + 1] - s->rd_off[rd]);
  for (long q = from; q < n; ++q) {
    long     src = reversed ? n - 1 - q : q;
    uint32_t aid = s->rows[s->rd_rows[s->rd_off[rd] + (size_t)src]].anchor_id;
    if (aid == id) return q;
  }
  return n;
}
static uint32_t list_id_at(const store *s, uint32_t rd, int reversed, long q) {
  long n   = (long)(s->rd_off[rd + 1] - s->rd_off[rd]);
  long src = reversed ? n - 1 - q : q;
  return s->rows[s->rd_rows[s->rd_off[rd] + (size_t)src]].anchor_id;
}

/* getMaxPairwisePaths, mpp.cpp:145-305.  `el` = the edge's EdgeMatches of one direction (any order). */
static void get_max_pairwise_paths(store *s, uint32_t v1, uint32_t v2, elem *el, size_t n, int direction,
                                   pathvec *result) {
  if (n == 0) return; /* :150-152 */
  qsort(el, n, sizeof(elem), cmp_elem); /* vStart; vEnd (:173-175) is computed but never read by the DP */

  pop_t *pop = (pop_t *)xrealloc(NULL, n * sizeof(pop_t));
  for (size_t i = 0; i < n; ++i) { /* :181-183 */
    pop[i].path  = NULL;
    pop[i].n     = 0;
    pop[i].score = el[i].em->score;
  }
  size_t limit = (n > 1 ? n : 1) - 1; /* :185 */
  for (size_t k = 0; k < limit; ++k) {
    for (size_t l = k + 1; l <= limit; ++l) {
      int    ok    = check_compat(s, &el[k], &el[l]);
      double score = pop[k].score + el[l].em->score;
      ok &= score > pop[l].score;
      if (ok) { /* :192-197 */
        size_t *np = (size_t *)xrealloc(NULL, (pop[k].n + 1) * sizeof(size_t));
        memcpy(np, pop[k].path, pop[k].n * sizeof(size_t));
        np[pop[k].n] = k;
        free(pop[l].path);
        pop[l].path  = np;
        pop[l].n     = pop[k].n + 1;
        pop[l].score = score;
      }
    }
  }
  double maxv = 0.0; /* :201-210 */
  size_t maxi = 0;
  for (size_t i = 0; i < n; ++i) {
    pop[i].path            = (size_t *)xrealloc(pop[i].path, (pop[i].n + 1) * sizeof(size_t));
    pop[i].path[pop[i].n++] = i;
    if (pop[i].score > maxv) {
      maxi = i;
      maxv = pop[i].score;
    }
  }
  uint32_t *tmp = (uint32_t *)xrealloc(NULL, n * sizeof(uint32_t));
  {
    const pop_t *mp = &pop[maxi]; /* :212-221 */
    int          hp = 0;
    for (size_t i = 0; i < mp->n; ++i) {
      tmp[i] = el[mp->path[i]].em->anchor_id;
      hp |= (el[mp->path[i]].em->flags >> 1) & 1;
    }
    hp |= mp->n > 2;
    pv_push(result, tmp, mp->n, (uint64_t)maxv, hp);
  }
  double thr = maxv * s->p->alt_frac; /* :223 */
  for (size_t e = 0; e < n; ++e) {    /* :224-249 */
    if (!(pop[e].score > thr)) continue;
    int disjoint = 1;
    for (size_t r = 0; r < result->n; ++r)
      for (size_t i = 0; i < pop[e].n; ++i) {
        uint32_t id = el[pop[e].path[i]].em->anchor_id;
        for (size_t j = 0; j < result->v[r].n; ++j)
          if (result->v[r].ids[j] == id) disjoint = 0;
      }
    if (!disjoint) continue;
    int prim = 0;
    for (size_t i = 0; i < pop[e].n; ++i) {
      tmp[i] = el[pop[e].path[i]].em->anchor_id;
      prim |= (el[pop[e].path[i]].em->flags >> 1) & 1;
    }
    pv_push(result, tmp, pop[e].n, (uint64_t)pop[e].score, prim);
  }
  free(tmp);
  for (size_t i = 0; i < n; ++i) free(pop[i].path);
  free(pop);

  /* :251-302 single primary result: demote unless anchored at a read end and not interleaved */
  if (result->n == 1 && result->v[0].primary) {
    const path_t *P   = &result->v[0];
    int           rev = !direction; /* vIDsEnd reversed when !direction, :269 */
    long          nS  = (long)(s->rd_off[v1 + 1] - s->rd_off[v1]);
    long          nE  = (long)(s->rd_off[v2 + 1] - s->rd_off[v2]);
    uint32_t      pf = P->ids[0], pb = P->ids[P->n - 1];
    if ((list_id_at(s, v1, 0, 0) != pf && list_id_at(s, v2, rev, 0) != pf) ||
        (list_id_at(s, v1, 0, nS - 1) != pb && list_id_at(s, v2, rev, nE - 1) != pb)) {
      result->v[0].primary = 0;
    } else {
      long i = 0, j = 0;
      int  is_shadow = 0;
      for (size_t t = 0; !is_shadow && t < P->n; ++t) {
        long rs = find_from(s, v1, 0, i, P->ids[t]);
        int  inter = rs > i;
        i += (rs - i) + 1;
        long re = find_from(s, v2, rev, j, P->ids[t]);
        inter &= re > j;
        j += (re - j) + 1;
        is_shadow = inter;
      }
      if (is_shadow) result->v[0].primary = 0;
    }
  }
}

/* getOverhangs, ol.cpp:31-50 */
static void get_overhangs(const store *s, uint32_t v, const ms_row *vm, const ms_edgematch *em, double *left,
                          double *right) {
  double ncl = (double)(em->ov_lo - vm->i_lo) / rratio(vm);
  double ncr = (double)(vm->i_hi - em->ov_hi) / rratio(vm);
  if (!(vm->flags & MS_ROW_DIR)) {
    double t = ncl;
    ncl = ncr;
    ncr = t;
  }
  *left   = (double)vm->n_lo + ncl;
  int len = (int)s->read_len[v];
  *right  = (double)(len - vm->n_hi) + ncr;
}

/* ------------------------------------------------------------------------------------------------------------------
 * driver: A1 tail + A2/A3 (MatchMap.cpp:161-224) + A7 (src/main.cpp:328-414)
 * ---------------------------------------------------------------------------------------------------------------- */

static const ms_row *g_sort_rows;
static int cmp_line_idx(const void *a, const void *b) {
  uint32_t la = g_sort_rows[*(const uint32_t *)a].line, lb = g_sort_rows[*(const uint32_t *)b].line;
  if (la != lb) return la < lb ? -1 : 1;
  uint32_t ia = *(const uint32_t *)a, ib = *(const uint32_t *)b;
  return ia < ib ? -1 : ia > ib;
}
static int cmp_row_line(const void *a, const void *b) {
  const ms_row *x = (const ms_row *)a, *y = (const ms_row *)b;
  return x->line < y->line ? -1 : x->line > y->line;
}
static int cmp_read_list(const void *a, const void *b) {
  const ms_row *x = &g_sort_rows[*(const uint32_t *)a], *y = &g_sort_rows[*(const uint32_t *)b];
  if (x->n_lo != y->n_lo) return x->n_lo < y->n_lo ? -1 : 1;
  if (x->n_hi != y->n_hi) return x->n_hi < y->n_hi ? -1 : 1;
  return x->anchor_id < y->anchor_id ? -1 : x->anchor_id > y->anchor_id;
}

typedef struct {
  uint32_t v1, v2;
} ekey;

typedef struct {
  ms_edgematch em;
  uint32_t     edge_tmp;
} em_tmp;

typedef struct {
  ms_edgematch em;
  int32_t      n_lo, n_hi;
} em_sort;
static int cmp_em_sort(const void *a, const void *b) {
  const em_sort *x = (const em_sort *)a, *y = (const em_sort *)b;
  if (x->n_lo != y->n_lo) return x->n_lo < y->n_lo ? -1 : 1;
  if (x->n_hi != y->n_hi) return x->n_hi < y->n_hi ? -1 : 1;
  return x->em.anchor_id < y->em.anchor_id ? -1 : x->em.anchor_id > y->em.anchor_id;
}

static const ekey *g_sort_edges;
static int cmp_edge_idx(const void *a, const void *b) {
  const ekey *x = &g_sort_edges[*(const uint32_t *)a], *y = &g_sort_edges[*(const uint32_t *)b];
  if (x->v1 != y->v1) return x->v1 < y->v1 ? -1 : 1;
  return x->v2 < y->v2 ? -1 : x->v2 > y->v2;
}

int ms_oracle_overlap(const ms_row *rows_in, size_t n_rows, const ms_params *p, ms_tables *out) {
  memset(out, 0, sizeof(*out));
  if (n_rows && !rows_in) return MS_E_ARG;
  if (n_rows >= UINT32_MAX) return MS_E_ARG;
  store S;
  memset(&S, 0, sizeof(S));
  S.p = p;

  /* single-thread reference order = ascending line index */
  ms_row *rows = (ms_row *)xrealloc(NULL, n_rows * sizeof(ms_row));
  memcpy(rows, rows_in, n_rows * sizeof(ms_row));
  {
    uint32_t *perm = (uint32_t *)xrealloc(NULL, n_rows * sizeof(uint32_t));
    for (size_t i = 0; i < n_rows; ++i) perm[i] = (uint32_t)i;
    g_sort_rows = rows_in;
    qsort(perm, n_rows, sizeof(uint32_t), cmp_line_idx);
    for (size_t i = 0; i < n_rows; ++i) rows[i] = rows_in[perm[i]];
    free(perm);
  }
  (void)cmp_row_line;
  S.rows   = rows;
  S.n_rows = n_rows;
  for (size_t i = 0; i < n_rows; ++i) {
    if (rows[i].read_id + 1 > S.n_reads) S.n_reads = rows[i].read_id + 1;
    if (rows[i].anchor_id + 1 > S.n_anchor_ids) S.n_anchor_ids = rows[i].anchor_id + 1;
  }
  S.read_len        = (int32_t *)calloc(S.n_reads ? S.n_reads : 1, sizeof(int32_t));
  S.read_first_line = (uint32_t *)calloc(S.n_reads ? S.n_reads : 1, sizeof(uint32_t));
  S.read_exists     = (uint8_t *)calloc(S.n_reads ? S.n_reads : 1, 1);
  map_init(&S.vm, n_rows);

  /* addVertex (Graph.cpp:148 emplace: first wins) + addVertexMatch (MatchMap.cpp:52-81: lowest line wins) */
  uint8_t *alive = (uint8_t *)calloc(n_rows ? n_rows : 1, 1);
  for (size_t i = 0; i < n_rows; ++i) {
    const ms_row *r = &rows[i];
    if (!S.read_exists[r->read_id]) {
      S.read_exists[r->read_id]     = 1;
      S.read_len[r->read_id]        = r->read_len;
      S.read_first_line[r->read_id] = r->line;
    }
    int       ins = 0;
    uint32_t *v   = map_insert(&S.vm, vmkey(r->read_id, r->anchor_id), (uint32_t)i, &ins);
    if (ins) {
      alive[i] = 1;
    } else if (rows[*v].line > r->line) {
      alive[*v] = 0;
      *v        = (uint32_t)i;
      alive[i]  = 1;
    }
  }

  /* scaffolds: anchor -> alive rows in line order (rows[] is in line order already) */
  size_t   *sc_off  = (size_t *)calloc((size_t)S.n_anchor_ids + 2, sizeof(size_t));
  size_t   *rd_off  = (size_t *)calloc((size_t)S.n_reads + 2, sizeof(size_t));
  uint64_t  n_alive = 0;
  for (size_t i = 0; i < n_rows; ++i)
    if (alive[i]) {
      sc_off[rows[i].anchor_id + 1]++;
      rd_off[rows[i].read_id + 1]++;
      n_alive++;
    }
  for (uint32_t a = 0; a < S.n_anchor_ids; ++a) sc_off[a + 1] += sc_off[a];
  for (uint32_t r = 0; r < S.n_reads; ++r) rd_off[r + 1] += rd_off[r];
  uint32_t *sc_rows = (uint32_t *)xrealloc(NULL, (n_alive ? n_alive : 1) * sizeof(uint32_t));
  uint32_t *rd_rows = (uint32_t *)xrealloc(NULL, (n_alive ? n_alive : 1) * sizeof(uint32_t));
  {
    size_t *c1 = (size_t *)xrealloc(NULL, ((size_t)S.n_anchor_ids + 1) * sizeof(size_t));
    size_t *c2 = (size_t *)xrealloc(NULL, ((size_t)S.n_reads + 1) * sizeof(size_t));
    memcpy(c1, sc_off, ((size_t)S.n_anchor_ids + 1) * sizeof(size_t));
    memcpy(c2, rd_off, ((size_t)S.n_reads + 1) * sizeof(size_t));
    for (size_t i = 0; i < n_rows; ++i)
      if (alive[i]) {
        sc_rows[c1[rows[i].anchor_id]++] = (uint32_t)i;
        rd_rows[c2[rows[i].read_id]++]   = (uint32_t)i;
      }
    free(c1);
    free(c2);
  }
  g_sort_rows = rows;
  for (uint32_t r = 0; r < S.n_reads; ++r)
    qsort(rd_rows + rd_off[r], rd_off[r + 1] - rd_off[r], sizeof(uint32_t), cmp_read_list);
  S.rd_off  = rd_off;
  S.rd_rows = rd_rows;
  out->rows_alive = n_alive;

  /* MatchMap::calculateEdges / processScaffold, MatchMap.cpp:161-224 */
  map64   emap;
  map_init(&emap, 1024);
  ekey   *ekeys = NULL;
  size_t  n_edges = 0, cap_edges = 0;
  em_tmp *ems = NULL;
  size_t  n_ems = 0, cap_ems = 0;
  for (uint32_t a = 0; a < S.n_anchor_ids; ++a) {
    size_t b = sc_off[a], e = sc_off[a + 1];
    if (e > b) out->n_anchors++;
    for (size_t oi = b + 1; oi < e; ++oi) {
      const ms_row *om = &rows[sc_rows[oi]];
      for (size_t ii = b; ii < oi; ++ii) {
        const ms_row *im = &rows[sc_rows[ii]];
        out->p_eval++;
        int ov_lo = om->i_lo > im->i_lo ? om->i_lo : im->i_lo;
        int ov_hi = om->i_hi < im->i_hi ? om->i_hi : im->i_hi;
        if (ov_lo <= ov_hi && ov_hi - ov_lo > (int)p->th_overlap) { /* :192 */
          int    dir  = ((om->flags & MS_ROW_DIR) != 0) == ((im->flags & MS_ROW_DIR) != 0);
          int    prim = (om->flags & MS_ROW_PRIMARY) && (im->flags & MS_ROW_PRIMARY);
          double ol   = (double)(om->i_hi - om->i_lo + 1);
          double il   = (double)(im->i_hi - im->i_lo + 1);
          double cl   = (double)(ov_hi - ov_lo + 1);
          double os   = (double)om->score * cl / ol;
          double is_  = (double)im->score * cl / il;
          double sum  = os + is_;
          uint32_t vo = om->read_id, vi = im->read_id;
          ekey     k;
          if (S.read_first_line[vo] < S.read_first_line[vi]) { /* :204-213 */
            k.v1 = vo;
            k.v2 = vi;
          } else {
            k.v1 = vi;
            k.v2 = vo;
          }
          int       ins = 0;
          uint32_t *ei  = map_insert(&emap, ((uint64_t)k.v1 << 32) | k.v2, (uint32_t)n_edges, &ins);
          if (ins) {
            if (n_edges == cap_edges) {
              cap_edges = cap_edges ? cap_edges * 2 : 1024;
              ekeys     = (ekey *)xrealloc(ekeys, cap_edges * sizeof(ekey));
            }
            ekeys[n_edges++] = k;
          }
          if (n_ems == cap_ems) {
            cap_ems = cap_ems ? cap_ems * 2 : 4096;
            ems     = (em_tmp *)xrealloc(ems, cap_ems * sizeof(em_tmp));
          }
          em_tmp *t        = &ems[n_ems++];
          t->em.ov_lo      = ov_lo;
          t->em.ov_hi      = ov_hi;
          t->em.score      = sum;
          t->em.anchor_id  = a;
          t->em.line       = om->line; /* :218 */
          t->em.flags      = (dir ? 1u : 0u) | (prim ? 2u : 0u);
          t->em.edge_idx   = 0;
          t->edge_tmp      = *ei;
        }
      }
    }
  }
  map_free(&emap);

  /* canonical edge order (v1, v2); EdgeMatches grouped by edge, anchors ascending (stable counting sort) */
  uint32_t *eperm = (uint32_t *)xrealloc(NULL, (n_edges ? n_edges : 1) * sizeof(uint32_t));
  uint32_t *einv  = (uint32_t *)xrealloc(NULL, (n_edges ? n_edges : 1) * sizeof(uint32_t));
  for (size_t i = 0; i < n_edges; ++i) eperm[i] = (uint32_t)i;
  g_sort_edges = ekeys;
  qsort(eperm, n_edges, sizeof(uint32_t), cmp_edge_idx);
  for (size_t i = 0; i < n_edges; ++i) einv[eperm[i]] = (uint32_t)i;
  ms_edge *edges = (ms_edge *)calloc(n_edges ? n_edges : 1, sizeof(ms_edge));
  for (size_t i = 0; i < n_edges; ++i) {
    edges[i].v1 = ekeys[eperm[i]].v1;
    edges[i].v2 = ekeys[eperm[i]].v2;
  }
  for (size_t i = 0; i < n_ems; ++i) edges[einv[ems[i].edge_tmp]].em_cnt++;
  {
    uint64_t off = 0;
    for (size_t i = 0; i < n_edges; ++i) {
      edges[i].em_off = off;
      off += edges[i].em_cnt;
    }
  }
  ms_edgematch *emt = (ms_edgematch *)xrealloc(NULL, (n_ems ? n_ems : 1) * sizeof(ms_edgematch));
  {
    uint64_t *cur = (uint64_t *)xrealloc(NULL, (n_edges ? n_edges : 1) * sizeof(uint64_t));
    for (size_t i = 0; i < n_edges; ++i) cur[i] = edges[i].em_off;
    for (size_t i = 0; i < n_ems; ++i) {
      uint32_t e       = einv[ems[i].edge_tmp];
      emt[cur[e]]      = ems[i].em;
      emt[cur[e]].edge_idx = e;
      cur[e]++;
    }
    free(cur);
  }
  free(ems);
  free(eperm);
  free(einv);
  free(ekeys);

  /* canonical EdgeMatch order inside an edge = the vStart order of mpp.cpp:164-172:
   * (nanoporeRange of the anchor on v1, anchor id) */
  {
    em_sort *tmp = NULL;
    size_t   cap = 0;
    for (size_t i = 0; i < n_edges; ++i) {
      ms_edge *E = &edges[i];
      if (E->em_cnt > cap) {
        cap = (size_t)E->em_cnt * 2;
        tmp = (em_sort *)xrealloc(tmp, cap * sizeof(em_sort));
      }
      for (uint32_t q = 0; q < E->em_cnt; ++q) {
        const ms_row *m = get_vertex_match(&S, E->v1, emt[E->em_off + q].anchor_id);
        tmp[q].em   = emt[E->em_off + q];
        tmp[q].n_lo = m->n_lo;
        tmp[q].n_hi = m->n_hi;
      }
      qsort(tmp, E->em_cnt, sizeof(em_sort), cmp_em_sort);
      for (uint32_t q = 0; q < E->em_cnt; ++q) emt[E->em_off + q] = tmp[q].em;
    }
    free(tmp);
  }

  /* chainingAndOverlaps, src/main.cpp:328-414 */
  ms_order *orders = NULL;
  size_t    n_orders = 0, cap_orders = 0;
  uint32_t *ids = NULL;
  size_t    n_ids = 0, cap_ids = 0;
  elem     *plus = NULL, *minus = NULL;
  size_t    cap_el = 0;
  for (size_t ei = 0; ei < n_edges; ++ei) {
    ms_edge *E = &edges[ei];
    E->order_off = n_orders;
    if (E->em_cnt == 0) continue; /* :336-339 */
    if (E->em_cnt > cap_el) {
      cap_el = E->em_cnt * 2;
      plus   = (elem *)xrealloc(plus, cap_el * sizeof(elem));
      minus  = (elem *)xrealloc(minus, cap_el * sizeof(elem));
    }
    size_t np = 0, nm = 0;
    for (uint32_t i = 0; i < E->em_cnt; ++i) { /* :341-349 */
      const ms_edgematch *em = &emt[E->em_off + i];
      elem                x;
      x.em = em;
      x.m1 = get_vertex_match(&S, E->v1, em->anchor_id);
      x.m2 = get_vertex_match(&S, E->v2, em->anchor_id);
      if (em->flags & 1u)
        plus[np++] = x;
      else
        minus[nm++] = x;
    }
    pathvec mp, pp;
    memset(&mp, 0, sizeof(mp));
    memset(&pp, 0, sizeof(pp));
    get_max_pairwise_paths(&S, E->v1, E->v2, minus, nm, 0, &mp); /* :352 */
    get_max_pairwise_paths(&S, E->v1, E->v2, plus, np, 1, &pp);  /* :353 */

    int has_primary = 0; /* :355-370 */
    for (size_t i = 0; i < pp.n; ++i) has_primary |= pp.v[i].primary;
    if (!has_primary)
      for (size_t i = 0; i < mp.n; ++i) has_primary |= mp.v[i].primary;
    uint8_t *keep_p = (uint8_t *)xrealloc(NULL, pp.n + 1), *keep_m = (uint8_t *)xrealloc(NULL, mp.n + 1);
    for (size_t i = 0; i < pp.n; ++i) keep_p[i] = !has_primary || pp.v[i].primary;
    for (size_t i = 0; i < mp.n; ++i) keep_m[i] = !has_primary || mp.v[i].primary;
    int has_multi = 0; /* :372-387 */
    for (size_t i = 0; i < pp.n; ++i) has_multi |= keep_p[i] && pp.v[i].n > 1;
    if (!has_multi)
      for (size_t i = 0; i < mp.n; ++i) has_multi |= keep_m[i] && mp.v[i].n > 1;
    if (has_multi) {
      for (size_t i = 0; i < pp.n; ++i) keep_p[i] &= pp.v[i].n > 1;
      for (size_t i = 0; i < mp.n; ++i) keep_m[i] &= mp.v[i].n > 1;
    }
    size_t cp = 0, cm = 0;
    for (size_t i = 0; i < pp.n; ++i) cp += keep_p[i];
    for (size_t i = 0; i < mp.n; ++i) cm += keep_m[i];
    if (cp + cm > 1) { /* :389-395 */
      E->shadow = 1;
    } else {
      const path_t *first = NULL;
      for (size_t i = 0; i < mp.n && !first; ++i)
        if (keep_m[i]) first = &mp.v[i];
      for (size_t i = 0; i < pp.n && !first; ++i)
        if (keep_p[i]) first = &pp.v[i];
      E->shadow = first ? !first->primary : 0;
    }
    if (E->shadow) out->shadow_edges++;

    for (int pass = 0; pass < 2; ++pass) { /* minus first (:397-403), then plus (:405-411) */
      pathvec *pv   = pass == 0 ? &mp : &pp;
      uint8_t *keep = pass == 0 ? keep_m : keep_p;
      int      dir  = pass == 0 ? 0 : 1;
      for (size_t i = 0; i < pv->n; ++i) {
        if (!keep[i]) continue;
        const path_t *P = &pv->v[i];
        /* getOverlap, ol.cpp:53-101 */
        uint32_t            fid = P->ids[0], lid = P->ids[P->n - 1];
        const ms_edgematch *emf = NULL, *eml = NULL;
        for (uint32_t q = 0; q < E->em_cnt; ++q) { /* getEdgeMatch, MatchMap.cpp:136-147 */
          if (emt[E->em_off + q].anchor_id == fid) emf = &emt[E->em_off + q];
          if (emt[E->em_off + q].anchor_id == lid) eml = &emt[E->em_off + q];
        }
        double f1l, f1r, l1l, l1r, f2l, f2r, l2l, l2r;
        get_overhangs(&S, E->v1, get_vertex_match(&S, E->v1, fid), emf, &f1l, &f1r);
        get_overhangs(&S, E->v1, get_vertex_match(&S, E->v1, lid), eml, &l1l, &l1r);
        get_overhangs(&S, E->v2, get_vertex_match(&S, E->v2, fid), emf, &f2l, &f2r);
        get_overhangs(&S, E->v2, get_vertex_match(&S, E->v2, lid), eml, &l2l, &l2r);
        double L1 = f1l, R1 = l1r, L2 = f2l, R2 = l2r;
        if (!dir) { /* :73-76 */
          L2 = f2r;
          R2 = l2l;
        }
        int      have = 0;
        ms_order o;
        memset(&o, 0, sizeof(o));
        if (L1 <= L2 && R1 <= R2) { /* :79-82 */
          have = 1;
          o.flags = MS_ORD_START_V1 | MS_ORD_CONTAINED;
          o.left_offset  = L2 - L1;
          o.right_offset = R2 - R1;
        } else if (L1 >= L2 && R1 >= R2) { /* :83-87 */
          have = 1;
          o.flags = MS_ORD_CONTAINED;
          o.left_offset  = L1 - L2;
          o.right_offset = R1 - R2;
        } else if (L1 > L2 && R1 < R2) { /* :88-92 */
          have = 1;
          o.flags = MS_ORD_START_V1;
          o.left_offset  = L1 - L2;
          o.right_offset = R2 - R1;
        } else if (L1 < L2 && R1 > R2) { /* :93-97 */
          have = 1;
          o.flags = 0;
          o.left_offset  = L2 - L1;
          o.right_offset = R1 - R2;
        }
        if (!have) continue;
        o.edge_idx = (uint32_t)ei;
        o.flags |= (dir ? MS_ORD_DIR : 0u) | (P->primary ? MS_ORD_PRIMARY : 0u);
        o.score   = P->score;
        o.ids_off = n_ids;
        o.ids_cnt = (uint32_t)P->n;
        o.start   = (o.flags & MS_ORD_START_V1) ? E->v1 : E->v2;
        o.end     = (o.flags & MS_ORD_START_V1) ? E->v2 : E->v1;
        o.base    = E->v1;
        if (n_ids + P->n > cap_ids) {
          cap_ids = (n_ids + P->n) * 2;
          ids     = (uint32_t *)xrealloc(ids, cap_ids * sizeof(uint32_t));
        }
        memcpy(ids + n_ids, P->ids, P->n * sizeof(uint32_t));
        n_ids += P->n;
        if (n_orders == cap_orders) {
          cap_orders = cap_orders ? cap_orders * 2 : 1024;
          orders     = (ms_order *)xrealloc(orders, cap_orders * sizeof(ms_order));
        }
        orders[n_orders++] = o;
        E->order_cnt++;
      }
    }
    free(keep_p);
    free(keep_m);
    pv_free(&mp);
    pv_free(&pp);
  }
  free(plus);
  free(minus);

  out->edges           = edges;
  out->n_edges         = n_edges;
  out->ems             = emt;
  out->n_ems           = n_ems;
  out->orders          = orders ? orders : (ms_order *)calloc(1, sizeof(ms_order));
  out->n_orders        = n_orders;
  out->ids             = ids ? ids : (uint32_t *)calloc(1, sizeof(uint32_t));
  out->n_ids           = n_ids;
  out->read_len        = S.read_len;
  out->read_first_line = S.read_first_line;
  out->n_reads         = S.n_reads;
  out->compat_checks   = S.compat_checks;

  free(S.read_exists);
  map_free(&S.vm);
  free(alive);
  free(sc_off);
  free(sc_rows);
  free(rd_off);
  free(rd_rows);
  free(rows);
  return MS_OK;
}

void ms_oracle_free_tables(ms_tables *t) {
  free(t->edges);
  free(t->ems);
  free(t->orders);
  free(t->ids);
  free(t->read_len);
  free(t->read_first_line);
  memset(t, 0, sizeof(*t));
}

/* ------------------------------------------------------------------------------------------------------------------
 * sequence side (A9, data-parallel half)
 * ---------------------------------------------------------------------------------------------------------------- */

#include <ctype.h>

typedef struct {
  const char *buf;
  size_t      len, pos;
  const char *line;     /* last line read (like pLine: unchanged by a failed read) */
  size_t      line_len; /* including the '\n' */
} rl_state;

/* readline (IO.cpp:54-97) over a memory image: returns the line length incl. '\n', or -1 at EOF */
static long rl_next(rl_state *st) {
  if (st->pos >= st->len) return -1;
  const char *nl = (const char *)memchr(st->buf + st->pos, '\n', st->len - st->pos);
  size_t      e  = nl ? (size_t)(nl - st->buf) + 1 : st->len;
  st->line       = st->buf + st->pos;
  st->line_len   = e - st->pos;
  st->pos        = e;
  return (long)st->line_len;
}

static int seq_is_fastq(const char *path) { /* isFastQ, SequenceAccessor.cpp:71-80 */
  const char *dot = strrchr(path, '.');
  const char *ext = dot ? dot + 1 : path; /* find_last_of('.') == npos -> substr(0) = whole name */
  char        low[64];
  size_t      i = 0;
  for (; ext[i] && i + 1 < sizeof(low); ++i) low[i] = (char)tolower((unsigned char)ext[i]);
  low[i] = '\0';
  return strcmp(low, "fa") != 0 && strcmp(low, "fasta") != 0;
}

int ms_oracle_seq_load(const char *path, int is_fastq, ms_seqs *out) {
  memset(out, 0, sizeof(*out));
  FILE *f = fopen(path, "rb");
  if (!f) return MS_E_IO; /* "Can't open sequence file(s)." SequenceAccessor.cpp:105-107 */
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *buf = (char *)xrealloc(NULL, (size_t)sz + 1);
  if (sz > 0 && fread(buf, 1, (size_t)sz, f) != (size_t)sz) {
    fclose(f);
    free(buf);
    return MS_E_IO;
  }
  fclose(f);
  if (is_fastq < 0) is_fastq = seq_is_fastq(path);
  const char desc  = is_fastq ? '@' : '>'; /* SequenceAccessor.cpp:40-42,144-145 */
  const char split = is_fastq ? '+' : '>';

  registry reg;
  reg_init(&reg);
  size_t    cap = 64, n = 0;
  uint64_t *rec_off  = (uint64_t *)xrealloc(NULL, cap * sizeof(uint64_t)); /* file offset of the record body */
  uint64_t *rec_len  = (uint64_t *)xrealloc(NULL, cap * sizeof(uint64_t)); /* raw length incl. newlines      */
  uint8_t  *rec_seen = (uint8_t *)calloc(cap, 1);

  rl_state st  = {buf, (size_t)sz, 0, "", 0};
  long     ret = rl_next(&st);
  size_t   offset_start = st.pos;
  while (ret != -1) { /* :149-158 skip to the first description line */
    if (st.line[0] == desc) break;
    ret          = rl_next(&st);
    offset_start = st.pos;
  }
  /* :160 `while (*pLine == desc)`; at EOF pLine keeps the last line, which the reference would test again -- a
   * last line starting with the description character loops forever there; here EOF ends the loop. */
  while (ret != -1 && st.line[0] == desc) {
    const char *idp = st.line + 1; /* :161-163 id = line + 1, cut at the first whitespace (cleanSequenceId) */
    size_t      idl = 0;
    while (1 + idl < st.line_len && !isspace((unsigned char)idp[idl])) ++idl;
    const uint32_t id = reg_get(&reg, idp, idl); /* (*m_pRegistry)[sequenceId] */
    size_t         length = 0;
    while (1) { /* :167-179 */
      ret = rl_next(&st);
      const size_t offset_end = st.pos;
      if (ret == -1 || st.line[0] == split) {
        if (id >= cap) {
          size_t nc = cap;
          while (nc <= id) nc *= 2;
          rec_off  = (uint64_t *)xrealloc(rec_off, nc * sizeof(uint64_t));
          rec_len  = (uint64_t *)xrealloc(rec_len, nc * sizeof(uint64_t));
          rec_seen = (uint8_t *)xrealloc(rec_seen, nc);
          memset(rec_seen + cap, 0, nc - cap);
          cap = nc;
        }
        if (!rec_seen[id]) { /* unordered_map::emplace keeps the first record of a name */
          rec_seen[id] = 1;
          rec_off[id]  = offset_start;
          rec_len[id]  = length;
          if (id + 1 > n) n = id + 1;
        }
        offset_start = offset_end;
        break;
      }
      length += (size_t)ret;
    }
    /* :181-184 (nanopore file only; a no-op for FASTA where split == desc): skip to the next description line */
    while (ret != -1 && st.line[0] != desc) {
      ret          = rl_next(&st);
      offset_start = st.pos;
    }
  }

  /* getSequenceFromFile for every record (:54-69): rec_len bytes, cut at the first NUL, isspace removed */
  out->n   = (uint32_t)n;
  out->off = (uint64_t *)xrealloc(NULL, (n + 1) * sizeof(uint64_t));
  size_t total = 0;
  for (size_t i = 0; i < n; ++i) total += rec_len[i];
  out->bases = (char *)xrealloc(NULL, total + 1);
  size_t w = 0;
  for (size_t i = 0; i < n; ++i) {
    out->off[i] = w;
    for (size_t k = 0; k < rec_len[i]; ++k) {
      char ch = buf[rec_off[i] + k];
      if (ch == '\0') break;
      if (!isspace((unsigned char)ch)) out->bases[w++] = ch;
    }
  }
  out->off[n]    = w;
  out->names     = reg.blob;
  out->names_len = reg.blob_len;
  reg.blob       = NULL;
  reg_free(&reg);
  free(rec_off);
  free(rec_len);
  free(rec_seen);
  free(buf);
  return MS_OK;
}

void ms_oracle_seq_free(ms_seqs *s) {
  free(s->names);
  free(s->bases);
  free(s->off);
  memset(s, 0, sizeof(*s));
}

size_t ms_oracle_str_slice(size_t size, int start, int end, size_t *len) { /* SequenceUtils.cpp:27-38 */
  int    sz = (int)size;
  int    i  = start >= 0 ? start : sz + start;
  int    j  = end >= 0 ? end : sz + end;
  size_t s  = (size_t)(i > 0 ? i : 0);
  size_t jj = (size_t)(j > 0 ? j : 0);
  size_t e  = size < jj ? size : jj;  /* min(original.size(), max(0, j)) */
  size_t ic = (size_t)i;              /* static_cast<std::size_t>(i): a negative i wraps to a huge value */
  if (e < ic) e = ic;                 /* max(..., size_t(i)) */
  /* substr(s, e - s + 1): throws std::out_of_range when s > size; clipped at size otherwise */
  if (s > size) {
    *len = 0;
    return size;
  }
  size_t cnt = e - s + 1;
  if (cnt > size - s) cnt = size - s;
  *len = cnt;
  return s;
}

void ms_oracle_revcomp(const char *in, size_t n, char *out) { /* SequenceUtils.cpp:41-61 */
  for (size_t i = 0; i < n; ++i) {
    char c = in[n - 1 - i];
    switch (c) {
    case 'A': c = 'T'; break;
    case 'T': c = 'A'; break;
    case 'G': c = 'C'; break;
    case 'C': c = 'G'; break;
    default: break;
    }
    out[i] = c;
  }
}

size_t ms_oracle_get_sequence(const char *seq, size_t size, int left, int right, int direction, char *out) {
  size_t len;
  size_t s = ms_oracle_str_slice(size, left, right + 1, &len); /* SequenceUtils.cpp:66,78 */
  if (direction)
    memcpy(out, seq + s, len);
  else
    ms_oracle_revcomp(seq + s, len, out);
  return len;
}

/* ------------------------------------------------------------------------------------------------------------------
 * segment builders of assemblePath (ap.cpp:191-203, 352-579)
 * ---------------------------------------------------------------------------------------------------------------- */

#include <math.h>

static size_t append_seq(char *out, size_t w, const char *seq, size_t size, int left, int right, int direction) {
  return w + ms_oracle_get_sequence(seq, size, left, right, direction, out + w);
}
static void revcomp_inplace(char *s, size_t n) {
  char *tmp = (char *)xrealloc(NULL, n ? n : 1);
  ms_oracle_revcomp(s, n, tmp);
  memcpy(s, tmp, n);
  free(tmp);
}

size_t ms_oracle_anchor_sequence(const ms_row *m, const char *illu, size_t illu_len, int ov_lo, int ov_hi,
                                 int direction, char *out) {
  /* combinedDirection = match.direction * direction (Toggle: XNOR), ap.cpp:428 */
  int combined = ((m->flags & MS_ROW_DIR) != 0) == (direction != 0);
  return append_seq(out, 0, illu, illu_len, ov_lo, ov_hi, combined);
}

size_t ms_oracle_left_of_anchor(const ms_row *m, const char *nano, size_t nano_len, const char *illu, size_t illu_len,
                                size_t nanopore_length, int ov_lo, int ov_hi, int direction, char *out) {
  int    mdir = (m->flags & MS_ROW_DIR) != 0;
  size_t w    = 0;
  if (!direction) { /* :358-373 */
    if (!mdir)
      w = append_seq(out, w, illu, illu_len, m->i_lo, ov_lo, 0);
    else
      w = append_seq(out, w, illu, illu_len, ov_hi, m->i_hi, 1);
    w = append_seq(out, w, nano, nano_len, m->n_hi, (int)nanopore_length - 1, 1);
    revcomp_inplace(out, w);
    return w;
  }
  w = append_seq(out, w, nano, nano_len, 0, m->n_lo, 1); /* :375 */
  if (!mdir)
    w = append_seq(out, w, illu, illu_len, ov_hi, m->i_hi, 0);
  else
    w = append_seq(out, w, illu, illu_len, m->i_lo, ov_lo, 1);
  return w;
}

size_t ms_oracle_right_of_anchor(const ms_row *m, const char *nano, size_t nano_len, const char *illu, size_t illu_len,
                                 size_t nanopore_length, int ov_lo, int ov_hi, int direction, char *out) {
  int    mdir = (m->flags & MS_ROW_DIR) != 0;
  size_t w    = 0;
  if (!direction) { /* :394-406 */
    w = append_seq(out, w, nano, nano_len, 0, m->n_lo, 1);
    if (!mdir)
      w = append_seq(out, w, illu, illu_len, ov_hi, m->i_hi, 0);
    else
      w = append_seq(out, w, illu, illu_len, m->i_lo, ov_lo, 1);
    revcomp_inplace(out, w);
    return w;
  }
  if (!mdir) /* :408-417 */
    w = append_seq(out, w, illu, illu_len, m->i_lo, ov_lo, 0);
  else
    w = append_seq(out, w, illu, illu_len, ov_hi, m->i_hi, 1);
  w = append_seq(out, w, nano, nano_len, m->n_hi, (int)nanopore_length - 1, 1);
  return w;
}

/* getCorrectedNanoporeRange, ap.cpp:191-203 */
static void corrected_nanopore_range(const ms_row *m, int ov_lo, int ov_hi, double *first, double *second) {
  double l = (ov_lo - m->i_lo) / rratio(m);
  double r = (m->i_hi - ov_hi) / rratio(m);
  if (!(m->flags & MS_ROW_DIR)) {
    double t = l;
    l = r;
    r = t;
  }
  *first  = m->n_lo + l;
  *second = m->n_hi - r;
}

int ms_oracle_between_anchors(const ms_row *ml, const ms_row *mr, const char *nano, size_t nano_len,
                              const char *illu_l, size_t illu_l_len, const char *illu_r, size_t illu_r_len, int ovl_lo,
                              int ovl_hi, int ovr_lo, int ovr_hi, int direction, int *distance, char *out,
                              size_t *out_len) {
  const int    ldir = (ml->flags & MS_ROW_DIR) != 0, rdir = (mr->flags & MS_ROW_DIR) != 0;
  const double rl = rratio(ml), rr = rratio(mr);
  int          corr_l = 0, corr_r = 0;
  size_t       w = 0;
  *out_len = 0;
  if (!direction) { /* :458-518 */
    double err = mr->n_hi - ml->n_lo;
    if (err > 0) {
      double clf, cls, crf, crs;
      corrected_nanopore_range(ml, ovl_lo, ovl_hi, &clf, &cls);
      corrected_nanopore_range(mr, ovr_lo, ovr_hi, &crf, &crs);
      if (clf < crs) {
        *distance = (int)floor(clf - crs);
        return 0;
      }
      double avail_l, avail_r;
      if (!ldir) {
        avail_l = (ml->i_hi - ovl_hi) / rl;
        corr_l  = ml->i_hi - ovl_hi;
      } else {
        avail_l = (ovl_lo - ml->i_lo) / rl;
        corr_l  = ovl_lo - ml->i_lo;
      }
      if (avail_l > err) {
        corr_l = (int)floor(err * rl);
        err    = 0;
      } else {
        err -= avail_l;
      }
      if (!rdir) {
        avail_r = (ovr_lo - mr->i_lo) / rr;
        corr_r  = ovr_lo - mr->i_lo;
      } else {
        avail_r = (mr->i_hi - ovr_hi) / rr;
        corr_r  = mr->i_hi - ovr_hi;
      }
      if (avail_r > err) corr_r = (int)floor(err * rr);
    }
    if (!rdir)
      w = append_seq(out, w, illu_r, illu_r_len, mr->i_lo + corr_r, ovr_lo, 0);
    else
      w = append_seq(out, w, illu_r, illu_r_len, ovr_hi, mr->i_hi - corr_r, 1);
    w = append_seq(out, w, nano, nano_len, mr->n_hi, ml->n_lo, 1);
    if (!ldir)
      w = append_seq(out, w, illu_l, illu_l_len, ovl_hi, ml->i_hi - corr_l, 0);
    else
      w = append_seq(out, w, illu_l, illu_l_len, ml->i_lo + corr_l, ovl_lo, 1);
    *distance = (int)w;
    revcomp_inplace(out, w);
    *out_len = w;
    return 1;
  }
  double err = ml->n_hi - mr->n_lo; /* :520-578 */
  if (err > 0) {
    double clf, cls, crf, crs;
    corrected_nanopore_range(ml, ovl_lo, ovl_hi, &clf, &cls);
    corrected_nanopore_range(mr, ovr_lo, ovr_hi, &crf, &crs);
    if (cls > crf) {
      *distance = (int)floor(crf - cls);
      return 0;
    }
    double avail_l, avail_r;
    if (!ldir) {
      avail_l = (ovl_lo - ml->i_lo) / rl;
      corr_l  = ovl_lo - ml->i_lo;
    } else {
      avail_l = (ml->i_hi - ovl_hi) / rl;
      corr_l  = ml->i_hi - ovl_hi;
    }
    if (avail_l > err) {
      corr_l = (int)floor(err * rl);
      err    = 0;
    } else {
      err -= avail_l;
    }
    if (!rdir) {
      avail_r = (mr->i_hi - ovr_hi) / rr;
      corr_r  = mr->i_hi - ovr_hi;
    } else {
      avail_r = (ovr_lo - mr->i_lo) / rr;
      corr_r  = ovr_lo - mr->i_lo;
    }
    if (avail_r > err) corr_r = (int)floor(err * rr);
  }
  if (!ldir)
    w = append_seq(out, w, illu_l, illu_l_len, ml->i_lo + corr_l, ovl_lo, 0);
  else
    w = append_seq(out, w, illu_l, illu_l_len, ovl_hi, ml->i_hi - corr_l, 1);
  w = append_seq(out, w, nano, nano_len, ml->n_hi, mr->n_lo, 1);
  if (!rdir)
    w = append_seq(out, w, illu_r, illu_r_len, ovr_hi, mr->i_hi - corr_r, 0);
  else
    w = append_seq(out, w, illu_r, illu_r_len, mr->i_lo + corr_r, ovr_lo, 1);
  *distance = (int)w;
  *out_len  = w;
  return 1;
}

uint32_t ms_oracle_edit_distance(const char *a, size_t n, const char *b, size_t m, uint32_t band) {
  uint32_t *prev = (uint32_t *)xrealloc(NULL, (m + 1) * sizeof(uint32_t));
  uint32_t *cur  = (uint32_t *)xrealloc(NULL, (m + 1) * sizeof(uint32_t));
  for (size_t j = 0; j <= m; ++j) prev[j] = (uint32_t)j;
  for (size_t i = 1; i <= n; ++i) {
    cur[0] = (uint32_t)i;
    for (size_t j = 1; j <= m; ++j) {
      uint32_t d = prev[j - 1] + (a[i - 1] != b[j - 1] ? 1u : 0u);
      uint32_t u = prev[j] + 1u, l = cur[j - 1] + 1u;
      if (u < d) d = u;
      if (l < d) d = l;
      cur[j] = d;
    }
    uint32_t *t = prev;
    prev = cur;
    cur  = t;
  }
  uint32_t r = prev[m];
  free(prev);
  free(cur);
  return r > band ? band + 1 : r;
}

/* Same distance restricted to the band |i - j| <= band (cells outside count as "more than band"): exact whenever the
 * true distance is <= band, O(n * band) -- for the long contig-vs-genome checks of the tests. */
uint32_t ms_oracle_edit_distance_banded(const char *a, size_t n, const char *b, size_t m, uint32_t band) {
  const uint32_t INF = 0x3fffffffu;
  size_t diff = n > m ? n - m : m - n;
  if (diff > band) return band + 1;
  uint32_t *prev = (uint32_t *)xrealloc(NULL, (m + 2) * sizeof(uint32_t));
  uint32_t *cur  = (uint32_t *)xrealloc(NULL, (m + 2) * sizeof(uint32_t));
  for (size_t j = 0; j <= m; ++j) prev[j] = j <= band ? (uint32_t)j : INF;
  for (size_t i = 1; i <= n; ++i) {
    size_t lo = i > band ? i - band : 1, hi = i + band < m ? i + band : m;
    cur[lo - 1] = (lo == 1 && i <= band) ? (uint32_t)i : INF;
    for (size_t j = lo; j <= hi; ++j) {
      uint32_t d = prev[j - 1] + (a[i - 1] != b[j - 1] ? 1u : 0u);
      uint32_t u = (j <= i - 1 + band) ? prev[j] + 1u : INF; /* prev row holds columns up to (i - 1) + band */
      uint32_t l = cur[j - 1] + 1u;
      if (u < d) d = u;
      if (l < d) d = l;
      cur[j] = d > INF ? INF : d;
    }
    if (hi < m) cur[hi + 1] = INF;
    uint32_t *t = prev;
    prev = cur;
    cur  = t;
  }
  uint32_t r = prev[m];
  free(prev);
  free(cur);
  return r > band ? band + 1 : r;
}
