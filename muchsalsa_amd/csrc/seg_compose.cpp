// seg_compose.cpp -- the segment builders of assemblePath as copy-piece composers (host side of the consensus stage).
//
// Replaces getAnchorSequence, getSequenceLeftOfAnchor, getSequenceRightOfAnchor, getSequenceBetweenAnchors and
// getCorrectedNanoporeRange (libms/src/kernel/ap.cpp:191-203, 352-579).  The reference builds std::strings by fetching
// whole records from disk, slicing, appending and reverse-complementing.  Here each builder only DESCRIBES its result as
// 1-3 msgpu_copy pieces (source store, range, orientation, offset inside the segment); the bytes are produced later, for
// a whole layout at once, by the gather kernel (msgpu_gather_run).  The reverse complement of a concatenation
// rc(s1 s2 s3) = rc(s3) rc(s2) rc(s1) is expressed by reversing the piece order and toggling MSGPU_COPY_REVCOMP.
#include <cmath>

#include "asm_internal.h"
#include "msgpu.h"

namespace {

struct Composer {
  msgpu_seqctx *ctx;
  msgpu_copy   *out;
  uint32_t      n   = 0;
  int           err = MSGPU_OK;

  // getIlluminaSequence / getNanoporeSequence(id, left, right, direction) appended to the segment
  void add(int kind, uint32_t id, int32_t left, int32_t right, bool direction) {
    if (err != MSGPU_OK) return;
    err = msgpu_seq_resolve(ctx, kind, id, left, right, direction ? 1 : 0, &out[n]);
    if (err == MSGPU_OK) ++n;
  }
  // lay the pieces out back to back; `reverse` = getReverseComplement of the whole segment
  uint64_t finish(bool reverse) {
    if (reverse) {
      for (uint32_t i = 0; i < n / 2; ++i) {
        msgpu_copy t   = out[i];
        out[i]         = out[n - 1 - i];
        out[n - 1 - i] = t;
      }
      for (uint32_t i = 0; i < n; ++i) out[i].flags ^= MSGPU_COPY_REVCOMP;
    }
    uint64_t off = 0;
    for (uint32_t i = 0; i < n; ++i) {
      out[i].dst_off = off;
      off += out[i].len;
    }
    return off;
  }
};

inline bool mdir(const msgpu_row *m) { return (m->flags & MSGPU_ROW_DIR) != 0; }
inline double rratio(const msgpu_row *m) { // VertexMatch::rRatio, BlastFileReader.cpp:118-119
  return static_cast<double>(m->i_hi - m->i_lo + 1) / static_cast<double>(m->n_hi - m->n_lo + 1);
}
// getCorrectedNanoporeRange, ap.cpp:191-203
inline void corrected_range(const msgpu_row *m, int ov_lo, int ov_hi, double &first, double &second) {
  double l = (ov_lo - m->i_lo) / rratio(m);
  double r = (m->i_hi - ov_hi) / rratio(m);
  if (!mdir(m)) {
    double t = l;
    l        = r;
    r        = t;
  }
  first  = m->n_lo + l;
  second = m->n_hi - r;
}

constexpr int NANO = 0, ILLU = 1;

} // namespace

extern "C" {

// getAnchorSequence, ap.cpp:424-433
int msgpu_seg_anchor(msgpu_seqctx *ctx, const msgpu_row *m, int32_t ov_lo, int32_t ov_hi, int direction,
                     msgpu_copy *out, uint32_t *n_out, uint64_t *len) {
  if (!ctx || !m || !out || !n_out) return MSGPU_E_ARG;
  Composer c{ctx, out};
  c.add(ILLU, m->anchor_id, ov_lo, ov_hi, msgpu::toggle_mul(mdir(m), direction != 0));
  const uint64_t l = c.finish(false);
  *n_out           = c.n;
  if (len) *len = l;
  return c.err;
}

// getSequenceLeftOfAnchor, ap.cpp:352-386.  nanopore_length = Vertex::getNanoporeLength() of the read.
int msgpu_seg_left_of_anchor(msgpu_seqctx *ctx, const msgpu_row *m, uint64_t nanopore_length, int32_t ov_lo,
                             int32_t ov_hi, int direction, msgpu_copy *out, uint32_t *n_out, uint64_t *len) {
  if (!ctx || !m || !out || !n_out) return MSGPU_E_ARG;
  Composer c{ctx, out};
  uint64_t l;
  if (!direction) {
    if (!mdir(m))
      c.add(ILLU, m->anchor_id, m->i_lo, ov_lo, false);
    else
      c.add(ILLU, m->anchor_id, ov_hi, m->i_hi, true);
    c.add(NANO, m->read_id, m->n_hi, static_cast<int>(nanopore_length) - 1, true);
    l = c.finish(true);
  } else {
    c.add(NANO, m->read_id, 0, m->n_lo, true);
    if (!mdir(m))
      c.add(ILLU, m->anchor_id, ov_hi, m->i_hi, false);
    else
      c.add(ILLU, m->anchor_id, m->i_lo, ov_lo, true);
    l = c.finish(false);
  }
  *n_out = c.n;
  if (len) *len = l;
  return c.err;
}

// getSequenceRightOfAnchor, ap.cpp:388-422
int msgpu_seg_right_of_anchor(msgpu_seqctx *ctx, const msgpu_row *m, uint64_t nanopore_length, int32_t ov_lo,
                              int32_t ov_hi, int direction, msgpu_copy *out, uint32_t *n_out, uint64_t *len) {
  if (!ctx || !m || !out || !n_out) return MSGPU_E_ARG;
  Composer c{ctx, out};
  uint64_t l;
  if (!direction) {
    c.add(NANO, m->read_id, 0, m->n_lo, true);
    if (!mdir(m))
      c.add(ILLU, m->anchor_id, ov_hi, m->i_hi, false);
    else
      c.add(ILLU, m->anchor_id, m->i_lo, ov_lo, true);
    l = c.finish(true);
  } else {
    if (!mdir(m))
      c.add(ILLU, m->anchor_id, m->i_lo, ov_lo, false);
    else
      c.add(ILLU, m->anchor_id, ov_hi, m->i_hi, true);
    c.add(NANO, m->read_id, m->n_hi, static_cast<int>(nanopore_length) - 1, true);
    l = c.finish(false);
  }
  *n_out = c.n;
  if (len) *len = l;
  return c.err;
}

// getSequenceBetweenAnchors, ap.cpp:435-579.  *has_sequence = 0 mirrors std::nullopt (the corrected ranges overlap);
// *distance = std::get<0> of the reference's result (sequence length, or the floor of the negative gap).
int msgpu_seg_between_anchors(msgpu_seqctx *ctx, const msgpu_row *ml, const msgpu_row *mr, int32_t ovl_lo,
                              int32_t ovl_hi, int32_t ovr_lo, int32_t ovr_hi, int direction, msgpu_copy *out,
                              uint32_t *n_out, int32_t *distance, int *has_sequence) {
  if (!ctx || !ml || !mr || !out || !n_out || !distance || !has_sequence) return MSGPU_E_ARG;
  if (ml->read_id != mr->read_id) return MSGPU_E_ARG;
  const double rl = rratio(ml), rr = rratio(mr);
  int          corr_l = 0, corr_r = 0;
  *n_out        = 0;
  *has_sequence = 0;
  Composer c{ctx, out};
  if (!direction) {
    double err = mr->n_hi - ml->n_lo;
    if (err > 0) {
      double clf, cls, crf, crs;
      corrected_range(ml, ovl_lo, ovl_hi, clf, cls);
      corrected_range(mr, ovr_lo, ovr_hi, crf, crs);
      if (clf < crs) {
        *distance = static_cast<int>(std::floor(clf - crs));
        return MSGPU_OK;
      }
      double avail_l, avail_r;
      if (!mdir(ml)) {
        avail_l = (ml->i_hi - ovl_hi) / rl;
        corr_l  = ml->i_hi - ovl_hi;
      } else {
        avail_l = (ovl_lo - ml->i_lo) / rl;
        corr_l  = ovl_lo - ml->i_lo;
      }
      if (avail_l > err) {
        corr_l = static_cast<int>(std::floor(err * rl));
        err    = 0;
      } else {
        err -= avail_l;
      }
      if (!mdir(mr)) {
        avail_r = (ovr_lo - mr->i_lo) / rr;
        corr_r  = ovr_lo - mr->i_lo;
      } else {
        avail_r = (mr->i_hi - ovr_hi) / rr;
        corr_r  = mr->i_hi - ovr_hi;
      }
      if (avail_r > err) corr_r = static_cast<int>(std::floor(err * rr));
    }
    if (!mdir(mr))
      c.add(ILLU, mr->anchor_id, mr->i_lo + corr_r, ovr_lo, false);
    else
      c.add(ILLU, mr->anchor_id, ovr_hi, mr->i_hi - corr_r, true);
    c.add(NANO, ml->read_id, mr->n_hi, ml->n_lo, true);
    if (!mdir(ml))
      c.add(ILLU, ml->anchor_id, ovl_hi, ml->i_hi - corr_l, false);
    else
      c.add(ILLU, ml->anchor_id, ml->i_lo + corr_l, ovl_lo, true);
    *distance     = static_cast<int32_t>(c.finish(true));
    *n_out        = c.n;
    *has_sequence = 1;
    return c.err;
  }
  double err = ml->n_hi - mr->n_lo;
  if (err > 0) {
    double clf, cls, crf, crs;
    corrected_range(ml, ovl_lo, ovl_hi, clf, cls);
    corrected_range(mr, ovr_lo, ovr_hi, crf, crs);
    if (cls > crf) {
      *distance = static_cast<int>(std::floor(crf - cls));
      return MSGPU_OK;
    }
    double avail_l, avail_r;
    if (!mdir(ml)) {
      avail_l = (ovl_lo - ml->i_lo) / rl;
      corr_l  = ovl_lo - ml->i_lo;
    } else {
      avail_l = (ml->i_hi - ovl_hi) / rl;
      corr_l  = ml->i_hi - ovl_hi;
    }
    if (avail_l > err) {
      corr_l = static_cast<int>(std::floor(err * rl));
      err    = 0;
    } else {
      err -= avail_l;
    }
    if (!mdir(mr)) {
      avail_r = (mr->i_hi - ovr_hi) / rr;
      corr_r  = mr->i_hi - ovr_hi;
    } else {
      avail_r = (ovr_lo - mr->i_lo) / rr;
      corr_r  = ovr_lo - mr->i_lo;
    }
    if (avail_r > err) corr_r = static_cast<int>(std::floor(err * rr));
  }
  if (!mdir(ml))
    c.add(ILLU, ml->anchor_id, ml->i_lo + corr_l, ovl_lo, false);
  else
    c.add(ILLU, ml->anchor_id, ovl_hi, ml->i_hi - corr_l, true);
  c.add(NANO, ml->read_id, ml->n_hi, mr->n_lo, true);
  if (!mdir(mr))
    c.add(ILLU, mr->anchor_id, ovr_hi, mr->i_hi - corr_r, false);
  else
    c.add(ILLU, mr->anchor_id, mr->i_lo + corr_r, ovr_lo, true);
  *distance     = static_cast<int32_t>(c.finish(false));
  *n_out        = c.n;
  *has_sequence = 1;
  return c.err;
}

} // extern "C"
