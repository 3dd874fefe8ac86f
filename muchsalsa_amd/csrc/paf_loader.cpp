// paf_loader.cpp -- host-side PAF loader of libmsgpu (A1 of SURVEY.md section 8).
//
// Replaces BlastFileAccessor::_buildIndex (libms/src/BlastFileAccessor.cpp:77-91) and
// BlastFileReader::read/parseLine (libms/src/BlastFileReader.cpp:72-130) with one pass over an mmap'ed file:
// no per-line job, no file mutex, no istringstream.  Behaviour kept:
//   * every '\n'-terminated line plus a non-empty unterminated tail is a line (libms/src/IO.cpp:54-97);
//   * the LAST line is never parsed (BlastFileReader.cpp:76);
//   * a line is kept iff col9 >= MINIMUM_MATCHES and col3-col2 >= MINIMUM_MATCHES (:106-107);
//   * Registry ids in first-seen order, nanopore id registered before the illumina id (:110-111);
//   * ranges are inclusive (col3-1, col8-1), isPrimary = span >= TH_LENGTH && matches >= TH_MATCHES (:121-122).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <climits>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "msgpu.h"

namespace {

// Registry (libms/src/Registry.cpp:36-45): name -> dense id in first-seen order, plus the reverse table.
// Names live in individually heap-allocated strings, so the string_view keys stay valid as the table grows.
class NameRegistry {
public:
  uint32_t operator[](std::string_view name) {
    auto it = m_ids.find(name);
    if (it != m_ids.end()) return it->second;
    const auto id = static_cast<uint32_t>(m_names.size());
    m_names.push_back(std::make_unique<std::string>(name));
    m_ids.emplace(std::string_view(*m_names.back()), id);
    return id;
  }
  uint32_t    size() const { return static_cast<uint32_t>(m_names.size()); }
  const char *name(uint32_t id) const { return id < m_names.size() ? m_names[id]->c_str() : nullptr; }

private:
  std::vector<std::unique_ptr<std::string>>      m_names;
  std::unordered_map<std::string_view, uint32_t> m_ids;
};

// std::stoi: optional whitespace, optional sign, at least one digit, value must fit int.
bool parse_int(const char *s, const char *e, int &out) {
  while (s < e && (*s == ' ' || (*s >= '\t' && *s <= '\r'))) ++s;
  bool neg = false;
  if (s < e && (*s == '+' || *s == '-')) neg = *s++ == '-';
  if (s >= e || *s < '0' || *s > '9') return false;
  long long v = 0;
  for (; s < e && *s >= '0' && *s <= '9'; ++s) {
    v = v * 10 + (*s - '0');
    if (v > static_cast<long long>(INT_MAX) + 1) return false;
  }
  if (neg) v = -v;
  if (v > INT_MAX || v < INT_MIN) return false;
  out = static_cast<int>(v);
  return true;
}

} // namespace

struct msgpu_paf {
  std::vector<msgpu_row> rows;
  size_t                 n_lines = 0;
  NameRegistry           reads, anchors;
};

extern "C" {

int msgpu_parse_paf(const char *path, const msgpu_params *params, msgpu_paf **out) {
  if (!path || !out) return MSGPU_E_ARG;
  *out = nullptr;
  msgpu_params p;
  if (params)
    p = *params;
  else
    msgpu_default_params(&p);

  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return MSGPU_E_IO;
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return MSGPU_E_IO;
  }
  size_t      len  = static_cast<size_t>(st.st_size);
  const char *data = nullptr;
  if (len) {
    void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      return MSGPU_E_IO;
    }
    madvise(m, len, MADV_SEQUENTIAL);
    data = static_cast<const char *>(m);
  }
  close(fd);

  msgpu_paf *paf = nullptr;
  int        rc  = MSGPU_OK;
  try { // no exception may cross the C-ABI
  paf = new msgpu_paf();
  // count lines first: the parser must know which line is the last one
  size_t n_lines = 0;
  for (const char *q = data, *end = data + len; q < end;) {
    ++n_lines;
    const void *nl = memchr(q, '\n', static_cast<size_t>(end - q));
    q              = nl ? static_cast<const char *>(nl) + 1 : end;
  }
  paf->n_lines = n_lines;
  paf->rows.reserve(n_lines);

  const char *q = data, *end = data + len;
  for (size_t li = 0; li + 1 < n_lines; ++li) {
    const void *nlp = memchr(q, '\n', static_cast<size_t>(end - q));
    const char *le  = nlp ? static_cast<const char *>(nlp) : end;
    // columns 0..9; std::getline drops an empty token after the final delimiter
    const char *tb[10], *te[10];
    int         nt = 0;
    const char *s  = q;
    for (const char *c = q;; ++c) {
      if (c == le || *c == '\t') {
        if (c == le && s == le) break;
        if (nt < 10) {
          tb[nt] = s;
          te[nt] = c;
        }
        ++nt;
        s = c + 1;
        if (c == le) break;
      }
    }
    q = nlp ? le + 1 : end;
    // the reference checks for >= 9 tokens and then reads tokens[9]; we ask for the 10 it needs
    if (nt < 10) {
      rc = MSGPU_E_FORMAT;
      break;
    }
    int irs, ire, nom, nle;
    if (!parse_int(tb[2], te[2], irs) || !parse_int(tb[3], te[3], ire) || !parse_int(tb[9], te[9], nom) ||
        !parse_int(tb[6], te[6], nle) || nom < 0) {
      rc = MSGPU_E_NUMBER;
      break;
    }
    const int span = (ire - 1) - irs + 1;
    if (!(static_cast<uint32_t>(nom) >= p.min_matches && span >= static_cast<int>(p.min_matches))) continue;
    msgpu_row r;
    r.read_id   = paf->reads[std::string_view(tb[5], static_cast<size_t>(te[5] - tb[5]))];
    r.anchor_id = paf->anchors[std::string_view(tb[0], static_cast<size_t>(te[0] - tb[0]))];
    int nrs, nre;
    if (!parse_int(tb[7], te[7], nrs) || !parse_int(tb[8], te[8], nre)) {
      rc = MSGPU_E_NUMBER;
      break;
    }
    const bool dir  = (te[4] - tb[4] == 1) && *tb[4] == '+';
    const bool prim = span >= static_cast<int>(p.th_length) && static_cast<uint32_t>(nom) >= p.th_matches;
    r.read_len = nle;
    r.i_lo     = irs;
    r.i_hi     = ire - 1;
    r.n_lo     = nrs;
    r.n_hi     = nre - 1;
    r.score    = static_cast<uint32_t>(nom);
    r.line     = static_cast<uint32_t>(li);
    r.flags    = (dir ? MSGPU_ROW_DIR : 0u) | (prim ? MSGPU_ROW_PRIMARY : 0u);
    paf->rows.push_back(r);
  }
  } catch (std::bad_alloc const &) {
    rc = MSGPU_E_NOMEM;
  }
  if (data) munmap(const_cast<char *>(data), len);
  if (rc != MSGPU_OK) {
    delete paf;
    return rc;
  }
  *out = paf;
  return MSGPU_OK;
}

void msgpu_paf_free(msgpu_paf *paf) { delete paf; }

const msgpu_row *msgpu_paf_rows(const msgpu_paf *paf, size_t *n_rows) {
  if (n_rows) *n_rows = paf ? paf->rows.size() : 0;
  return paf && !paf->rows.empty() ? paf->rows.data() : nullptr;
}
size_t      msgpu_paf_line_count(const msgpu_paf *paf) { return paf ? paf->n_lines : 0; }
uint32_t    msgpu_paf_read_count(const msgpu_paf *paf) { return paf ? paf->reads.size() : 0; }
uint32_t    msgpu_paf_anchor_count(const msgpu_paf *paf) { return paf ? paf->anchors.size() : 0; }
const char *msgpu_paf_read_name(const msgpu_paf *paf, uint32_t id) { return paf ? paf->reads.name(id) : nullptr; }
const char *msgpu_paf_anchor_name(const msgpu_paf *paf, uint32_t id) { return paf ? paf->anchors.name(id) : nullptr; }

} // extern "C"
