// seq_loader.cpp -- host-side FASTA/FASTQ loader of libmsgpu (sequence store of the consensus stage).
//
// Replaces SequenceAccessor::buildIndex (_buildNanoporeIdx / _buildIlluminaIdx, libms/src/SequenceAccessor.cpp:143-231)
// and the per-call getSequenceFromFile (:54-69).  The reference keeps only file offsets and re-reads + re-strips the
// whole record from disk under a mutex for every slice; here the file is read once (mmap) and every record is kept
// whitespace-free, one byte per base, ready to be uploaded to HBM.  Behaviour kept:
//   * FASTQ iff the extension is neither "fa" nor "fasta" (isFastQ, :71-80); description '>' / '@', split '>' / '+';
//   * record id = description line without its first character, cut at the first whitespace (cleanSequenceId, :82-87);
//   * a record's bytes are all lines up to the next description (FASTA) or '+' (FASTQ) line; in FASTQ everything up
//     to the next line starting with '@' is skipped (:181-184) -- including the quirk that a quality line starting
//     with '@' is taken for a description line;
//   * std::isspace characters are removed, the record is cut at an embedded NUL (std::string(buffer.data()), :65-67);
//   * of two records with the same id the first one wins (unordered_map::emplace, :171).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <string_view>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "msgpu.h"

struct msgpu_seqfile {
  std::vector<std::string> names; // cleaned ids, first-occurrence order
  std::vector<uint64_t>    off;   // names.size() + 1 offsets into bases
  std::string              bases;
};

namespace {

struct LineReader { // readline (libms/src/IO.cpp:54-97) over a memory image
  const char *buf;
  size_t      len, pos = 0;
  const char *line = "";
  size_t      line_len = 0;
  long next() {
    if (pos >= len) return -1;
    const void *nl = memchr(buf + pos, '\n', len - pos);
    size_t      e  = nl ? static_cast<size_t>(static_cast<const char *>(nl) - buf) + 1 : len;
    line           = buf + pos;
    line_len       = e - pos;
    pos            = e;
    return static_cast<long>(line_len);
  }
};

// 1: std::isspace in the "C" locale (removed), 2: NUL (cuts the record), 0: kept
struct BlankTable {
  unsigned char t[256] = {0};
  constexpr BlankTable() {
    t[static_cast<unsigned char>(' ')] = t[static_cast<unsigned char>('\t')] = t[static_cast<unsigned char>('\n')] = 1;
    t[static_cast<unsigned char>('\v')] = t[static_cast<unsigned char>('\f')] = t[static_cast<unsigned char>('\r')] = 1;
    t[0] = 2;
  }
  constexpr unsigned char operator[](unsigned char c) const { return t[c]; }
};
constexpr BlankTable BLANK{};

// any byte < 0x21 among n bytes (8 at a time): every std::isspace character and NUL is below 0x21
inline bool has_byte_below_0x21(const char *p, size_t n) {
  size_t k = 0;
  for (; k + 8 <= n; k += 8) {
    uint64_t x;
    std::memcpy(&x, p + k, 8);
    if ((x - 0x2121212121212121ull) & ~x & 0x8080808080808080ull) return true;
  }
  for (; k < n; ++k)
    if (static_cast<unsigned char>(p[k]) < 0x21) return true;
  return false;
}

struct ChunkRecords { // the records of one stretch of the file: off has names.size() + 1 entries into bases
  std::vector<std::string> names;
  std::vector<uint64_t>    off{0};
  std::string              bases;
};

// The record loop of SequenceAccessor::_build*Idx + getSequenceFromFile (SequenceAccessor.cpp:54-69, 143-231) over
// [begin, len): description lines at or behind `limit` are not taken.  -> the position of the description line it
// stopped at (len at the end of the file).  Records with an id this stretch has seen already are dropped (first wins).
size_t parse_range(const char *data, size_t len, size_t begin, size_t limit, char desc, char split, ChunkRecords &out) {
  out.bases.reserve((limit > begin ? limit - begin : 0) + 16);
  std::unordered_map<std::string, uint32_t> ids;
  LineReader rl{data, len};
  rl.pos   = begin;
  long ret = rl.next();
  while (ret != -1 && rl.line[0] != desc) ret = rl.next(); // :149-158
  while (ret != -1 && rl.line[0] == desc) {                 // :160 (a last line starting with the description character would loop forever in the reference; EOF ends it here)
    const size_t at = static_cast<size_t>(rl.line - data);
    if (at >= limit) return at;
    size_t idl = 0;
    while (1 + idl < rl.line_len && !std::isspace(static_cast<unsigned char>(rl.line[1 + idl]))) ++idl;
    std::string id(rl.line + 1, idl);
    const bool  is_new = ids.emplace(id, static_cast<uint32_t>(out.names.size())).second;
    if (is_new) out.names.push_back(std::move(id));
    bool cut = false; // an embedded NUL ends the record (std::string(buffer.data()), :65-67)
    while (true) {    // :167-179
      ret = rl.next();
      if (ret == -1 || rl.line[0] == split) break;
      if (!is_new || cut) continue;
      // strip std::isspace characters: the usual line is bases + '\n' (or "\r\n"), so trim the tail and copy the
      // rest with one memcpy when no other blank or NUL is inside; fall back to the byte loop otherwise
      const char *p = rl.line;
      size_t      n = rl.line_len;
      while (n && BLANK[static_cast<unsigned char>(p[n - 1])] == 1) --n;
      if (!has_byte_below_0x21(p, n)) { // no blank, no NUL, no control byte inside
        out.bases.append(p, n);
      } else {
        for (size_t q = 0; q < n; ++q) {
          const unsigned char ch = static_cast<unsigned char>(p[q]);
          if (ch == 0) {
            cut = true;
            break;
          }
          if (BLANK[ch] != 1) out.bases.push_back(static_cast<char>(ch));
        }
      }
    }
    if (is_new) out.off.push_back(out.bases.size());
    while (ret != -1 && rl.line[0] != desc) ret = rl.next(); // :181-184
  }
  return len;
}

bool is_fastq_name(const char *path) {
  std::string_view p(path);
  size_t           dot = p.find_last_of('.');
  std::string      ext(p.substr(dot == std::string_view::npos ? 0 : dot + 1));
  for (auto &c : ext) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return ext != "fa" && ext != "fasta";
}

} // namespace

extern "C" {

int msgpu_seq_parse(const char *path, int is_fastq, msgpu_seqfile **out) {
  if (!path || !out) return MSGPU_E_ARG;
  *out   = nullptr;
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return MSGPU_E_IO; // "Can't open sequence file(s)." SequenceAccessor.cpp:105-107
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
  if (is_fastq < 0) is_fastq = is_fastq_name(path) ? 1 : 0;
  const char desc = is_fastq ? '@' : '>', split = is_fastq ? '+' : '>';

  std::unique_ptr<msgpu_seqfile> f;
  try {
    f = std::make_unique<msgpu_seqfile>();
    // Large files are parsed in chunks on several threads.  A chunk starts at a line that begins with the description
    // character and is parsed by the SAME sequential state machine; it stops at the first description line it would take
    // at or behind the next chunk's start.  The cut is right iff that line IS the next chunk's start (in FASTQ a quality
    // line may begin with '@', and which '@' lines are descriptions depends on everything before them): checked for
    // every chunk, and any mismatch sends the whole file through one sequential pass.
    unsigned nt = std::thread::hardware_concurrency();
    nt          = nt == 0 ? 1 : (nt > 8 ? 8 : nt); // (two files are parsed side by side, beside the PAF parser)
    if (const char *e = std::getenv("MSGPU_SEQ_THREADS")) nt = static_cast<unsigned>(std::max(1, std::atoi(e))); // (tests: any size)
    else if (len < (size_t(32) << 20)) nt = 1;
    std::vector<size_t> starts{0};
    for (unsigned k = 1; k < nt; ++k) {
      size_t p = len / nt * k;
      while (true) { // the next line start whose first character is the description character
        const void *nl = p < len ? memchr(data + p, '\n', len - p) : nullptr;
        if (!nl) {
          p = len;
          break;
        }
        p = static_cast<size_t>(static_cast<const char *>(nl) - data) + 1;
        if (p < len && data[p] == desc) break;
      }
      if (p < len && p > starts.back()) starts.push_back(p);
    }
    const unsigned            nc = static_cast<unsigned>(starts.size());
    std::vector<ChunkRecords> parts(nc);
    std::vector<size_t>       stops(nc, 0);
    std::vector<int>          oom(nc, 0);
    auto                      work = [&](unsigned k) {
      try {
        stops[k] = parse_range(data, len, starts[k], k + 1 < nc ? starts[k + 1] : len, desc, split, parts[k]);
      } catch (std::bad_alloc const &) { oom[k] = 1; }
    };
    {
      std::vector<std::thread> pool;
      for (unsigned k = 1; k < nc; ++k) pool.emplace_back(work, k);
      work(0);
      for (auto &t : pool) t.join();
    }
    for (int o : oom)
      if (o) throw std::bad_alloc();
    bool consistent = true;
    for (unsigned k = 0; k + 1 < nc; ++k) consistent = consistent && stops[k] == starts[k + 1];
    if (!consistent) { // (only a FASTQ whose quality lines start with '@' in unlucky places gets here)
      parts.assign(1, ChunkRecords());
      parse_range(data, len, 0, len, desc, split, parts[0]);
    }
    // merge in file order: of two records with the same id the first one wins (unordered_map::emplace, :171)
    std::unordered_map<std::string_view, uint32_t> ids;
    std::vector<std::pair<uint32_t, uint32_t>>     kept; // (chunk, record) of every record that stays
    uint64_t                                       total = 0;
    f->off.push_back(0);
    for (uint32_t c = 0; c < parts.size(); ++c)
      for (uint32_t r = 0; r < parts[c].names.size(); ++r) {
        if (parts.size() > 1 && !ids.emplace(std::string_view(parts[c].names[r]), 0u).second) continue;
        kept.emplace_back(c, r);
        total += parts[c].off[r + 1] - parts[c].off[r];
        f->off.push_back(total);
      }
    if (parts.size() == 1) {
      f->names = std::move(parts[0].names);
      f->bases = std::move(parts[0].bases);
    } else {
      f->bases.resize(total);
      f->names.resize(kept.size());
      const size_t             nk = kept.size();
      std::vector<std::thread> pool;
      auto                     copy = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
          ChunkRecords &p = parts[kept[i].first];
          const uint32_t r = kept[i].second;
          f->names[i]      = std::move(p.names[r]);
          memcpy(&f->bases[f->off[i]], p.bases.data() + p.off[r], p.off[r + 1] - p.off[r]);
        }
      };
      ids.clear(); // (its keys view the names that are about to move)
      for (unsigned t = 1; t < nt; ++t) pool.emplace_back(copy, nk * t / nt, nk * (t + 1) / nt);
      copy(0, nk / nt);
      for (auto &t : pool) t.join();
    }
  } catch (std::bad_alloc const &) {
    if (data) munmap(const_cast<char *>(data), len);
    return MSGPU_E_NOMEM;
  } catch (std::system_error const &) { // could not start a thread
    if (data) munmap(const_cast<char *>(data), len);
    return MSGPU_E_NOMEM;
  }
  if (data) munmap(const_cast<char *>(data), len);
  *out = f.release();
  return MSGPU_OK;
}

void msgpu_seq_free(msgpu_seqfile *f) { delete f; }
uint32_t msgpu_seq_count(const msgpu_seqfile *f) { return f ? static_cast<uint32_t>(f->names.size()) : 0; }
const char *msgpu_seq_name(const msgpu_seqfile *f, uint32_t i) {
  return f && i < f->names.size() ? f->names[i].c_str() : nullptr;
}
uint64_t msgpu_seq_length(const msgpu_seqfile *f, uint32_t i) {
  return f && i < f->names.size() ? f->off[i + 1] - f->off[i] : 0;
}
const char *msgpu_seq_bases(const msgpu_seqfile *f, uint32_t i) {
  return f && i < f->names.size() ? f->bases.data() + f->off[i] : nullptr;
}

// strSlice (libms/src/SequenceUtils.cpp:27-38) as (offset, length): Python-like indices, INCLUSIVE clipped end.
uint64_t msgpu_str_slice(uint64_t size, int32_t start, int32_t end, uint64_t *len) {
  const long long sz = static_cast<long long>(size);
  const long long i  = start >= 0 ? start : sz + start;
  const long long j  = end >= 0 ? end : sz + end;
  const uint64_t  s  = static_cast<uint64_t>(i > 0 ? i : 0);
  uint64_t        e  = static_cast<uint64_t>(j > 0 ? j : 0);
  if (e > size) e = size;
  const uint64_t ic = static_cast<uint64_t>(i); // static_cast<std::size_t>(i): a negative i wraps
  if (e < ic) e = ic;
  if (s > size) { // substr would throw std::out_of_range
    if (len) *len = 0;
    return size;
  }
  uint64_t cnt = e - s + 1;
  if (cnt > size - s) cnt = size - s;
  if (len) *len = cnt;
  return s;
}

} // extern "C"
