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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <string_view>
#include <system_error>
#include <thread>
#include <vector>

#include "asm_internal.h"
#include "msgpu.h"

// The records' bytes live in ONE anonymous mapping as large as the file: a stretch of the file is stripped into the same
// stretch of the mapping (stripping only removes bytes), so the parser threads write their records where they stay -- no
// per-thread buffers, no merge copy, and the pages are first touched by the thread that fills them.  Between the records
// lie the bytes the description lines, line ends and dropped duplicates took in the file.
struct BaseBuffer {
  char  *p   = nullptr; // 2 MiB aligned
  size_t cap = 0;
  BaseBuffer() = default;
  BaseBuffer(const BaseBuffer &) = delete;
  BaseBuffer &operator=(const BaseBuffer &) = delete;
  ~BaseBuffer() {
    // in slices: giving back a gigabyte of small pages takes the address-space lock for tens of milliseconds, and every
    // other thread that maps or pins memory meanwhile (the HIP runtime does) would wait for all of it
    constexpr size_t SLICE = size_t(8) << 20;
    for (size_t o = 0; o < m_len; o += SLICE) munmap(m_raw + o, std::min(SLICE, m_len - o));
  }
  void allocate(size_t bytes) {
    if (!bytes) return;
    constexpr size_t HUGE = size_t(2) << 20;
    const size_t     want = ((bytes + HUGE - 1) & ~(HUGE - 1)) + HUGE;
    void            *m    = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    m_raw = static_cast<char *>(m);
    m_len = want;
    p     = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(m_raw) + HUGE - 1) & ~(uintptr_t(HUGE) - 1));
    cap   = bytes;
    madvise(p, want - HUGE, MADV_HUGEPAGE); // 1.5 GB of first touches: 2 MiB pages where the kernel grants them
  }

private:
  char  *m_raw = nullptr;
  size_t m_len = 0;
};

struct msgpu_seqfile {
  std::vector<std::string> names; // cleaned ids, first-occurrence order
  std::vector<uint64_t>    off;   // per record: offset of its bytes in the buffer
  std::vector<uint64_t>    len;   // per record: its length
  BaseBuffer               bases;
  uint64_t                 extent = 0; // end of the last record in the buffer
};

namespace {

struct JoinAll { // a thread that could not be started must not leave the started ones unjoined behind an exception
  std::vector<std::thread> &t;
  ~JoinAll() {
    for (auto &x : t)
      if (x.joinable()) x.join();
  }
};

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

struct Record { // one record of a stretch of the file; the name lies in the stretch's own name block
  const char *name;
  uint32_t    name_len;
  uint32_t    duplicate; // set by the first-wins pass: an earlier record has this id
  uint64_t    hash;      // FNV-1a of the name
  uint64_t    off, len;  // the record's bytes in the file-sized buffer
};
struct ChunkRecords { // the records of one stretch of the file, in the order of the file
  std::vector<Record> recs;
  std::string         names; // the ids one after the other (the file image is given back as soon as a stretch is parsed)
  size_t              end = 0; // behind the last byte written (the records of a stretch lie back to back from its start)
  void                bind_names() { // once nothing is appended any more: Record::name = offset -> pointer
    for (Record &r : recs) r.name = names.data() + reinterpret_cast<uintptr_t>(r.name);
  }
};

inline uint64_t name_hash(const char *s, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= static_cast<unsigned char>(s[i]);
    h *= 1099511628211ull;
  }
  return h;
}

constexpr size_t CUT_CROSSED = ~size_t(0); // parse_range: a record ran over the stretch's limit (the cut was not a record boundary)

// The record loop of SequenceAccessor::_build*Idx + getSequenceFromFile (SequenceAccessor.cpp:54-69, 143-231) over
// [begin, len): description lines at or behind `limit` are not taken.  -> the position of the description line it
// stopped at (len at the end of the file), CUT_CROSSED when a record's lines reach over `limit` (nothing is written behind
// `limit`).  The records' bytes go to dst at positions back to back from `begin` on: never ahead of the bytes they are read from.
// Which of two records with one id stays (the first, unordered_map::emplace, :171) is decided afterwards, over the whole
// file: a later duplicate's bytes lie unused between its neighbours.
size_t parse_range(const char *data, size_t len, size_t begin, size_t limit, char desc, char split, msgpu::ByteSink &dst,
                   ChunkRecords &out) {
  LineReader rl{data, len};
  rl.pos   = begin;
  long ret = rl.next();
  while (ret != -1 && rl.line[0] != desc) ret = rl.next(); // :149-158
  size_t w = begin;                                         // write cursor: <= the read position at any time
  out.end  = begin;
  while (ret != -1 && rl.line[0] == desc) {                 // :160 (a last line starting with the description character would loop forever in the reference; EOF ends it here)
    const size_t at = static_cast<size_t>(rl.line - data);
    if (at >= limit) return at;
    size_t idl = 0;
    while (1 + idl < rl.line_len && !std::isspace(static_cast<unsigned char>(rl.line[1 + idl]))) ++idl;
    Record r;
    r.name      = reinterpret_cast<const char *>(static_cast<uintptr_t>(out.names.size())); // (offset until bind_names)
    r.name_len  = static_cast<uint32_t>(idl);
    r.duplicate = 0;
    r.hash      = name_hash(rl.line + 1, idl);
    out.names.append(rl.line + 1, idl);
    r.off       = w;
    bool cut    = false; // an embedded NUL ends the record (std::string(buffer.data()), :65-67)
    while (true) {       // :167-179
      ret = rl.next();
      if (ret == -1 || rl.line[0] == split) break;
      if (static_cast<size_t>(rl.line - data) >= limit) return CUT_CROSSED;
      if (cut) continue;
      // strip std::isspace characters: the usual line is bases + '\n' (or "\r\n"), so trim the tail and copy the
      // rest with one memcpy when no other blank or NUL is inside; fall back to the byte loop otherwise
      const char *p = rl.line;
      size_t      n = rl.line_len;
      while (n && BLANK[static_cast<unsigned char>(p[n - 1])] == 1) --n;
      if (!has_byte_below_0x21(p, n)) { // no blank, no NUL, no control byte inside
        dst.put(w, p, n);
        w += n;
      } else {
        for (size_t q = 0; q < n; ++q) {
          const unsigned char ch = static_cast<unsigned char>(p[q]);
          if (ch == 0) {
            cut = true;
            break;
          }
          if (BLANK[ch] != 1) {
            const char one = static_cast<char>(ch);
            dst.put(w++, &one, 1);
          }
        }
      }
    }
    r.len = w - r.off;
    out.recs.push_back(r);
    out.end = w;
    while (ret != -1 && rl.line[0] != desc) ret = rl.next(); // :181-184
  }
  return len;
}

// "of two records with the same id the first one wins" over all stretches, on nt threads: thread t owns the ids whose hash
// falls into its share, walks every stretch in file order and keeps its ids in an open-addressing table of its own.
void mark_duplicates(std::vector<ChunkRecords> &parts, unsigned nt) {
  size_t n_all = 0;
  for (auto &c : parts) n_all += c.recs.size();
  if (nt > 1 && n_all < 4096) nt = 1;
  auto share = [&](unsigned t) {
    size_t cap = 1024;
    while (cap < 4 * (n_all / nt + 1)) cap <<= 1;
    std::vector<Record *> slot(cap, nullptr);
    size_t                used = 0;
    for (auto &c : parts)
      for (Record &r : c.recs) {
        if (nt > 1 && static_cast<unsigned>((r.hash >> 40) % nt) != t) continue;
        if ((used + 1) * 2 > cap) { // (only a very uneven hash gets here)
          std::vector<Record *> ns(cap * 2, nullptr);
          for (Record *x : slot)
            if (x) {
              size_t i = x->hash & (cap * 2 - 1);
              while (ns[i]) i = (i + 1) & (cap * 2 - 1);
              ns[i] = x;
            }
          slot.swap(ns);
          cap *= 2;
        }
        size_t i = r.hash & (cap - 1);
        for (; slot[i]; i = (i + 1) & (cap - 1)) {
          const Record *x = slot[i];
          if (x->hash == r.hash && x->name_len == r.name_len && memcmp(x->name, r.name, r.name_len) == 0) break;
        }
        if (slot[i]) {
          r.duplicate = 1;
        } else {
          slot[i] = &r;
          ++used;
        }
      }
  };
  if (nt <= 1) {
    nt = 1;
    share(0);
    return;
  }
  std::vector<std::thread> pool;
  JoinAll                  join_all{pool};
  std::vector<int>         oom(nt, 0);
  auto                     guarded = [&](unsigned t) {
    try {
      share(t);
    } catch (std::bad_alloc const &) { oom[t] = 1; }
  };
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(guarded, t);
  guarded(0);
  for (auto &t : pool) t.join();
  for (int o : oom)
    if (o) throw std::bad_alloc();
}

struct HostDestination final : msgpu::SeqDestination { // the loader's own buffer, as large as the file
  struct Sink final : msgpu::ByteSink {
    char *base = nullptr;
    void  put(uint64_t at, const char *p, size_t n) override { memcpy(base + at, p, n); }
  } sink;
  BaseBuffer &buf;
  explicit HostDestination(BaseBuffer &b) : buf(b) {}
  void prepare(uint64_t file_bytes, unsigned) override {
    buf.allocate(file_bytes);
    sink.base = buf.p;
  }
  msgpu::ByteSink *stretch(unsigned) override { return &sink; } // (positions of different stretches do not meet)
  void             restart() override {}
  void             fill(uint64_t from, uint64_t to) override { memset(buf.p + from, 'A', to - from); }
  int              finish(uint64_t) override { return MSGPU_OK; }
};

bool is_fastq_name(const char *path) {
  std::string_view p(path);
  size_t           dot = p.find_last_of('.');
  std::string      ext(p.substr(dot == std::string_view::npos ? 0 : dot + 1));
  for (auto &c : ext) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return ext != "fa" && ext != "fasta";
}

} // namespace

int msgpu::seq_parse_into(const char *path, int is_fastq, msgpu::SeqDestination *dest, msgpu_seqfile **out) {
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

  const bool dbg = std::getenv("MSGPU_PARSE_DEBUG") != nullptr;
  auto       t0  = std::chrono::steady_clock::now();
  auto       lap = [&](const char *what) {
    if (!dbg) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "msgpu_seq_parse %s: %-8s %.1f ms\n", path, what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
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
    HostDestination host(f->bases);
    if (!dest) dest = &host;
    lap("cuts");
    dest->prepare(len, nc);
    lap("prepare");
    auto work = [&](unsigned k) {
      try {
        msgpu::ByteSink &sink = *dest->stretch(k);
        const size_t     lim  = k + 1 < nc ? starts[k + 1] : len;
        stops[k] = parse_range(data, len, starts[k], lim, desc, split, sink, parts[k]);
        sink.done();
        parts[k].bind_names();
        // the pages of this stretch leave the mapping here, on this thread: taking a gigabyte of page-cache pages out of
        // the address space is tens of milliseconds of work, done on one thread if it is left to the final munmap
        const size_t page = static_cast<size_t>(sysconf(_SC_PAGESIZE));
        const size_t lo = (starts[k] + page - 1) / page * page, hi = lim / page * page;
        if (hi > lo) madvise(const_cast<char *>(data) + lo, hi - lo, MADV_DONTNEED);
      } catch (std::bad_alloc const &) { oom[k] = 1; }
    };
    {
      std::vector<std::thread> pool;
      JoinAll                  join_all{pool};
      for (unsigned k = 1; k < nc; ++k) pool.emplace_back(work, k);
      work(0);
      for (auto &t : pool) t.join();
    }
    for (int o : oom)
      if (o) throw std::bad_alloc();
    lap("records");
    bool consistent = true;
    for (unsigned k = 0; k + 1 < nc; ++k) consistent = consistent && stops[k] == starts[k + 1];
    if (!consistent) { // (only a FASTQ whose quality lines start with '@' in unlucky places gets here)
      if (dbg) fprintf(stderr, "msgpu_seq_parse %s: the cuts did not verify, one pass over the file\n", path);
      parts.assign(1, ChunkRecords());
      dest->restart();
      msgpu::ByteSink &sink = *dest->stretch(0);
      parse_range(data, len, 0, len, desc, split, sink, parts[0]);
      sink.done();
      parts[0].bind_names();
    }
    // the records in file order, first occurrence of an id only (unordered_map::emplace, :171)
    mark_duplicates(parts, nt);
    lap("first-wins");
    std::vector<size_t> base(parts.size() + 1, 0);
    for (size_t k = 0; k < parts.size(); ++k) {
      size_t kept = 0;
      for (const Record &r : parts[k].recs) kept += !r.duplicate;
      base[k + 1] = base[k] + kept;
    }
    f->names.resize(base.back());
    f->off.resize(base.back());
    f->len.resize(base.back());
    auto fill = [&](size_t k) {
      size_t i = base[k];
      for (const Record &r : parts[k].recs) {
        if (r.duplicate) continue;
        f->names[i].assign(r.name, r.name_len);
        f->off[i] = r.off;
        f->len[i] = r.len;
        ++i;
      }
    };
    {
      std::vector<std::thread> pool;
      JoinAll                  join_all{pool};
      std::vector<int>         bad(parts.size(), 0);
      auto                     guarded = [&](size_t k) {
        try {
          fill(k);
        } catch (std::bad_alloc const &) { bad[k] = 1; }
      };
      for (size_t k = 1; k < parts.size(); ++k) pool.emplace_back(guarded, k);
      guarded(0);
      for (auto &t : pool) t.join();
      for (int o : bad)
        if (o) throw std::bad_alloc();
    }
    // what the description lines and line ends of a stretch took lies unused between its last record and the next
    // stretch: filled with a base, so that the 2-bit form of the store (msgpu_seq_pack) sees no exceptional bytes there
    for (size_t k = 0; k + 1 < parts.size(); ++k)
      if (starts[k + 1] > parts[k].end) dest->fill(parts[k].end, starts[k + 1]);
    f->extent = parts.empty() ? 0 : parts.back().end;
    const int drc = dest->finish(f->extent);
    lap("finish");
    if (drc != MSGPU_OK) {
      if (data) munmap(const_cast<char *>(data), len);
      return drc;
    }
    lap("merge");
    if (dbg) { // how much of the process' anonymous memory sits in 2 MiB pages right now
      if (FILE *fp = fopen("/proc/self/smaps_rollup", "r")) {
        char line[256];
        while (fgets(line, sizeof line, fp))
          if (!strncmp(line, "AnonHugePages:", 14) || !strncmp(line, "Anonymous:", 10)) fprintf(stderr, "  %s", line);
        fclose(fp);
      }
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

extern "C" {

int msgpu_seq_parse(const char *path, int is_fastq, msgpu_seqfile **out) { return msgpu::seq_parse_into(path, is_fastq, nullptr, out); }

void msgpu_seq_free(msgpu_seqfile *f) { delete f; }
uint32_t msgpu_seq_count(const msgpu_seqfile *f) { return f ? static_cast<uint32_t>(f->names.size()) : 0; }
const char *msgpu_seq_name(const msgpu_seqfile *f, uint32_t i) {
  return f && i < f->names.size() ? f->names[i].c_str() : nullptr;
}
uint64_t msgpu_seq_length(const msgpu_seqfile *f, uint32_t i) { return f && i < f->names.size() ? f->len[i] : 0; }
const char *msgpu_seq_bases(const msgpu_seqfile *f, uint32_t i) {
  return f && i < f->names.size() && f->bases.p ? f->bases.p + f->off[i] : nullptr;
}
uint64_t msgpu_seq_offset(const msgpu_seqfile *f, uint32_t i) { return f && i < f->names.size() ? f->off[i] : 0; }
const char *msgpu_seq_buffer(const msgpu_seqfile *f, uint64_t *bytes) {
  if (bytes) *bytes = f ? f->extent : 0;
  return f ? f->bases.p : nullptr;
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
