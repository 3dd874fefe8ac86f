// muchsalsa_gpu -- the reference's executable on libmsgpu, no Python in the process:
//     muchsalsa_gpu <contigs.paf> <unitigs.fa> <nanopore.fa|fq> <outdir> [threads] [wiggleRoom = 300]
// (the argument list of src/Application.cpp:34-39; writes outdir/temp_1.target.fa, temp_1.query.fa, temp_1.align.paf like
// src/main.cpp:130-322).  The body of main() is msgpu::assemble (include/msgpu_adapter.hpp): straight calls into the C-ABI.
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <thread>

#include "msgpu_adapter.hpp"

int main(int argc, char **argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s <contigs.paf> <unitigs.fa> <nanopore.fa|fq> <outdir> [threads] [wiggleRoom=300]\n", argv[0]);
    return -1; // Application::checkIntegrity failing (src/main.cpp:134-137)
  }
  unsigned threads = std::thread::hardware_concurrency();
  threads          = threads == 0 ? 1 : (threads > 16 ? 16 : threads);
  if (argc > 5) threads = static_cast<unsigned>(std::max(1, std::atoi(argv[5])));
  const std::size_t wiggle = argc > 6 ? static_cast<std::size_t>(std::max(0, std::atoi(argv[6]))) : 300;
  try {
    const msgpu::AssemblyCounts n = msgpu::assemble(argv[1], argv[2], argv[3], argv[4], threads, wiggle, 0);
    std::printf("{\"rows\": %llu, \"reads\": %llu, \"edges\": %llu, \"orders\": %llu, \"contraction_edges\": %llu, \"paths\": %llu, "
                "\"paths_skipped\": %llu, \"contigs\": %llu, \"target_bases\": %llu, \"queries\": %llu}\n",
                static_cast<unsigned long long>(n.rows), static_cast<unsigned long long>(n.reads),
                static_cast<unsigned long long>(n.edges), static_cast<unsigned long long>(n.orders),
                static_cast<unsigned long long>(n.contractionEdges), static_cast<unsigned long long>(n.paths),
                static_cast<unsigned long long>(n.pathsSkipped), static_cast<unsigned long long>(n.contigs),
                static_cast<unsigned long long>(n.targetBases), static_cast<unsigned long long>(n.queries));
    std::puts("Finished assembly"); // src/main.cpp:320
  } catch (std::exception const &e) {
    std::fprintf(stderr, "muchsalsa_gpu: %s\n", e.what());
    return 1;
  }
  return 0;
}
