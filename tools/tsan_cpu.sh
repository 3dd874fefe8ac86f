#!/bin/bash
# ThreadSanitizer run of the multi-threaded host stages (layout fan-out + parked thread pool, threaded row-table install,
# per-component graph workers): builds libmsgpu with -fsanitize=thread for the host code and runs their CPU tests.
#   tools/tsan_cpu.sh [pytest args]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CLANG=/opt/rocm/lib/llvm/bin/clang
OUT=${TMPDIR:-/tmp}/libmsgpu_tsan.so
cd "$ROOT/muchsalsa_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -pthread -ffp-contract=off \
  -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer -I../../include -I. -shared -o "$OUT" \
  msgpu_api.hip msgpu_kernels.hip msgpu_index.hip msgpu_graph.hip msgpu_seq.hip msgpu_group.cpp wire_host.cpp paf_loader.cpp seq_loader.cpp seg_compose.cpp \
  consensus_base.cpp assemble_path.cpp graph_stage.cpp
cd "$ROOT"
LD_PRELOAD="$($CLANG -print-file-name=libclang_rt.tsan-x86_64.so)" TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 \
  MSGPU_LIB="$OUT" MSGPU_GRAPH_PAR_MIN=64 MSGPU_GRAPH_THREADS=6 MSGPU_SEQ_THREADS=5 \
  python -m pytest tests/test_assemble_path.py tests/test_graph_stage.py tests/test_graph_fullsize.py tests/test_paf_loader.py \
  tests/test_sequences_loader.py tests/test_wire_host.py tests/test_ref_test_vectors.py -x -q -m "not gpu" "$@"
