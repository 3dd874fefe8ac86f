#!/bin/bash
# Host-side sanitizer run (CPU build container; GPU AddressSanitizer is not available on the pool): builds libmsgpu
# with -fsanitize=address,undefined for the host code only and runs the CPU tests of the host stages against it.
#   tools/asan_cpu.sh [pytest args]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CLANG=/opt/rocm/lib/llvm/bin/clang
OUT=${TMPDIR:-/tmp}/libmsgpu_asan.so
cd "$ROOT/muchsalsa_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -pthread -ffp-contract=off \
  -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -I../../include -I. -shared -o "$OUT" \
  msgpu_api.hip msgpu_kernels.hip msgpu_index.hip msgpu_graph.hip msgpu_seq.hip msgpu_group.cpp wire_host.cpp paf_loader.cpp seq_loader.cpp seg_compose.cpp \
  consensus_base.cpp assemble_path.cpp graph_stage.cpp
cd "$ROOT"
LD_PRELOAD="$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 MSGPU_LIB="$OUT" \
MSGPU_GRAPH_PAR_MIN=64 MSGPU_GRAPH_THREADS=6 MSGPU_SEQ_THREADS=5 \
  python -m pytest tests/test_assemble_path.py tests/test_graph_stage.py tests/test_graph_fullsize.py tests/test_segments.py \
  tests/test_sequences_loader.py tests/test_paf_loader.py tests/test_cfg1_plumbing.py tests/test_ref_test_vectors.py tests/test_wire_host.py -x -q -m "not gpu" "$@"
