#!/bin/bash
# Which engine carries the dispatcher's device-to-host copies, and what it costs the windows beside them: the lean region under
# the runtime's copy-engine switches (each run is its own process: the switches are read at start-up).
out=${1:-gpurun_out/d2h_engine}
mkdir -p $out
run() { label=$1; shift; env "$@" python tools/experiments/lean_wall.py "$label" 0 2>>$out/err.txt | tee -a $out/summary.txt || exit 1; }
run default A=1
run sdma1 HSA_ENABLE_SDMA=1
run sdma0 HSA_ENABLE_SDMA=0
run blit_engine_kernel GPU_BLIT_ENGINE_TYPE=3
run blit_engine_host GPU_BLIT_ENGINE_TYPE=1
run force_blit_0 GPU_FORCE_BLIT_COPY_SIZE=0
run force_blit_big GPU_FORCE_BLIT_COPY_SIZE=1048576
env A=1 python tools/experiments/lean_wall.py default_w8 8 | tee -a $out/summary.txt
