set -o pipefail
for f in muchsalsa_amd/csrc/*.hip muchsalsa_amd/csrc/*.cpp muchsalsa_amd/csrc/*.h include/*.h; do
  if [ "$f" -nt muchsalsa_amd/libmsgpu.so ]; then echo "STALE LIBRARY: $f"; exit 1; fi
done
mkdir -p gpurun_out/r5_11
timeout -k 10 300 python -m pytest tests/test_packed_rows.py -x -q --timeout 120 > gpurun_out/r5_11/pytest_packed.log 2>&1; tail -3 gpurun_out/r5_11/pytest_packed.log
timeout -k 10 400 python bench.py --cpu-sample-reads 0 --assemble-window-mb -1 --no-consensus --no-tiled --no-e2e > gpurun_out/r5_11/bench_h2h.json 2> gpurun_out/r5_11/bench_h2h.err || tail -20 gpurun_out/r5_11/bench_h2h.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_11/bench_h2h.json"))
print("ms_per_step", d["ms_per_step"])
c=d["config"]
print({k:v for k,v in c.items() if k.startswith("survey_8d") and k!="survey_8d_note"})
print(d["host_to_host"]["without_edgematches"].get("packed_rows"))
print({k:d["host_to_host"]["without_edgematches"][k] for k in ("ms","load_ms","compute_done_ms","ms_samples")})
print(d.get("group"))
PY
