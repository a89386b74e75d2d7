#!/bin/bash
# Round-5 check on one GPU box: the whole GPU suite (parity tests through the C-ABI), then one bench line.
#   gpurun --timeout 2700 -- 'bash tools/gpu_round5_check.sh'   ->  gpurun_out/r05/{test_gpu.log, bench_line_check.json}
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/test_gpu.log 2>&1; echo "gpu suite exit $?"; tail -3 $O/test_gpu.log
timeout 600 python bench.py > $O/bench_line_check.json 2> $O/bench_check.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_line_check.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_ms", "frac", "traffic")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")}, "port", (d["cpu_baseline"].get("port") or {}).get("value"))
print("hbm", {k: d["roofline_hbm"].get(k) for k in ("stream_copy_gbs", "copy_ceiling_gbs")})
print({k: (v.get("structures_per_s") or v.get("steps_per_s")) for k, v in d["configs"].items()})
PY
