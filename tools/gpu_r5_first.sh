#!/bin/bash
# round 5, first GPU call: the new tests (released 0.2.0 architecture, ABI) + the whole GPU suite + one bench line (reference CPU baseline, copy ceiling)
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_v020.py -x -q -m gpu > $O/test_v020.log 2>&1; echo "v020 exit $?"; tail -3 $O/test_v020.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/test_gpu.log 2>&1; echo "gpu suite exit $?"; tail -3 $O/test_gpu.log
timeout 600 python bench.py > $O/bench_line_first.json 2> $O/bench_first.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_line_first.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_ms", "frac")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")}, "port", (d["cpu_baseline"].get("port") or {}).get("value"))
print("hbm", {k: d["roofline_hbm"].get(k) for k in ("stream_copy_gbs", "copy_ceiling_gbs")})
print("kernel_ms", d["kernel_ms_per_step"])
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in d["configs"].items()})
PY
