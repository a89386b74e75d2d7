#!/bin/bash
# kernel trace of the MD-size step (one 256-atom cell): kernel durations vs the gaps between dependent launches
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mdtrace
mkdir -p $OUT
CHGNET_HIP_GRAPHS=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python $GRAFT_REPO_ROOT/tools/gpu_md_breakdown.py ef > $OUT/run.log 2>&1
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/mdtrace"
f = glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the 50-replay loop: take a window of consecutive launches in the middle of the trace
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
n = len(ev)
lo, hi = n // 4, n // 4 + 83 * 20
dur = collections.defaultdict(list); gap = []
for i in range(lo, min(hi, n - 1)):
    s, e, k = ev[i]
    dur[k.split("(")[0][:60]].append(e - s)
    gap.append(ev[i + 1][0] - e)
tot_d = sum(sum(v) for v in dur.values()); tot_g = sum(gap)
print(f"launches {hi - lo}: kernel time {tot_d / 1e3 / 20:.1f} us per step, gaps {tot_g / 1e3 / 20:.1f} us per step, median gap {sorted(gap)[len(gap) // 2] / 1e3:.2f} us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"  {k:60s} n={len(v):5d} avg={sum(v) / len(v) / 1e3:7.2f} us  total/step={sum(v) / 1e3 / 20:7.1f} us")
PY
