O=$PWD/gpurun_out/r06/chain1; R=$PWD; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sc in 2,2,2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$sc -o t -- python $R/tools/gpu_md_step_probe.py 60 $sc > $O/prof_run_$sc.log 2>&1
f=$(find $O/prof_$sc -name "*kernel_stats.csv" | head -1); cp $f $O/step_kernel_stats_$sc.csv; rm -rf $O/prof_$sc
done
python - <<PY
import csv
for sc in ("2,2,2",):
    rows=list(csv.DictReader(open("$O/step_kernel_stats_%s.csv"%sc)))
    print(sc)
    tot=0
    for r in rows[:44]:
        per=float(r['TotalDurationNs'])/130/1e3
        tot+=per
        print(f"{r['Name'][:80]:80s} calls/step {int(r['Calls'])/130:5.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  per step {per:7.1f}")
    print("sum per step", tot)
PY
