#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the tile kernels for each engine build given.
R=$PWD; O=$R/gpurun_out/pmcab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    CHGNET_HIP_LIB=$R/$lib timeout 150 rocprofv3 --pmc $c --output-format csv -d $O -o ${tag}_$c -- python $R/tools/gpu_scale_probe.py 1024 > $O/${tag}_$c.log 2>&1
  done
  python - "$O" "$tag" <<'PY'
import csv, sys, collections
O, tag = sys.argv[1], sys.argv[2]
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f"{O}/{tag}_{c}_counter_collection.csv")):
        if row["Counter_Name"] == c:
            a = acc[row["Kernel_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, (n, s) in acc.items(): out[k][c] = s / n
for k, v in sorted(out.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:8]:
    f, w = v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)
    print(f"{tag:22s} {k[:48]:48s} fetch {2*f*1024/1e9:6.2f} GB write {w*1024/1e9:6.2f} GB")
PY
done
