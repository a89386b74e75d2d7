"""Summarise rocprofv3 outputs (gpurun_out/prof) into committed, judged artefacts under profiles/.

    python profiles/summarize.py r01            # reads gpurun_out/prof/{ktrace,pmc_fetch,pmc_write}_*.csv

Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --kernel-trace --stats summary),
profiles/<tag>_pmc_summary.csv (per kernel: launches, mean FETCH_SIZE / WRITE_SIZE in KiB as reported,
and HBM bytes per launch with the gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE counts
128-B requests at 64 B for wide coalesced reads -> x2; WRITE_SIZE taken as reported) and
profiles/pmc_latest.json (kernel label -> bytes per launch) which bench.py reports as roofline.traffic.
"""
import csv, json, os, shutil, sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r01", "prof")
if not os.path.isdir(SRC):
    SRC = os.path.join(REPO, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

LABELS = {  # kernel-name substring -> bench.py label
    "k_angle<true, true": "bondconv_bwd", "k_angle<true, false": "bondconv_fwd",
    "k_angle_bwd_w<false": "angleupd_bwd", "k_angle<false, false": "angleupd_fwd",   # (the plain k_angle<false, true> launch returns at once)
    "k_atomconv_bwd": "atomconv_bwd", "k_atomconv_fwd": "atomconv_fwd",
}

shutil.copy(os.path.join(SRC, "ktrace_kernel_stats.csv"), os.path.join(REPO, "profiles", f"{tag}_kernel_stats.csv"))

def mean_counter(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                a = acc[row["Kernel_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}

fetch = mean_counter(os.path.join(SRC, "pmc_fetch_counter_collection.csv"), "FETCH_SIZE")
write = mean_counter(os.path.join(SRC, "pmc_write_counter_collection.csv"), "WRITE_SIZE")
rows, latest = [], {}
for name in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1])):
    n, f = fetch.get(name, (0, 0.0))
    _, w = write.get(name, (0, 0.0))
    hbm = (2.0 * f + w) * 1024.0
    rows.append([name, n, round(f, 1), round(w, 1), int(hbm)])
    for sub, label in LABELS.items():
        if sub in name:
            latest[label] = {"hbm_bytes_per_launch": int(hbm), "fetch_kib_reported": round(f, 1), "write_kib_reported": round(w, 1),
                             "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB), MI355X_MICROARCH.md HBM section", "profile": f"profiles/{tag}_pmc_summary.csv"}
with open(os.path.join(REPO, "profiles", f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
    w_ = csv.writer(fh)
    w_.writerow(["kernel", "launches", "mean_FETCH_SIZE_KiB_reported", "mean_WRITE_SIZE_KiB_reported", "hbm_bytes_per_launch_corrected"])
    w_.writerows(rows)
json.dump(latest, open(os.path.join(REPO, "profiles", "pmc_latest.json"), "w"), indent=1)
for r in rows[:12]:
    print(r)
