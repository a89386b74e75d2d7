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
sys.path.insert(0, REPO)
from bench import csrc_hash  # noqa: E402  the counters are stamped with the hash of the kernel sources they were taken on

SRC_HASH = csrc_hash()
SRC = os.path.join(REPO, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r01", "prof")
if not os.path.isdir(SRC):
    SRC = os.path.join(REPO, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

LABELS = {  # kernel-name substring -> bench.py label
    "k_angle_bwd_w<true": "bondconv_bwd", "k_angle<true, false": "bondconv_fwd",     # per-atom adjoints (kernels_angle_w.h); the plain
    "k_angle_bwd_w<false": "angleupd_bwd", "k_angleupd_fwd_a": "angleupd_fwd",       # k_angle<*, true> / <false, false> launches return
                                                                                      # at once (per-atom forward: kernels_angle_fa.h)
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
        if sub in name and label not in latest:      # rows are sorted by traffic: the launch that did the work comes first
            latest[label] = {"hbm_bytes_per_launch": int(hbm), "fetch_kib_reported": round(f, 1), "write_kib_reported": round(w, 1),
                             "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB), MI355X_MICROARCH.md HBM section", "profile": f"profiles/{tag}_pmc_summary.csv"}
with open(os.path.join(REPO, "profiles", f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
    w_ = csv.writer(fh)
    w_.writerow(["kernel", "launches", "mean_FETCH_SIZE_KiB_reported", "mean_WRITE_SIZE_KiB_reported", "hbm_bytes_per_launch_corrected"])
    w_.writerows(rows)
# L2 hit rates (separate --pmc TCC_HIT_sum TCC_MISS_sum pass, tools/gpu_round5_profiles.sh) next to the traffic
l2_path = os.path.join(SRC, "l2a_counter_collection.csv")
if os.path.exists(l2_path):
    hit, miss = mean_counter(l2_path, "TCC_HIT_sum"), mean_counter(l2_path, "TCC_MISS_sum")
    l2_rows = []
    for name in sorted(hit, key=lambda k: -(hit[k][1] + miss.get(k, (0, 0.0))[1])):
        h, m = hit[name][1], miss.get(name, (0, 0.0))[1]
        if h + m <= 0:
            continue
        l2_rows.append([name, hit[name][0], int(h), int(m), round(h / (h + m), 4)])
        for sub, label in LABELS.items():
            if sub in name and label in latest and "l2_hit_rate" not in latest[label]:
                latest[label]["l2_hit_rate"] = round(h / (h + m), 4)
    with open(os.path.join(REPO, "profiles", f"{tag}_l2_counters.csv"), "w", newline="") as fh:
        w_ = csv.writer(fh)
        w_.writerow(["kernel", "launches", "mean_TCC_HIT_sum", "mean_TCC_MISS_sum", "hit_rate"])
        w_.writerows(l2_rows[:40])
latest["csrc_hash"] = SRC_HASH
json.dump(latest, open(os.path.join(REPO, "profiles", "pmc_latest.json"), "w"), indent=1)
for r in rows[:12]:
    print(r)


# ---- SQ counters (separate --pmc pass): what the wave slots of each kernel spend their cycles on -> profiles/sq_latest.json,
#      quoted by bench.py next to the roofline fractions (classification: matrix pipe / vector issue / waiting on memory)
SQ_LABELS = dict(LABELS, **{"k_bond_embed_t<false": "bond_embed_fwd", "k_bond_embed_t<true": "bond_embed_bwd",
                            "k_angle_embed_t<false": "angle_embed_fwd", "k_angle_embed_t<true": "angle_embed_bwd",
                            "k_edge_force": "edge_force", "k_rows_gemm<128, 64, 1>": "gemm_GQ"})
sq_path = os.path.join(SRC, "pmc_sq_counter_collection.csv")
if os.path.exists(sq_path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(sq_path) as fh:
        for row in csv.DictReader(fh):
            for sub, lab in SQ_LABELS.items():
                if sub in row["Kernel_Name"]:
                    a = acc[lab][row["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    sq = {}
    for lab, cs in acc.items():
        c = {k: v[1] / max(v[0], 1) for k, v in cs.items()}
        wc, gui = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0, c.get("GRBM_GUI_ACTIVE", 0.0)
        sq[lab] = {"wait_any": round(c.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_any": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                   "active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "active_inst_valu": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                   "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024), 3) if gui else None}
        s_ = sq[lab]
        s_["bound"] = ("memory (wave slots idle: HBM bandwidth / latency)" if s_["active_inst_any"] < 0.2 else
                       "memory latency (waits on LDS / L2 / atomics)" if s_["wait_any"] >= 0.45 else "vector-ALU issue")
    json.dump({"unit": "fraction of SQ_WAVE_CYCLES (mfma_busy: of the SIMD cycles of the launch)", "profile": f"gpurun_out/{tag}/prof/pmc_sq_*", "csrc_hash": SRC_HASH,
               "kernels": sq}, open(os.path.join(REPO, "profiles", "sq_latest.json"), "w"), indent=1)
    for k, v in sq.items():
        print(k, v)


# ---- one training step (tools/gpu_train_probe.py): kernel stats verbatim, HBM counters of the second-order sweep's kernels
for name in ("train", "md", "md512"):
    src = os.path.join(SRC, f"{name}_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(REPO, "profiles", f"{tag}_{name}_kernel_stats.csv"))
tf, tw = os.path.join(SRC, "train_fetch_counter_collection.csv"), os.path.join(SRC, "train_write_counter_collection.csv")
if os.path.exists(tf) and os.path.exists(tw):
    fetch_t, write_t = mean_counter(tf, "FETCH_SIZE"), mean_counter(tw, "WRITE_SIZE")
    rows_t = []
    for name in sorted(set(fetch_t) | set(write_t), key=lambda k: -(fetch_t.get(k, (0, 0))[0] * (2 * fetch_t.get(k, (0, 0))[1] + write_t.get(k, (0, 0))[1]))):
        n, f = fetch_t.get(name, (0, 0.0))
        _, w = write_t.get(name, (0, 0.0))
        rows_t.append([name, n, round(f, 1), round(w, 1), int((2.0 * f + w) * 1024.0), int(n * (2.0 * f + w) * 1024.0)])
    with open(os.path.join(REPO, "profiles", f"{tag}_train_pmc_summary.csv"), "w", newline="") as fh:
        w_ = csv.writer(fh)
        w_.writerow(["kernel", "launches_in_probe", "mean_FETCH_SIZE_KiB_reported", "mean_WRITE_SIZE_KiB_reported", "hbm_bytes_per_launch_corrected",
                     "hbm_bytes_total_in_probe"])
        w_.writerows(rows_t[:40])
    print("train pmc:", rows_t[:6])
