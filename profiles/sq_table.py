"""Table of SQ counters per tile kernel from the three passes of tools/gpu_round5_profiles.sh (SQ pass) (gpurun_out/prof2/sq{1,2,3}_*.csv).

    python profiles/sq_table.py            # prints the markdown table used in r01_sq_counters.md
"""
import csv, glob, os, sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "prof2")
LABELS = [("k_angle<true, true", "bondconv_bwd"), ("k_angle<true, false", "bondconv_fwd"), ("k_angle<false, true", "angleupd_bwd"),
          ("k_angle<false, false", "angleupd_fwd"), ("k_atomconv_bwd", "atomconv_bwd"), ("k_atomconv_fwd", "atomconv_fwd")]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in glob.glob(os.path.join(SRC, "sq*_counter_collection.csv")):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            for sub, lab in LABELS:
                if sub in row["Kernel_Name"]:
                    a = acc[lab][row["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
print("| kernel | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | VALU | LDS | MFMA busy | VALU insts / MFMA inst |")
print("|---|---|---|---|---|---|---|---|")
for _, lab in LABELS:
    c = {k: v[1] / max(v[0], 1) for k, v in acc[lab].items()}
    if not c:
        continue
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mfma_busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024) if gui else float("nan")
    print(f"| {lab} | {100*c.get('SQ_WAIT_ANY',0)/wc:.0f} % | {100*c.get('SQ_WAIT_INST_ANY',0)/wc:.0f} % | {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f} % | "
          f"{100*c.get('SQ_ACTIVE_INST_VALU',0)/wc:.0f} % | {100*c.get('SQ_ACTIVE_INST_LDS',0)/wc:.1f} % | {100*mfma_busy:.0f} % | "
          f"{c.get('SQ_INSTS_VALU',0)/max(c.get('SQ_INSTS_MFMA',1),1):.1f} |")
