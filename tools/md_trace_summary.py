"""Summary of a rocprofv3 --kernel-trace csv of tools/gpu_md_replay_probe.py: the steady-state launch sequence of one prediction,
per position the kernel's median duration and the median gap to the previous dispatch's end."""
import csv, glob, os, sys, statistics as st
root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# one prediction = the launches from one k_cart to the next
starts = [i for i, n in enumerate(names) if "k_cart" in n]
seqs = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
L = st.mode([len(s) for s in seqs])
seqs = [s for s in seqs if len(s) == L][len(seqs) // 4:]          # steady state: drop the first quarter
print(f"{len(seqs)} predictions of {L} dispatches")
tot_k = tot_g = 0.0
for i in range(L):
    dur = st.median([s[i][1] - s[i][0] for s in seqs]) / 1e3
    gap = st.median([s[i][0] - s[i - 1][1] for s in seqs]) / 1e3 if i else 0.0
    tot_k += dur; tot_g += gap
    nm = seqs[0][i][2]
    nm = nm[:nm.index("(")] if "(" in nm else nm
    print(f"{i:3d} {nm[:70]:70s} dur {dur:7.1f} us  gap before {gap:6.1f} us")
span = st.median([s[-1][1] - s[0][0] for s in seqs]) / 1e3
period = st.median([b[0][0] - a[0][0] for a, b in zip(seqs[:-1], seqs[1:])]) / 1e3
print(f"kernel time {tot_k:.1f} us + gaps {tot_g:.1f} us = span {span:.1f} us; period between predictions {period:.1f} us")
