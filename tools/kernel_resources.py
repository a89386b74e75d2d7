"""Register / scratch usage of the tile kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> remarks.txt; python tools/kernel_resources.py remarks.txt [filter ...]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
filters = sys.argv[2:] or ["k_angle", "k_atomconv"]
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not any(f in dn for f in filters):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{dn[:64]:64s} VGPR {g('    VGPRs'):>3} AGPR {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {scratch:>4} SGPR {g('TotalSGPRs'):>3} occ {occ}")
