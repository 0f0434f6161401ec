"""Register / scratch / occupancy summary of the library's kernels from hipcc's resource-usage remarks:
    hipcc <flags of the Makefile> -c -o /dev/null linear-programming_amd/csrc/lpamd_engine.hip -Rpass-analysis=kernel-resource-usage 2> res.txt
    python scripts/kernel_resources.py res.txt [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
names, rows = [], []
for m in re.finditer(r"remark: Function Name: (\S+)", txt):
    names.append((m.start(), m.group(1)))
names.append((len(txt), None))
dem = subprocess.run(["c++filt"] + [n for _, n in names[:-1]], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
for k, (pos, name) in enumerate(names[:-1]):
    body = txt[pos:names[k + 1][0]]
    d = dem[k].split("(")[0].replace("void lpamd::", "")
    if want and not any(w in d for w in want):
        continue
    g = lambda key: (re.search(key + r"[^:\n]*: (\S+)", body) or [None, "?"])[1]
    print("%-46s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s spillV %s spillS %s" % (
        d[-46:], g("    VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize"), g("Occupancy"), g("VGPRs Spill"), g("SGPRs Spill")))
