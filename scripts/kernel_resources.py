"""Registers, spills, LDS and occupancy of every kernel in one csrc file, as hipcc reports them for gfx950.

    python scripts/kernel_resources.py knn [substring-of-kernel-name]

Runs here (hipcc cross-compiles without a GPU); nothing is written into the tree.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morna_amd.build import FLAGS  # noqa: E402

FIELDS = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill",
          "LDS Size [bytes/block]"]


def main():
    src = os.path.join(ROOT, "morna_amd", "csrc", sys.argv[1] + ".hip")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc"] + [f for f in FLAGS if f != "-shared"] + [
            "-c", src, "-o", os.path.join(tmp, "x.o"), "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    name, rows, cur = None, [], {}
    for line in err.splitlines():
        m = re.search(r"remark: [^ ]+ +(?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            if name:
                rows.append((name, cur))
            name, cur = m.group(1), {}
            continue
        for f in FIELDS:
            m = re.search(re.escape(f) + r": (\d+)", line)
            if m:
                cur[f] = int(m.group(1))
    if name:
        rows.append((name, cur))
    print("%-60s %5s %5s %7s %4s %6s %6s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "sSpill", "vSpill", "LDS"))
    for name, cur in rows:
        short = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", short).replace("morna::", "").replace("void ", "")
        if want not in short:
            continue
        print("%-60s %5d %5d %7d %4d %6d %6d %7d" % tuple([short[:60]] + [cur.get(f, -1) for f in FIELDS]))


if __name__ == "__main__":
    main()
