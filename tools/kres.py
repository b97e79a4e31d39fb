"""Kernel resource table from hipcc -Rpass-analysis=kernel-resource-usage remarks (file argv[1], name filter argv[2])."""
import re
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, rows = None, {}
for line in open(sys.argv[1]):
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for n, r in rows.items():
    if flt in n:
        print("%-70s vgpr %3d sgpr %3d scratch %4d spill %3d occ %d" % (
            n[:70], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize [bytes/lane]", -1),
            r.get("VGPRs Spill", -1), r.get("Occupancy [waves/SIMD]", -1)))
