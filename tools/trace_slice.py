"""Frames [first, last) of a rocprofv3 --kernel-trace CSV of bench.py (frames counted from the start by their k_merge_ticks launch) as a
small CSV for offline analysis: name, stream, queue, start / end (ns from the slice's first kernel), grid, workgroup, LDS, VGPRs.
    python tools/trace_slice.py <trace dir> <first> <last> > gpurun_out/slice.csv"""
import csv
import glob
import os
import re
import sys

d, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_merge_ticks" in r["Kernel_Name"]]
lo, hi = starts[first], starts[last]
t0 = int(rows[lo]["Start_Timestamp"])
sh = lambda n: re.sub(r"\((anonymous namespace)::\w+Args.*|\(float.*|\(int.*|\(HIP_vector.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
k = rows[0].keys()
get = lambda r, *names: next((r[n] for n in names if n in k), "")
w = csv.writer(sys.stdout)
w.writerow(["name", "stream", "queue", "start_ns", "end_ns", "grid", "wg", "lds", "vgpr"])
for r in rows[lo:hi]:
    w.writerow([sh(r["Kernel_Name"]), get(r, "Stream_Id"), get(r, "Queue_Id"), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0,
                get(r, "Grid_Size_X", "Grid_Size"), get(r, "Workgroup_Size_X", "Workgroup_Size"), get(r, "LDS_Block_Size", "LDS_Block_Size_v"),
                get(r, "VGPR_Count", "Arch_VGPR_Count")])
