#!/usr/bin/env python3
"""profiles/<tag>_pmc_pillar.json from the tools/pmc_report.py summaries of the FETCH_SIZE / WRITE_SIZE (/ MFMA busy) passes over
tools/pmc_pillar.py.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md), WRITE_SIZE is
taken as reported; raw values are KB per dispatch.

    python tools/pmc_pillar_json.py <fetch.txt> <write.txt> [<mfma.txt>] > profiles/r04_pmc_pillar.json
"""
import ast
import json
import re
import sys


def parse(path):
    """kernel -> (second cloud's counters, first cloud's counters)"""
    out, last = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?) (\{.*\}) n=\d+", line)
        if m:
            last = "k_rows" if "k_rows" in m.group(1) else ("k_bin" if "k_bin" in m.group(1) else None)
            if last:
                out[last] = [ast.literal_eval(m.group(2)), None]
            continue
        m = re.match(r"^\s+first cloud: (\{.*\})", line)
        if m and last:
            out[last][1] = ast.literal_eval(m.group(1))
    return out


fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
mfma = parse(sys.argv[3]) if len(sys.argv) > 3 else {}
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE" + (" / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" if mfma else "") +
       " (separate passes) on tools/pmc_pillar.py, MI355X, round 6 (k_bin / k_rows with 64-byte records, zeros written by k_rows, the canvas bound written by k_rows: the default); FETCH_SIZE doubled per "
       "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as reported; tools/pmc_pillar_json.py"}
for k in ("k_rows", "k_bin"):
    res[k] = {}
    for idx, pts in ((1, "32769"), (0, "196608")):      # pmc_pillar.py runs the 32 769-point cloud first
        f = fetch[k][idx]["FETCH_SIZE"]
        w = write[k][idx]["WRITE_SIZE"]
        e = {"fetch_kb_raw": f, "write_kb_raw": w, "traffic_bytes": int(round((2 * f + w) * 1024, -2))}
        if k in mfma and mfma[k][idx]:
            b, s = mfma[k][idx].get("SQ_VALU_MFMA_BUSY_CYCLES"), mfma[k][idx].get("SQ_BUSY_CYCLES")
            if b and s:
                e.update(sq_busy_cycles=s, mfma_busy_cycles=b, mfma_busy_frac=round(b / (32 * s), 3))
        res[k][pts] = e
print(json.dumps(res, indent=1))
