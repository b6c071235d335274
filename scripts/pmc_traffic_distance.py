#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each in its own run) of `bench.py --workload distance` -> HBM bytes per LAUNCH of the distance
kernel and of the ranking kernel, with the corrections of MI355X_MICROARCH.md (KiB units; gfx950 tallies the 128-byte requests of wide streaming reads at
64 B: reads doubled).      python scripts/pmc_traffic_distance.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import kernel_source_hash, load

fdir, wdir, out_json = sys.argv[1:4]
fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
res = {"kernel_source_hash": kernel_source_hash(),
       "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE KiB x1024; averages per launch", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    n = max(fetch.get(k, [0, 0])[1], write.get(k, [0, 0])[1])
    if n == 0 or not any(t in k for t in ("pairdist", "rank_query", "rows_prep")): continue
    rd = fetch.get(k, [0, 1])[0] * 1024 * 2.0 / max(fetch.get(k, [0, 1])[1], 1)
    wr = write.get(k, [0, 1])[0] * 1024 / max(write.get(k, [0, 1])[1], 1)
    res["kernels"][k[:90]] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr)}
    if "pairdist_dma_kernel<3>" in k: res["pairdist_bf16x3_hbm_bytes_per_launch"] = round(rd + wr)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res, indent=1))
