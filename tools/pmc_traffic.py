#!/usr/bin/env python
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py into per-launch HBM-side traffic of the GEMM kernels.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [workload] [precise groups, comma list]

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, "HBM"), so reads are doubled; WRITE_SIZE is taken as reported (it matches the algorithmic output
bytes of the GEMMs exactly, see profiles/r01_pmc_summary.txt).
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


def main():
    fetch, fcnt = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, wcnt = per_kernel(sys.argv[2], "WRITE_SIZE")
    is_gemm = lambda n: "gemm_" in n and "kernel" in n
    names = sorted(n for n in fetch if is_gemm(n))
    launches = sum(fcnt[n] for n in names)
    fkib = sum(fetch[n] for n in names)
    wkib = sum(write.get(n, 0.0) for n in names)
    from labelanything_amd.engine import PRECISE_WIDE
    workload = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    episodes = bench.WORKLOADS[workload]["default_episodes"]          # the passes run `bench.py` with its default batch
    precise = list(PRECISE_WIDE) if len(sys.argv) <= 5 or sys.argv[5] == "default" else [g for g in sys.argv[5].split(",") if g and g != "none"]
    tw = sorted(n for n in fetch if "twoway_" in n and "merge" not in n)
    out = {
        "workload": workload, "encoder_split_precision": precise,      # bench.py only attaches this file to a matching run
        "episodes_per_step": episodes,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --no-graphs --no-cpu-baseline`",
        "kernels": names,
        "launches_counted": launches,
        "fetch_kib_per_launch_raw": round(fkib / max(launches, 1), 1),
        "write_kib_per_launch_raw": round(wkib / max(launches, 1), 1),
        "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
        "bytes_per_launch": round((2.0 * fkib + wkib) * 1024.0 / max(launches, 1)),
        "per_kernel": {n: {"launches": fcnt[n], "fetch_kib_raw": round(fetch[n] / fcnt[n], 1),
                           "write_kib_raw": round(write.get(n, 0.0) / max(wcnt.get(n, 1), 1), 1)} for n in names},
        # fused image-side kernels of the two-way transformer (csrc/twoway.hip): bytes per launch with the same x2 read correction;
        # algorithmic = the (groups, hw, 256) fp32 stream once (t2i: read) or twice (i2t: read + write)
        "twoway": {n: {"launches": fcnt[n], "bytes_per_launch": round((2.0 * fetch[n] / fcnt[n] + write.get(n, 0.0) / max(wcnt.get(n, 1), 1)) * 1024.0)}
                   for n in tw},
    }
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
