#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs (one directory per pass) into profiles/<name>.csv and
profiles/pmc_latest.json (read by bench.py for roofline.traffic).

    python tools/pmc_summary.py <out_name> <workload> <pass_dir> [<pass_dir> ...]

<workload> = the key bench.py forms for the run the counters were collected on, "c<config>:<P>:<W>x<H>" (e.g.
c3:1000000:1920x1080); bench.py uses the counters only for that workload.

Traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate passes;
bytes = FETCH_SIZE*1024*2 (gfx950 reports exactly half of a wide coalesced read stream) + WRITE_SIZE*1024.
"""
import collections
import csv
import glob
import json
import os
import statistics as st
import sys


def main():
    name, workload, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    counters = sorted({c for k in agg for c in agg[k]})
    rows, latest = [], {}
    for k in sorted(agg):
        if not k.startswith("k_"):
            continue
        row = {"kernel": k, "dispatches": max(len(v) for v in agg[k].values())}
        for c in counters:
            row[c] = st.mean(agg[k][c]) if c in agg[k] else ""
        if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            row["hbm_read_bytes_corrected"] = 2 * 1024 * st.mean(agg[k]["FETCH_SIZE"])
            row["hbm_write_bytes"] = 1024 * st.mean(agg[k]["WRITE_SIZE"])
            latest[k] = {"read_bytes": row["hbm_read_bytes_corrected"], "write_bytes": row["hbm_write_bytes"],
                         "traffic_bytes": row["hbm_read_bytes_corrected"] + row["hbm_write_bytes"]}
            if "SQ_INSTS_VALU" in agg[k]:      # wave-level VALU instructions per launch (issue-bound kernels)
                latest[k]["valu_insts"] = st.mean(agg[k]["SQ_INSTS_VALU"])
        rows.append(row)
    fields = ["kernel", "dispatches"] + counters + ["hbm_read_bytes_corrected", "hbm_write_bytes"]
    with open(os.path.join(root, "profiles", name + ".csv"), "w") as fo:
        w = csv.DictWriter(fo, fieldnames=fields)
        w.writeheader()
        for r in rows:
            w.writerow({f: r.get(f, "") for f in fields})
    # pmc_latest.json holds one entry per workload; bench.py picks the one matching its own run
    path = os.path.join(root, "profiles", "pmc_latest.json")
    try:
        doc = json.load(open(path))
        if "workloads" not in doc:
            doc = {"workloads": {doc["workload"]: {"source": doc["source"], "kernels": doc["kernels"]}}} if "workload" in doc \
                else {"workloads": {}}
    except Exception:
        doc = {"workloads": {}}
    doc["workloads"][workload] = {"source": name + ".csv", "kernels": latest}
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", name + ".csv", "and pmc_latest.json for", len(rows), "kernels")


if __name__ == "__main__":
    main()
