#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of profiles/pmc_calib.hip's kernels against their known byte counts.

usage: pmc_calib_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <program stdout> [out.json]

For every pattern: counter bytes (KiB x 1024, averaged over the launches) divided by the bytes the lanes asked for, by
the bytes at 64-B sector granularity and by the bytes at 128-B line granularity.  The ratio closest to 1 (or to 0.5)
says what the counter tallies for that pattern."""
import csv
import json
import re
import sys


def load(path, counter):
    rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]))
            for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort()
    per, seen = {}, {}
    for _, k, v in rows:
        name = re.sub(r"\(.*", "", k).strip()
        if name == "read_gather8":  # launched twice per repetition: 1 GiB table first, 32 MiB table second
            seen[name] = seen.get(name, 0) + 1
            name += "_big" if seen[name] % 2 == 1 else "_mall"
        per.setdefault(name, []).append(v * 1024)
    return {k: sum(v) / len(v) for k, v in per.items()}


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for line in open(sys.argv[3]):
        m = re.match(r"(\w+)\s+asked=(\d+) sector64=(\d+) line128=(\d+)", line)
        if not m:
            continue
        name, asked, s64, l128 = m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4))
        c = (w if name.startswith("write") else f).get(name)
        if c is None:
            continue
        out[name] = {"counter_bytes": int(c), "asked_bytes": int(asked), "counter/asked": round(c / asked, 3),
                     "counter/sector64": round(c / s64, 3), "counter/line128": round(c / l128, 3)}
        print(f"{name:20s} counter={c/1e6:10.1f} MB  /asked={c/asked:6.3f}  /sector64={c/s64:6.3f}  /line128={c/l128:6.3f}")
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
