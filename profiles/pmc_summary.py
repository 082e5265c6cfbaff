#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json] [sq_counter_collection.csv]

With the optional fourth file (a separate `--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES` pass) the summary also
holds the vector / scalar / LDS wave-instructions each kernel issued per launch: `valu_insts` / time against the chip's
vector issue rate (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 /s) is the issue utilisation
of a VALU-bound kernel.

Counters are in KiB per dispatch.  On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md §HBM): `fetch_bytes_x2` applies that correction; WRITE_SIZE is exact for
16-B-per-lane streaming stores.  Other access widths are uncalibrated, so both raw and corrected
figures are kept.  Only this library's kernels are listed."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(pings::raster::\w+|\(anonymous namespace\)::\w+|rocprim::\w+)", name)
    base = m.group(1).split("::")[-1] if m else name[:40]
    t = re.search(r"<([^<>]*)>\(", name)
    return base + (f"<{t.group(1)}>" if t and "pings" in name or "anonymous" in name and t else "")


def load(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        if "pings" not in k and "anonymous namespace" not in k:
            continue
        name = short(k)
        fk = sum(f[k]) / len(f[k]) * 1024
        wk = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1) * 1024
        out[name] = {"launches": len(f[k]), "fetch_bytes_raw": int(fk), "fetch_bytes_x2": int(2 * fk),
                     "write_bytes": int(wk), "hbm_bytes_corrected": int(2 * fk + wk)}
    if len(sys.argv) > 4:
        sq = {c: load(sys.argv[4], c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")}
        for k in sq["SQ_INSTS_VALU"]:
            name = short(k)
            if name in out:
                for c, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"),
                               ("SQ_INSTS_LDS", "lds_insts"), ("SQ_WAVES", "waves")):
                    vals = sq[c].get(k, [0])
                    out[name][key] = int(sum(vals) / max(len(vals), 1))
    for n, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"]):
        print(f"{n:44s} n={v['launches']:3d} fetch_raw={v['fetch_bytes_raw']/1e6:9.1f} MB  write={v['write_bytes']/1e6:9.1f} MB  "
              f"corrected={v['hbm_bytes_corrected']/1e6:9.1f} MB")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
