#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json] [sq_counter_collection.csv]
           ['{"gaussians": 1000000, "width": 1920, "height": 1080, "mode": "surfel"}']

The optional fifth argument records the workload the passes were taken on (`_workload`); bench.py attaches a traffic
figure to its roofline object only when that record matches the run.

With the optional fourth file (a separate `--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES` pass) the summary also
holds the vector / scalar / LDS wave-instructions each kernel issued per launch: `valu_insts` / time against the chip's
vector issue rate (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 /s) is the issue utilisation
of a VALU-bound kernel.

Counters are in KiB per dispatch.  On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md §HBM): `fetch_bytes_x2` applies that correction and `hbm_bytes_corrected` = 2 x FETCH + WRITE
is the guide's figure.  The guide leaves other access widths uncalibrated, so profiles/pmc_calib.hip measures them
(profiles/r02/pmc_calibration.json): FETCH_SIZE counts one 64-B unit per REQUEST, a request covers the 64-B halves
of one 128-B line a wave instruction needs — streaming reads (every request a whole line): counter = 1/2 of the bytes;
64-B tile-row pieces and 8-B random probes (half-line requests): counter = the sector bytes exactly; random 48-B records
(1.25 lines, 1.5 sectors each): counter = 0.83 of the sector bytes; WRITE_SIZE is exact at 32-B granularity for every
pattern tried.  `FETCH_FACTOR` below holds the factor of the kernels whose reads are NOT streaming;
`hbm_bytes_calibrated` = factor x FETCH + WRITE.  Only this library's kernels are listed."""
import collections
import csv
import json
import re
import sys


# kernel (short name prefix) -> (fetch factor, why).  Everything else reads wide and coalesced: factor 2.
FETCH_FACTOR = {
    # the blend kernels read 64-B tile-row pieces of the pixel planes (exact) and the per-Gaussian blend records, which
    # are four float4 = 64 B at a 64-B stride (csrc/raster_fwd.hip: rec[4 g + 0..3]): one aligned sector each (exact,
    # like the 8-B probes of the calibration; the 48-B pattern of pmc_calib.hip does not apply to them)
    "blend_bwd_kernel": (1.0, "64-B tile-row pieces + 64-B aligned records: one half-line request each (exact)"),
    "blend_bwd_scan_kernel": (1.0, "as blend_bwd_kernel"),
    "blend_fwd_tile_kernel": (1.0, "64-B aligned records; its pixel traffic is writes"),
    "blend_fwd_wave_kernel": (1.0, "64-B aligned records"),
    "blend_fwd_seg_kernel": (1.0, "64-B aligned records"),
    "blend_fwd_wave_segT_kernel": (1.0, "64-B aligned records"),
    "sdf_forward": (1.0, "32-B block entries / records and 4..128-B row gathers: half-line requests (exact)"),
    "qf_forward_kernel": (1.0, "as sdf_forward"),
    "knn_search_kernel": (1.0, "as sdf_forward"),
}


def fetch_factor(name):
    for k, v in FETCH_FACTOR.items():
        if name.startswith(k):
            return v
    return (2.0, "wide coalesced reads (guide)")


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
                     "write_bytes": int(wk), "hbm_bytes_corrected": int(2 * fk + wk),
                     "fetch_factor": fetch_factor(name)[0], "fetch_factor_why": fetch_factor(name)[1],
                     "hbm_bytes_calibrated": int(fetch_factor(name)[0] * fk + wk)}
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
              f"x2={v['hbm_bytes_corrected']/1e6:9.1f} MB  calibrated={v['hbm_bytes_calibrated']/1e6:9.1f} MB")
    if len(sys.argv) > 5:
        out["_workload"] = json.loads(sys.argv[5])
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
