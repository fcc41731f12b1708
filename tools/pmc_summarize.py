"""Summarise rocprofv3 --pmc passes (CSV, one counter group per pass; tools/pmc_round.sh) into one JSON.

  python tools/pmc_summarize.py <dir with pmc*/ subdirs> <out.json> [--families]

Per kernel (grouped by demangled name + grid size, i.e. per shape): launches and the mean of every counter per launch
(summed over XCDs / dimensions).  Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is reported in KiB and counts a
wide streaming read at half its bytes on gfx950 -> hbm_read_bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is uncalibrated
there, so both are additionally calibrated on the fp16 add kernel of tools/pmc_kernels.py (known bytes: 2 GiB read,
1 GiB written) when it is part of the pass — the factors are recorded in the output.
--families: also aggregate by the library's kernel family names (the rows of bench.py's `rooflines`)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict


def demangle(n):
    if not n.startswith("_Z"):
        return n
    try:
        out = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        return n
    out = re.sub(r"\(anonymous namespace\)::", "", out or n)
    out = re.sub(r"^void ", "", out)
    return re.sub(r"\(.*\)$", "", out)


def family(name):
    if "gemm2_kernel" in name or "gemm_small_kernel" in name:
        m = re.search(r"gemm2_kernel<[^>]*?(true|false), (true|false)>", name)
        conv = (m.group(1) == "true") if m else ("gemm_small_kernel<true>" in name)
        return "gemm_kernel<true> (conv3x3)" if conv else "gemm_kernel<false>"
    if "splitk_reduce" in name:
        return "gemm_kernel<false>"     # shared by the conv and the linear split-K; it carries no contraction
    for key, fam in (("ref_attn", "ref_attn_kernel"), ("temporal_attn", "temporal_attn_kernel"), ("gn_stats", "gn_stats_kernel"),
                     ("gn_apply", "gn_apply_kernel"), ("layernorm_kernel", "layernorm_kernel"), ("softmax_rows", "softmax_rows_kernel"),
                     ("conv_direct", "conv_small_kernel"), ("conv_small", "conv_small_kernel"), ("linear_small", "linear_small_kernel"),
                     ("bn_", "batchnorm_kernels"), ("ffn_geglu", "gemm_kernel<false>")):
        if key in name:
            return fam
    if any(k in name for k in ("add_kernel", "window_accumulate", "cfg_ddim", "ncfhw", "nhwc", "u8_to_f16")):
        return "elementwise"
    return None


def main():
    root, out = sys.argv[1], sys.argv[2]
    fam = "--families" in sys.argv
    rows = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # (kernel, grid) -> counter -> [sum, dispatches]
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            per_dispatch = defaultdict(float)
            meta = {}
            for r in rd:
                key = (f, r.get("Dispatch_Id") or r.get("Correlation_Id"))
                per_dispatch[(key, r["Counter_Name"])] += float(r["Counter_Value"])
                meta[key] = (r["Kernel_Name"], r.get("Grid_Size", "?"))
            for (key, cname), val in per_dispatch.items():
                kn, grid = meta[key]
                a = rows[(demangle(kn), grid)][cname]
                a[0] += val
                a[1] += 1
    res = []
    for (kn, grid), ctrs in rows.items():
        launches = max(v[1] for v in ctrs.values())
        res.append(dict(kernel=kn, grid=grid, launches=launches, mean={c: v[0] / v[1] for c, v in ctrs.items()},
                        total={c: v[0] for c, v in ctrs.items()}, family=family(kn)))
    # calibration on the 1-GiB fp16 add
    cal = dict(fetch=2.0, write=1.0, source="guide default (FETCH_SIZE x2, WRITE_SIZE as reported)")
    for r in res:
        if "add_kernel" in r["kernel"] and r["mean"].get("FETCH_SIZE", 0) > 1e5:
            known_r, known_w = 2.0 * 2**30, 1.0 * 2**30
            cal["fetch"] = known_r / (r["mean"]["FETCH_SIZE"] * 1024)
            cal["source"] = "calibrated on add_kernel (2 GiB read / 1 GiB written per launch)"
        if "add_kernel" in r["kernel"] and r["mean"].get("WRITE_SIZE", 0) > 1e5:
            cal["write"] = 2**30 / (r["mean"]["WRITE_SIZE"] * 1024)
    for r in res:
        m = r["mean"]
        if "FETCH_SIZE" in m:
            r["hbm_read_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * cal["fetch"]
        if "WRITE_SIZE" in m:
            r["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024 * cal["write"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
            # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
            r["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
            r["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0)
    res.sort(key=lambda r: -r["total"].get("GRBM_GUI_ACTIVE", r["total"].get("FETCH_SIZE", 0)))
    outd = dict(calibration=cal, kernels=res)
    if fam:
        fams = defaultdict(lambda: dict(launches=0, read=0.0, write=0.0))
        for r in res:
            if r["family"] is None:
                continue
            f_ = fams[r["family"]]
            f_["launches"] = max(f_["launches"], 0) + (r["launches"] if "splitk" not in r["kernel"] else 0)
            f_["read"] += r.get("hbm_read_bytes_per_launch", 0.0) * r["launches"]
            f_["write"] += r.get("hbm_write_bytes_per_launch", 0.0) * r["launches"]
        outd["families"] = {k: dict(launches=v["launches"], read_bytes=v["read"], write_bytes=v["write"],
                                    bytes_per_launch=(v["read"] + v["write"]) / max(1, v["launches"])) for k, v in fams.items()}
    with open(out, "w") as fh:
        json.dump(outd, fh, indent=1)
    print(f"{len(res)} kernel shapes -> {out}; calibration {cal}")


if __name__ == "__main__":
    main()
