"""Summarise rocprofv3 --pmc passes (CSV, one counter group per pass; tools/pmc_round.sh) into one JSON.

  python tools/pmc_summarize.py <dir with pmc*/ subdirs> <out.json> [--families]

Per kernel (grouped by demangled name + grid size, i.e. per shape): launches and the mean of every counter per launch
(summed over XCDs / dimensions).  Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is reported in KiB and counts a
wide streaming read at half its bytes on gfx950 -> hbm_read_bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is uncalibrated
there, so both are additionally calibrated on the fp16 add kernel of tools/pmc_kernels.py (known bytes: 2 GiB read,
1 GiB written) when it is part of the pass — the factors are recorded in the output.
--families: also aggregate by the library's kernel family names (the rows of bench.py's `rooflines`).
--calls <pmc_calls.json> (written by tools/pmc_unet_step.py): pair the counter rows of every pass with the wrapper calls
BY DISPATCH ORDER and emit `by_shape` (per family + shape descriptor: launches, HBM read / write bytes per launch,
algorithmic bytes per launch and their ratio) and `families` from the CALL's family — the kernel symbol cannot tell a
3x3 conv from a Linear (both are `gemm2_kernel`), the call list can."""
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
    for key, fam in (("ref_attn", "ref_attn_kernel"), ("temporal_qkv_attn", "gemm_kernel<false>"), ("rowgemm320", "gemm_kernel<false>"),
                     ("temporal_attn", "temporal_attn_kernel"), ("gn_stats", "gn_stats_kernel"), ("gn_scale_shift", "gn_apply_kernel"),
                     ("gn_apply", "gn_apply_kernel"), ("gn_slab", "gn_apply_kernel"), ("layernorm_kernel", "layernorm_kernel"), ("softmax_rows", "softmax_rows_kernel"),
                     ("conv_direct", "conv_small_kernel"), ("conv3x3_c4", "conv_small_kernel"), ("conv_small", "conv_small_kernel"), ("linear_small", "linear_small_kernel"),
                     ("bn_", "batchnorm_kernels"), ("ffn_geglu", "gemm_kernel<false>")):
        if key in name:
            return fam
    if any(k in name for k in ("add_kernel", "window_accumulate", "cfg_ddim", "ncfhw", "nhwc", "u8_to_f16", "f16_to_u8")):
        return "elementwise"
    return None


_GEMM_LIKE = ("gemm_kernel<false>", "gemm_kernel<true> (conv3x3)")
_AUX = ("splitk_reduce", "bn_finalize", "bn_apply")     # second / third launch of one wrapper call


def pair_with_calls(root, calls, cal):
    """per pass (one counter_collection.csv): library dispatches in Dispatch_Id order <-> wrapper calls in launch order.
    Returns {(family, shape): {launches, read, write, abytes, work, unit}} summed over the passes that carry each counter."""
    agg = defaultdict(lambda: dict(n_r=0, n_w=0, read=0.0, write=0.0, abytes=0, work=0, unit="B", launches=0))
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        disp = {}
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                d = disp.setdefault(int(r.get("Dispatch_Id") or r.get("Correlation_Id")), dict(name=r["Kernel_Name"], c=defaultdict(float)))
                d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        rows = [(k, demangle(v["name"]), v["c"]) for k, v in sorted(disp.items())]
        rows = [(k, n, c, family(n)) for k, n, c in rows if family(n) is not None]
        ptr = 0
        per_call = [defaultdict(float) for _ in calls]
        for k, n, c, fam_ in rows:
            if any(a in n for a in _AUX):
                tgt = ptr - 1
            else:
                if ptr >= len(calls):
                    raise SystemExit(f"{f}: more library dispatches than traced calls (dispatch {k} {n})")
                cf = calls[ptr]["family"]
                if not (cf == fam_ or (cf in _GEMM_LIKE and fam_ in _GEMM_LIKE)):
                    raise SystemExit(f"{f}: dispatch {k} {n} ({fam_}) does not pair with call #{ptr} {cf} {calls[ptr]['shape']}")
                tgt = ptr
                ptr += 1
            for cn, v in c.items():
                per_call[tgt][cn] += v
        if ptr != len(calls):
            raise SystemExit(f"{f}: {len(calls) - ptr} traced calls have no dispatch")
        for call, c in zip(calls, per_call):
            a = agg[(call["family"], call["shape"])]
            if "FETCH_SIZE" in c:
                a["read"] += c["FETCH_SIZE"] * 1024 * cal["fetch"]
                a["n_r"] += 1
            if "WRITE_SIZE" in c:
                a["write"] += c["WRITE_SIZE"] * 1024 * cal["write"]
                a["n_w"] += 1
            a["launches"] = max(a["n_r"], a["n_w"])
            a["abytes"], a["work"], a["unit"] = call["algorithmic_bytes"], call["work"], call["unit"]
    return agg


def main():
    root, out = sys.argv[1], sys.argv[2]
    fam = "--families" in sys.argv
    calls_file = sys.argv[sys.argv.index("--calls") + 1] if "--calls" in sys.argv else None
    rows = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # (kernel, grid) -> counter -> [sum, dispatches]
    durs = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # (kernel, grid) -> file (= pass) -> [sum of dispatch ns, dispatches]
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            per_dispatch = defaultdict(float)
            meta = {}
            for r in rd:
                key = (f, r.get("Dispatch_Id") or r.get("Correlation_Id"))
                per_dispatch[(key, r["Counter_Name"])] += float(r["Counter_Value"])
                meta[key] = (r["Kernel_Name"], r.get("Grid_Size", "?"), r.get("Start_Timestamp"), r.get("End_Timestamp"))
            for (key, cname), val in per_dispatch.items():
                kn, grid = meta[key][:2]
                a = rows[(demangle(kn), grid)][cname]
                a[0] += val
                a[1] += 1
            for key, (kn, grid, t0, t1) in meta.items():
                try:
                    ns = float(t1) - float(t0)
                except (TypeError, ValueError):
                    continue
                if ns > 0:
                    d = durs[(demangle(kn), grid)][f]
                    d[0] += ns
                    d[1] += 1
    res = []
    for (kn, grid), ctrs in rows.items():
        launches = max(v[1] for v in ctrs.values())
        res.append(dict(kernel=kn, grid=grid, launches=launches, mean={c: v[0] / v[1] for c, v in ctrs.items()},
                        total={c: v[0] for c, v in ctrs.items()}, family=family(kn)))
        # dispatch duration (kernel-trace timestamps of the same CSV) in the pass that carries GRBM_GUI_ACTIVE: busy cycles per
        # XCD / duration = the clock the kernel actually ran at (the "clock-limited on real data" reading of DESIGN 5.3)
        for f, (ns, n) in durs.get((kn, grid), {}).items():
            if n and "GRBM_GUI_ACTIVE" in ctrs:
                res[-1].setdefault("mean_dispatch_us_by_pass", {})[os.path.basename(os.path.dirname(f))] = ns / n / 1e3
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
        if "GRBM_GUI_ACTIVE" in m and r.get("mean_dispatch_us_by_pass"):
            us = min(r["mean_dispatch_us_by_pass"].values())          # the least perturbed pass
            r["effective_clock_ghz"] = (m["GRBM_GUI_ACTIVE"] / 8.0) / (us * 1e3)
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
    if calls_file:
        with open(calls_file) as fh:
            tr = json.load(fh)
        agg = pair_with_calls(root, tr["calls"], cal)
        shapes = []
        for (fm, shp), a in agg.items():
            rd_ = a["read"] / max(1, a["n_r"])
            wr_ = a["write"] / max(1, a["n_w"])
            shapes.append(dict(family=fm, shape=shp, launches=a["launches"], hbm_read_bytes_per_launch=rd_,
                               hbm_write_bytes_per_launch=wr_, algorithmic_bytes_per_launch=a["abytes"],
                               traffic_over_algorithmic=((rd_ + wr_) / a["abytes"]) if a["abytes"] else None,
                               work_per_launch=a["work"], work_unit=a["unit"]))
        shapes.sort(key=lambda r: -(r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]) * r["launches"])
        outd["by_shape"] = shapes
        fams = defaultdict(lambda: dict(launches=0, read=0.0, write=0.0, ab=0.0))
        for r in shapes:
            f_ = fams[r["family"]]
            f_["launches"] += r["launches"]
            f_["read"] += r["hbm_read_bytes_per_launch"] * r["launches"]
            f_["write"] += r["hbm_write_bytes_per_launch"] * r["launches"]
            f_["ab"] += r["algorithmic_bytes_per_launch"] * r["launches"]
        outd["families"] = {k: dict(launches=v["launches"], read_bytes=v["read"], write_bytes=v["write"],
                                    bytes_per_launch=(v["read"] + v["write"]) / max(1, v["launches"]),
                                    algorithmic_bytes_per_launch=v["ab"] / max(1, v["launches"]),
                                    traffic_over_algorithmic=(v["read"] + v["write"]) / v["ab"] if v["ab"] else None)
                            for k, v in fams.items()}
        outd["paired_with_calls"] = dict(file=os.path.basename(calls_file), forwards=tr.get("forwards"),
                                         measured_at_commit=tr.get("commit"))
    with open(out, "w") as fh:
        json.dump(outd, fh, indent=1)
    print(f"{len(res)} kernel shapes -> {out}; calibration {cal}")


if __name__ == "__main__":
    main()
