#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> one JSON table per (kernel, grid).

    python3 tools/pmc_report.py --dir gpurun_out/<tag>/pmc_bench --out profiles/r02_pmc_bench.json [--algo spec.json]

`--dir` holds one sub-directory per counter pass (tools/profile_round.sh: fetch/, write/, sq1/, sq2/), each with the
*counter_collection.csv files of one `rocprofv3 --pmc ...` run of the SAME command.  Counters are averaged over the
launches of one (kernel name, grid size); derived figures follow /opt/skills/guides/MI355X_MICROARCH.md:

  hbm_bytes           = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  FETCH_SIZE tallies 128-byte requests at 64 bytes on
                        gfx950 for wide coalesced reads; the factor is calibrated in the same run on the library's float4
                        copy kernel (reads exactly what it writes) and reported as `fetch_correction_measured`.
  cycles              = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
  valu_busy           = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles).  NOT a fraction of a ceiling: the counter books one
                        quad-cycle per VALU instruction (two per transcendental), but gfx950 issues v_mul / v_add / v_fma_f32 /
                        v_mov / v_and ... in TWO cycles, so plain-f32 kernels read above 1 (r03: 1.06-1.29).  Kept for comparison.
  valu_slot_frac      = (SQ_INSTS_VALU + SQ_INSTS_VALU_TRANS_F32) x 4 / (1024 x cycles): the same 4-cycle model from counts
  mix (r04)           = the kernel's VALU instruction mix priced with the MEASURED issue cost of every opcode
                        (tools/valu_mix.py, profiles/valu_costs.json from tools/valu_microbench):
                          bare_stream_cycles = sum_m n_m x cycles(m) per wave x SQ_WAVES / 1024 SIMDs -- what the VALU
                                               instructions alone take with every SIMD issuing back to back
                          valu_cycles_frac   = bare_stream_cycles / (GRBM_GUI_ACTIVE / 8): both sides in shader cycles of the SAME
                                               profiled launch, so the clock the chip holds (DVFS) cancels.  cycles(m) = the issue
                                               rate of the opcode's measured class: 2, 4, 8 (16: v_rcp_f64) cycles
                          valu_cycles_frac_at_measured_costs = the same with the micro-benchmark's own figures (2.3-2.5 / 4.1-4.4 /
                                               8.2: they carry its loop overhead and operand-read stalls, so this one can exceed 1)
                          valu_cycles_floor_frac = counters only, no assembly: (2 x non-transcendental + 8 x transcendental
                                               instructions) / 1024 / cycles -- what no mix estimate can fall below
                          valu_ns_frac       = the same with the wall-time costs over End - Start of the launch (the micro-benchmark's
                                               pure streams run at a lower clock than a kernel's mix: can read a few % high)
                        valu_cycles_frac is an ESTIMATE (the class costs are measured costs of pure instruction streams, rounded to
                        2 / 4 / 8 cycles -- not lower bounds: dual issue or a different clock can move it past 1); it is the
                        "fraction of the VALU-issue ceiling" DESIGN.md quotes per kernel.
  waves_per_simd      = SQ_WAVE_CYCLES x 4 / (1024 x cycles)                    (mean resident waves per SIMD)
  lds_conflict_frac   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
`--algo` optionally maps a kernel-name substring to algorithmic bytes per launch so that traffic ratios are in the file."""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _code_only(text):
    """the text without comments and with runs of whitespace collapsed: a comment edit does not make a profile stale"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":                       # string / character literal: copied verbatim
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    return re.sub(r"\s+", " ", "".join(out)).strip()


def source_hash():
    """sha256 over the kernel sources (code only, comments and layout ignored): bench.py only quotes a profile taken on
    exactly this code"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kinectdepthmapenhancement_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(_code_only(open(os.path.join(d, name), errors="replace").read()).encode())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"(kde::)?\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--algo", default="", help="JSON: {kernel-name substring: algorithmic bytes per launch}")
    ap.add_argument("--command", default="", help="the profiled command, recorded in the file")
    ap.add_argument("--keep", default="jbf|presmooth|mrf|copy_kernel|bgr3_copy|enhance|edge_|calc_ld|analyze|sample_clusters|p2r|points_map|buf_|spdsr|moments|plane|jacobi",
                    help="regex of kernel names to keep")
    a = ap.parse_args()
    keep = re.compile(a.keep)
    algo = json.load(open(a.algo)) if a.algo else {}

    table = defaultdict(lambda: {"counters": defaultdict(list), "meta": {}, "dur": []})
    try:
        import valu_mix
        costs, kern_asm = valu_mix.load_costs(), valu_mix.all_kernels()
    except Exception as ex:                     # no cost table / no compiler: the table is written without the mix
        print(f"pmc_report: no instruction-mix pricing ({type(ex).__name__}: {ex})", file=sys.stderr)
        costs, kern_asm = None, {}
    for path in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if not keep.search(name):
                    continue
                key = (short(name), int(row["Grid_Size"]))
                e = table[key]
                e["counters"][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row.get("End_Timestamp") and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    e["dur"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                e["meta"] = {"workgroup": int(row.get("Workgroup_Size", 0) or 0), "lds_bytes": int(row.get("LDS_Block_Size", 0) or 0),
                             "vgprs": int(row.get("VGPR_Count", 0) or 0), "sgprs": int(row.get("SGPR_Count", 0) or 0)}

    out = {"source": "rocprofv3 --pmc, one counter group per pass (tools/profile_round.sh); MI355X",
           "note": "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE: the factor 2 (128-byte requests tallied at 64 bytes on gfx950) is "
                   "calibrated in this run on a float4 copy (fetch_correction_measured); for kernels with narrower loads it is "
                   "not separately calibrated -- K1's dword / byte tile loads reproduce their algorithmic 7 B/pixel with it. "
                   "Single-frame working sets are cache-resident: their figures count fabric requests, not DRAM traffic.",
           "command": a.command, "kernel_source_sha16": source_hash(), "kernels": []}
    corr = None
    for (name, grid), e in sorted(table.items()):
        c = {k: sum(v) / len(v) for k, v in e["counters"].items()}
        d = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            d["fetch_bytes_x2"] = 2.0 * c["FETCH_SIZE"] * 1024.0
            d["write_bytes"] = c["WRITE_SIZE"] * 1024.0
            if "copy_kernel" in name and c["FETCH_SIZE"] > 1e5:
                corr = c["WRITE_SIZE"] / c["FETCH_SIZE"]
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0:
            d["cycles"] = cyc
            if "SQ_ACTIVE_INST_VALU" in c:
                d["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
            if "SQ_INSTS_VALU" in c:
                tr = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
                d["valu_slot_frac"] = (c["SQ_INSTS_VALU"] + tr) * 4.0 / (1024.0 * cyc)
                if c.get("SQ_WAVES"):
                    d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
                    d["trans_insts_per_wave"] = tr / c["SQ_WAVES"]
            if "SQ_WAVE_CYCLES" in c:
                d["waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc)
            if "SQ_ACTIVE_INST_LDS" in c:
                d["lds_inst_busy"] = c["SQ_ACTIVE_INST_LDS"] * 4.0 / (1024.0 * cyc)
        if e["dur"]:
            d["duration_ns"] = sum(e["dur"]) / len(e["dur"])
        mix = None
        if costs and name in kern_asm and c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
            cls = {k: c["SQ_INSTS_VALU_" + k] / c["SQ_WAVES"] for k in valu_mix.CLASSES if "SQ_INSTS_VALU_" + k in c}
            mix = valu_mix.estimate(kern_asm[name], costs, c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / c["SQ_WAVES"],
                                    cls if len(cls) == len(valu_mix.CLASSES) else None)
            mix["bare_stream_ns"] = mix["bare_ns_per_wave"] * c["SQ_WAVES"] / 1024.0
            if d.get("duration_ns"):
                d["valu_ns_frac"] = mix["bare_stream_ns"] / d["duration_ns"]
            if "bare_cycles_per_wave" in mix and cyc > 0:
                mix["bare_stream_cycles"] = mix["bare_cycles_per_wave"] * c["SQ_WAVES"] / 1024.0
                d["valu_cycles_frac"] = mix["bare_stream_cycles"] / cyc
                d["valu_cycles_frac_at_measured_costs"] = mix["bare_cycles_per_wave_at_measured_costs"] * c["SQ_WAVES"] / 1024.0 / cyc
                # rigorous floor, counters only: every instruction at the cheapest rate (2 cycles), the hardware-counted
                # transcendentals at 8
                tr_ = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
                d["valu_cycles_floor_frac"] = (2.0 * (c["SQ_INSTS_VALU"] - tr_) + 8.0 * tr_) / 1024.0 / cyc
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        for sub, nbytes in algo.items():
            key, _, g = sub.partition("@")
            if key in name and (not g or int(g) == grid):
                d["algorithmic_bytes"] = float(nbytes)
                if "hbm_bytes" in d:
                    d["traffic_ratio"] = d["hbm_bytes"] / float(nbytes)
        out["kernels"].append({"kernel": name, "grid": grid, **e["meta"],
                               "launches": max(len(v) for v in e["counters"].values()),
                               "counters": {k: round(v, 2) for k, v in sorted(c.items())},
                               "derived": {k: (round(v, 5) if abs(v) < 100 else round(v, 1)) for k, v in d.items()},
                               **({"mix": mix} if mix else {})})
    out["fetch_correction_measured"] = corr
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for k in out["kernels"]:
        print(f'{k["kernel"][:70]:70s} grid {k["grid"]:>9d}  ' + "  ".join(f"{n}={v}" for n, v in k["derived"].items()), file=sys.stderr)


if __name__ == "__main__":
    main()
