#!/usr/bin/env python3
"""tools/valu_microbench output (+ its PMC passes) -> the VALU issue-cost table (profiles/valu_costs.json).

    python3 tools/valu_costs.py gpurun_out/<tag>/valu_microbench.txt [--pmc-dir gpurun_out/<tag>/micro_pmc] > profiles/valu_costs.json

Every row of the micro-benchmark whose label is ONE mnemonic is a pure kind: every SIMD of the chip runs the same independent
stream of that opcode at 1..8 waves per SIMD.
  ns      launch wall time per wave-instruction per SIMD, minimum over the occupancies.  Includes the clock the chip holds under
          THAT load (DVFS): a pure v_pk_fma stream draws more power than a kernel's mix, so a kernel can beat a sum of ns costs.
  cycles  (with --pmc-dir: tools/profile_round.sh micro_pmc) GRBM_GUI_ACTIVE / 8 shader cycles of the launch over SQ_INSTS_VALU /
          1024 SIMDs, minimum over the occupancies: the cost in CYCLES, free of the clock.  This is what the ceiling uses:
              bare cycles of a kernel = sum_m n_m x cycles(m) x waves / 1024   <=   GRBM_GUI_ACTIVE / 8 of its launch.
  counted_by  which SQ_INSTS_VALU_* class counter books the opcode (instructions it adds per wave-instruction).
Mixture rows are kept as `mixtures` with the cost the pure rows predict next to the measured one (costs add; a transcendental
overlaps a little with packed math; a 2-cycle opcode next to 4-cycle ones loses part of its advantage)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

COLS = [1, 2, 3, 4, 5, 6, 8]


def parse(path):
    kinds, label = {}, None
    for ln in open(path):
        m = re.match(r"^(.{50}) ticks(.*)$", ln)
        if m:
            label = m.group(1).strip()
            continue
        m = re.match(r"^\s+ns\s+(.*?)\s+\(ticks/ns", ln)
        if m and label:
            kinds[label] = [float(v) for v in m.group(1).split()]
            label = None
    return kinds


def pmc(pmc_dir, labels):
    """per kind: cycles per wave-instruction per SIMD at every occupancy, and the class counters per instruction"""
    disp = defaultdict(lambda: defaultdict(dict))       # kind index -> dispatch id -> counter -> value
    for path in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        tag = os.path.basename(os.path.dirname(path))
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                m = re.search(r"bench<(?:\(Kind\))?(\d+)>", row["Kernel_Name"])
                if not m:
                    continue
                d = disp[int(m.group(1))][(tag, int(row["Dispatch_Id"]))]
                d[row["Counter_Name"]] = float(row["Counter_Value"])
                d["grid"] = int(row["Grid_Size"])
    out = {}
    for k, ds in disp.items():
        if k >= len(labels):
            continue
        cyc, classes = [], defaultdict(list)
        for (tag, _), c in ds.items():
            insts = c.get("SQ_INSTS_VALU", 0.0)
            if insts < 1e8:             # the short warm-up launch of each (kind, occupancy)
                continue
            if "GRBM_GUI_ACTIVE" in c:
                cyc.append(c["GRBM_GUI_ACTIVE"] / 8.0 / (insts / 1024.0))
            for name, v in c.items():
                if name.startswith("SQ_INSTS_VALU_") or name == "SQ_ACTIVE_INST_VALU":
                    classes[name].append(v / insts)
        if cyc:
            out[labels[k]] = {"cycles": min(cyc), "cycles_all": sorted(round(v, 3) for v in cyc),
                              "counted_by": {n: round(sum(v) / len(v), 3) for n, v in sorted(classes.items()) if sum(v) / len(v) > 0.01}}
    return out


def table(path, pmc_dir=None):
    kinds = parse(path)
    labels = list(kinds)
    pure = {k: v for k, v in kinds.items() if re.fullmatch(r"v_[a-z0-9_]+", k)}
    cost = {k: {"ns": min(v), "ns_at_4_waves": v[COLS.index(4)], "ns_by_waves_per_simd": dict(zip(map(str, COLS), v))} for k, v in pure.items()}
    pm = pmc(pmc_dir, labels) if pmc_dir else {}
    for k, c in cost.items():
        if k in pm:
            c.update(pm[k])
        # classes by measured cost (nominal: 2 / 4 / 8 cycles of a SIMD-32; at 2.4 GHz 0.83 / 1.67 / 3.33 ns)
        ref = c.get("cycles")
        c["class"] = (("half-rate (8-cycle)" if ref > 6.0 else "full (4-cycle)" if ref > 3.0 else "dual (2-cycle)") if ref is not None else
                      ("half-rate (8-cycle)" if c["ns"] > 2.6 else "full (4-cycle)" if c["ns"] > 1.4 else "dual (2-cycle)"))
    # v_cndmask_b32 with the condition in VCC, issued back to back with NO instruction writing VCC in between, measures 22 cycles
    # (5 x the SGPR-pair form); next to the v_cmp that feeds it -- the only way the kernels use it -- the pair costs 4.3 cycles
    # per instruction (row "v_cmp_lt_f32 + v_cndmask_b32").  The pure row is an artefact of the stream, not an issue cost:
    # it is kept aside and the opcode is priced like its e64 form
    anomalies = {}
    if "v_cndmask_b32" in cost and "v_cndmask_b32_e64" in cost and cost["v_cndmask_b32"]["ns"] > 2.5 * cost["v_cndmask_b32_e64"]["ns"]:
        anomalies["v_cndmask_b32 (condition in VCC, no VCC writer in the stream)"] = cost.pop("v_cndmask_b32")
        pair = kinds.get("v_cmp_lt_f32 + v_cndmask_b32 (per instruction)")
        cost["v_cndmask_b32"] = dict(cost["v_cndmask_b32_e64"], note="priced as v_cndmask_b32_e64"
                                     + (f"; v_cmp + v_cndmask pair measured {min(pair):.3f} ns per instruction" if pair else ""))
    mixtures = {}
    parts = {"v_exp_f32": "v_exp_f32", "v_pk_fma_f32": "v_pk_fma_f32", "v_pk_fma": "v_pk_fma_f32", "v_fma_f32": "v_fma_f32", "v_add_u32": "v_add_u32",
             "v_dot4": "v_dot4_u32_u8", "dot4": "v_dot4_u32_u8", "v_lshl_add": "v_lshl_add_u32", "lshl_add": "v_lshl_add_u32",
             "exp": "v_exp_f32", "pk": "v_pk_fma_f32"}
    for k, v in kinds.items():
        if k in pure or "per instruction" in k:
            continue
        terms = re.findall(r"(\d+) ([a-z_0-9]+)", k)
        if terms and all(t in parts and parts[t] in cost for _, t in terms):
            n = sum(int(c) for c, _ in terms)
            m = {"measured_ns_per_instruction": min(v), "sum_of_pure_costs_ns": sum(int(c) * cost[parts[t]]["ns"] for c, t in terms) / n}
            m["measured_over_sum"] = m["measured_ns_per_instruction"] / m["sum_of_pure_costs_ns"]
            if k in pm and all("cycles" in cost[parts[t]] for _, t in terms):
                m["measured_cycles_per_instruction"] = pm[k]["cycles"]
                m["sum_of_pure_costs_cycles"] = sum(int(c) * cost[parts[t]]["cycles"] for c, t in terms) / n
            mixtures[k] = m
    return {"source": "tools/valu_microbench on MI355X (gfx950): every SIMD of the chip runs the same independent stream of one opcode",
            "definition": "ns = launch wall time per wave64 instruction per SIMD, min over 1..8 waves per SIMD (clock-dependent); cycles = "
                          "GRBM_GUI_ACTIVE / 8 over SQ_INSTS_VALU / 1024 of the same launches under rocprofv3 --pmc (clock-free; the ceiling "
                          "uses these); nominal SIMD-32 issue is 2 / 4 / 8 cycles",
            "file": path, "cost": cost, "mixtures": mixtures, "anomalies": anomalies}


if __name__ == "__main__":
    args = sys.argv[1:]
    pmc_dir = args[args.index("--pmc-dir") + 1] if "--pmc-dir" in args else None
    t = table(args[0], pmc_dir)
    if "--table" in args:
        for k, c in sorted(t["cost"].items(), key=lambda kv: kv[1].get("cycles", kv[1]["ns"])):
            cy = f"{c['cycles']:.2f} cycles  " if "cycles" in c else ""
            print(f"{k:24s} {cy}{c['ns']:.3f} ns  (4 waves {c['ns_at_4_waves']:.3f})  {c['class']}  {c.get('counted_by', '')}")
        for k, m in t["mixtures"].items():
            print(f"{k:52s} measured {m['measured_ns_per_instruction']:.3f} ns  sum of pure {m['sum_of_pure_costs_ns']:.3f}  ratio {m['measured_over_sum']:.3f}"
                  + (f"  | {m['measured_cycles_per_instruction']:.2f} vs {m['sum_of_pure_costs_cycles']:.2f} cycles" if "measured_cycles_per_instruction" in m else ""))
    else:
        json.dump(t, sys.stdout, indent=1)
