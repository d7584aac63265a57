#!/usr/bin/env python3
"""tools/valu_microbench output -> the VALU issue-cost table (profiles/valu_costs.json).

    python3 tools/valu_costs.py gpurun_out/<tag>/valu_microbench.txt > profiles/valu_costs.json

Every row of the micro-benchmark whose label is ONE mnemonic is a pure kind: the launch wall time per wave-instruction per
SIMD, with every SIMD of the chip running the same independent stream, at 1..8 waves per SIMD (DVFS included: this is what
the chip delivers under that load, not a nominal cycle count).  `ns` = the minimum over the occupancies -- the cheapest the
instruction gets -- so that a ceiling built from it is not exceeded: time >= sum_i count_i x ns_i / SIMDs.
Mixture rows are kept as `mixtures` with the cost the pure rows predict next to the measured one (costs add; a
transcendental overlaps a little with packed math)."""
import json
import re
import sys

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


def table(path):
    kinds = parse(path)
    pure = {k: v for k, v in kinds.items() if re.fullmatch(r"v_[a-z0-9_]+", k)}
    cost = {k: {"ns": min(v), "ns_at_4_waves": v[COLS.index(4)], "ns_by_waves_per_simd": dict(zip(map(str, COLS), v))} for k, v in pure.items()}
    # classes by measured cost (nominal: 2 / 4 / 8 cycles of a 2.4 GHz SIMD-32 = 0.83 / 1.67 / 3.33 ns)
    for k, c in cost.items():
        c["class"] = "half-rate (8-cycle)" if c["ns"] > 2.6 else "full (4-cycle)" if c["ns"] > 1.4 else "dual (2-cycle)"
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
            pred = sum(int(c) * cost[parts[t]]["ns"] for c, t in terms) / n
            mixtures[k] = {"measured_ns_per_instruction": min(v), "sum_of_pure_costs_ns": pred, "measured_over_sum": min(v) / pred}
    return {"source": "tools/valu_microbench on MI355X (gfx950), launch wall time per wave64 instruction per SIMD, all 1024 SIMDs busy",
            "definition": "ns = min over 1..8 waves per SIMD; nominal 2 / 4 / 8 SIMD-32 cycles at 2.4 GHz are 0.83 / 1.67 / 3.33 ns",
            "file": path, "cost": cost, "mixtures": mixtures}


if __name__ == "__main__":
    t = table(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "--table":
        for k, c in sorted(t["cost"].items(), key=lambda kv: kv[1]["ns"]):
            print(f"{k:24s} {c['ns']:.3f} ns  (4 waves {c['ns_at_4_waves']:.3f})  {c['class']}")
        for k, m in t["mixtures"].items():
            print(f"{k:52s} measured {m['measured_ns_per_instruction']:.3f}  sum of pure {m['sum_of_pure_costs_ns']:.3f}  ratio {m['measured_over_sum']:.3f}")
    else:
        json.dump(t, sys.stdout, indent=1)
