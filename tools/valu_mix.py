#!/usr/bin/env python3
"""A kernel's VALU instruction mix priced with MEASURED issue costs: the ceiling that replaces the nominal 4-cycle slot model.

gfx950 issues a wave64 VALU instruction in 2, 4 or 8 SIMD-32 cycles depending on the opcode (tools/valu_microbench,
profiles/valu_costs.json: v_mul / v_add / v_fma_f32 / v_mov / v_and ... ~1.0-1.1 ns per wave-instruction per SIMD with every SIMD
busy; v_pk_* / v_dot4 / v_lshl_add / v_cndmask / v_max / v_cvt / v_cmp ... ~1.75-1.8 ns; v_exp / v_rcp / v_rsq / v_sqrt 3.41 ns).
A model that books 4 cycles for every instruction is therefore not a ceiling (SQ_ACTIVE_INST_VALU x 4 reads > 1 on plain-f32
kernels).  This module prices a kernel by what its own instruction stream costs:

    bare_ns(kernel) = sum over VALU mnemonics m of  n_m x cost_ns(m)          per wave
    ceiling fraction = bare_ns x waves / 1024 SIMDs / launch time             (an estimate: measured class costs, not lower bounds)

n_m comes from the kernel's gfx950 assembly (compiled here with the library's own flags): the static histogram, with the bodies
of loops weighted by ONE common trip factor x chosen so that the total matches the hardware's dynamic count per wave
(SQ_INSTS_VALU / SQ_WAVES); the number of transcendentals this predicts is reported next to SQ_INSTS_VALU_TRANS_F32 as the
check of that weighting.  Without counters (CLI use) x is given or 1.

    python3 tools/valu_mix.py --file ers_kernels.hip --kernel enhance7_pk_kernel [--valu-per-wave 1875 --trans-per-wave 203]
"""
import argparse
import collections
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kinectdepthmapenhancement_amd", "csrc")
COSTS = os.path.join(ROOT, "profiles", "valu_costs.json")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
FILES = ("jbf_fast.hip", "jbf_kernels.hip", "ers_kernels.hip", "dasp_kernels.hip", "spdsr_kernels.hip", "stream_kernels.hip")
_CACHE = os.path.join(ROOT, "tools", ".asm_cache")      # git-ignored, travels to the GPU box with the snapshot: no compile there


class Costs(dict):
    """mnemonic -> ns per wave-instruction per SIMD; .cycles: the same in shader cycles (when the table has them);
    .klass: mnemonic -> the SQ_INSTS_VALU_* class counter that books it (measured: tools/profile_round.sh micro_pmc)"""
    cycles = None
    nominal = None
    klass = None


CLASSES = ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT")


def class_of(m, costs):
    """which SQ_INSTS_VALU_<class> counter books opcode m ("OTHER": none of them -- moves, logic ops, shifts, selects, f32
    compares / min / max, lane ops, f64).  Measured per opcode where the micro-benchmark covers it, by family otherwise."""
    n = norm(m)
    k = getattr(costs, "klass", None) or {}
    if n in k:
        return k[n]
    if n.startswith(TRANS):
        return "TRANS_F32" if n.endswith("f32") else "OTHER"
    if n.startswith("v_cvt_"):
        return "CVT"
    if re.search(r"(f64|f16|u16|i16|b16)(_sdwa|_dpp)?$", n):
        return "OTHER"
    if re.match(r"v_(pk_)?(add|sub|subrev)_f32", n):
        return "ADD_F32"
    if re.match(r"v_(pk_)?mul_f32", n):
        return "MUL_F32"
    if re.match(r"v_(pk_)?(fma|fmac|fmamk|fmaak|mad|mac)_f32|v_div_fmas_f32", n):
        return "FMA_F32"
    if re.search(r"(u64|i64)", n) and not n.startswith(("v_cmp", "v_lshr", "v_lshl", "v_ashr")):
        return "INT64"
    if re.match(r"v_(add|sub|subrev|addc|subb|subbrev|mul|mad|min|max|min3|max3|med3|bfe|dot4|dot2|sad|lshl_add|add_lshl|add3|ashrrev|mbcnt|xad)"
                r"(_co)?(_lo|_hi)?_?(u32|i32|u32_u24|i32_i24|u32_u8|i32_i8|u32_b32|lshl_u32|add_u32)", n) or re.match(r"v_cmp_\w+_(i32|u32)", n):
        return "INT32"
    return "OTHER"


def load_costs(path=COSTS):
    t = json.load(open(path))
    c = Costs({k: v["ns"] for k, v in t["cost"].items()})
    if all("cycles" in v for v in t["cost"].values()):
        c.cycles = {k: v["cycles"] for k, v in t["cost"].items()}
        # the issue rate behind each measured cost: 2 / 4 / 8 / 16 SIMD-32 cycles (the measured 2.3-2.5 / 4.1-4.4 / 8.2 / 16.2 carry the
        # micro-benchmark's own loop overhead and operand-read stalls; a CEILING must not include them)
        c.nominal = {k: min((2.0, 4.0, 8.0, 16.0), key=lambda n: abs(n - v) / n) for k, v in c.cycles.items()}
        c.klass = {}
        for k, v in t["cost"].items():
            booked = [n[len("SQ_INSTS_VALU_"):] for n, x in v.get("counted_by", {}).items() if n.startswith("SQ_INSTS_VALU_") and x > 0.5]
            c.klass[k] = booked[0] if booked else "OTHER"
    return c


def compile_asm(name):
    """gfx950 assembly of one kernel source, device side only, with the library's flags (csrc/Makefile); cached by content"""
    src = os.path.join(CSRC, name)
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f == name or f.endswith(".h"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    os.makedirs(_CACHE, exist_ok=True)
    out = os.path.join(_CACHE, f"{name}.{h.hexdigest()[:16]}.s")
    if not os.path.exists(out):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-fno-gpu-rdc",
                               "--cuda-device-only", "-S", "-o", out + ".tmp", src], stderr=subprocess.DEVNULL)
        os.replace(out + ".tmp", out)
    return open(out).read()


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.splitlines()


def short(name):
    """as tools/pmc_report.short(): the name without return type, namespaces and the argument list"""
    name = re.sub(r"(kde::)?\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def kernels_of(asm):
    """{short demangled name: [instruction mnemonics with loop membership]} for every kernel of one assembly file"""
    lines = asm.splitlines()
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s*(;.*)?$", ln)] if m]
    kern = {m.group(1) for ln in lines for m in [re.match(r"^\s+\.amdhsa_kernel\s+(\S+)", ln)] if m}
    starts = [(i, n) for i, n in starts if n in kern]
    dem = demangle([n for _, n in starts])
    out = {}
    for (i, n), d in zip(starts, dem):
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        out[short(d)] = parse_body(lines[i + 1:end])
    return out


def norm(m):
    """mnemonic as the cost table spells it: no encoding suffix"""
    return re.sub(r"_(e32|e64)$", "", m)


def parse_body(body):
    insts, labels = [], {}
    for ln in body:
        t = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        if re.match(r"^[a-z_0-9]+", t):
            insts.append((t.split()[0], t))
    # loops = backward branches; several branches to one label (continue paths) are ONE loop: its extent runs to the last of them
    extent = {}
    for i, (m, t) in enumerate(insts):
        if m.startswith("s_cbranch") or m == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                extent[labels[tgt]] = max(extent.get(labels[tgt], -1), i)
    depth = [0] * len(insts)            # number of loops enclosing the instruction
    for lo, hi in extent.items():
        for j in range(lo, hi + 1):
            depth[j] += 1
    return [(m, dp) for (m, _), dp in zip(insts, depth)]


def cost_of(m, costs):
    """(ns, known?) of one VALU mnemonic; opcodes the micro-benchmark does not cover take their class's cheapest member so that
    the sum stays a lower bound of the stream's time"""
    n = norm(m)
    if n in costs:
        return costs[n], True
    # families measured through one member (every member tried costs the same: profiles/valu_costs.json)
    for prefix, member in (("v_cmp_class", "v_cmp_class_f32"), ("v_cmp_", "v_cmp_lt_f32"), ("v_cvt_f32_ubyte", "v_cvt_f32_ubyte1"),
                           ("v_mbcnt_", "v_mbcnt_lo_u32_b32"), ("v_max_u32_sdwa", "v_min_u32_sdwa"), ("v_sub_u32_sdwa", "v_add_u32_sdwa"),
                           ("v_min_i32", "v_max_i32"), ("v_subb_co", "v_addc_co_u32"), ("v_subbrev_co", "v_addc_co_u32"),
                           ("v_sub_co", "v_addc_co_u32"), ("v_subrev_co", "v_addc_co_u32"), ("v_add_co", "v_addc_co_u32")):
        if n.startswith(prefix) and member in costs and not re.search(r"(f64|u64|i64|u16|i16|f16)$", n):
            return costs[member], True
    if n.startswith(TRANS):
        return costs["v_exp_f32"], False
    if n.endswith(("_sdwa", "_dpp")) or n.startswith(("v_pk_", "v_cmp", "v_cvt", "v_dot", "v_mad", "v_min", "v_max", "v_med", "v_div", "v_bfe",
                                                       "v_bfi", "v_perm", "v_alignb", "v_readlane", "v_writelane", "v_mbcnt", "v_mul_hi", "v_mul_lo",
                                                       "v_mul_i32", "v_ldexp", "v_frexp", "v_trunc", "v_floor", "v_ceil", "v_cndmask")) \
            or re.search(r"_(f64|u64|i64|b64)$", n) or re.search(r"3_|_add_|_or_b32$", n):
        return costs["v_pk_fma_f32"], False
    return min(costs.values()), False


def estimate(insts, costs, valu_per_wave=None, trans_per_wave=None, classes_per_wave=None):
    """price one kernel.  insts: [(mnemonic, in_loop)] (parse_body).  classes_per_wave: {"ADD_F32": n, ...} = the hardware's
    SQ_INSTS_VALU_<class> counts per wave: the histogram is then scaled CLASS BY CLASS to what the hardware counted (the opcodes
    no class counter books take the rest of SQ_INSTS_VALU), so that only the split between the members of one class is left to
    the static code -- kernels whose branches skip part of the code (K7's pruned candidates) are otherwise mispriced."""
    valu = [(m, int(lp)) for m, lp in insts if m.startswith("v_")]
    maxd = max([d for _, d in valu], default=0)
    hd = [collections.Counter(norm(m) for m, d in valu if d == k) for k in range(maxd + 1)]
    vd = [sum(h.values()) for h in hd]
    v0, v1 = vd[0], sum(vd[1:])
    x, how = 1.0, "static histogram (no loops or no counters)"
    if valu_per_wave and v1 > 0 and valu_per_wave > v0 + v1:
        # one common trip factor x per loop level: an instruction inside d nested loops is weighted x^d, x such that the total
        # is the hardware's dynamic count per wave
        lo, hi = 1.0, 1.0e6
        for _ in range(200):
            x = 0.5 * (lo + hi)
            if sum(v * x ** d for d, v in enumerate(vd)) > valu_per_wave:
                hi = x
            else:
                lo = x
        how = f"an instruction inside d nested loops x {x:.2f}^d so that the total is SQ_INSTS_VALU / SQ_WAVES"
        scale = [x ** d for d in range(maxd + 1)]
    elif valu_per_wave:
        f = valu_per_wave / max(1, v0 + v1)                    # straight-line code with branches: the whole body scaled
        how = f"whole body x {f:.2f} (branches / early exits) so that the total is SQ_INSTS_VALU / SQ_WAVES"
        scale = [f] * (maxd + 1)
    else:
        scale = [1.0] * (maxd + 1)
    mix = collections.Counter()
    for d, h in enumerate(hd):
        for k, v in h.items():
            mix[k] += v * scale[d]
    pred_trans = sum(v for k, v in mix.items() if k.startswith(TRANS))
    by_counter = None
    if valu_per_wave and classes_per_wave:
        stat = collections.Counter()
        for k, v in mix.items():
            stat[class_of(k, costs)] += v
        target = {c: float(classes_per_wave.get(c, 0.0)) for c in CLASSES}
        target["OTHER"] = max(0.0, valu_per_wave - sum(target.values()))
        scaled = collections.Counter()
        for k, v in mix.items():
            c = class_of(k, costs)
            if stat[c] > 0:
                scaled[k] = v * target[c] / stat[c]
        # a class the hardware counted but the static code does not show (or the reverse) is left to the cheapest opcode
        missing = sum(t for c, t in target.items() if stat[c] <= 0)
        if missing > 0:
            scaled["v_mov_b32"] += missing
        by_counter = {c: {"counted": round(target[c], 1), "static_share_scaled_from": round(stat[c], 1)} for c in target if target[c] > 0 or stat[c] > 0}
        mix = scaled
        how += "; then every SQ_INSTS_VALU_<class> scaled to the hardware's count (opcodes no class books: the rest of SQ_INSTS_VALU)"
    elif valu_per_wave and trans_per_wave is not None and pred_trans > 0 and trans_per_wave > 0:
        # the transcendentals are counted by the hardware (SQ_INSTS_VALU_TRANS_F32): take that count for them (split among the
        # transcendental opcodes as in the histogram) and scale the other opcodes to the rest of SQ_INSTS_VALU
        rest = sum(v for k, v in mix.items() if not k.startswith(TRANS))
        ft, fr = trans_per_wave / pred_trans, (valu_per_wave - trans_per_wave) / max(rest, 1e-9)
        mix = collections.Counter({k: v * (ft if k.startswith(TRANS) else fr) for k, v in mix.items()})
        how += "; transcendentals = SQ_INSTS_VALU_TRANS_F32, the other opcodes scaled to the rest"
    total = sum(mix.values())
    bare, bare_cyc, bare_nom, unknown, by_class = 0.0, 0.0, 0.0, collections.Counter(), collections.Counter()
    cyc, nom = getattr(costs, "cycles", None), getattr(costs, "nominal", None)
    for k, v in mix.items():
        c, known = cost_of(k, costs)
        bare += v * c
        if cyc:
            bare_cyc += v * cost_of(k, cyc)[0]
            bare_nom += v * cost_of(k, nom)[0]
        by_class["8-cycle (transcendental)" if c > 2.6 else "4-cycle" if c > 1.4 else "2-cycle"] += v
        if not known:
            unknown[k] += v
    r = {"valu_per_wave": total, "bare_ns_per_wave": bare, **({"bare_cycles_per_wave": bare_nom, "mean_cycles_per_instruction": bare_nom / max(total, 1e-9),
            "bare_cycles_per_wave_at_measured_costs": bare_cyc} if cyc else {}), "mean_ns_per_instruction": bare / max(total, 1e-9), "weighting": how,
         "static_valu_outside_loops": v0, "static_valu_in_loops": v1,
         "class_fractions": {k: v / max(total, 1e-9) for k, v in sorted(by_class.items())},
         "transcendentals_predicted_by_the_loop_weighting": pred_trans,
         "not_in_cost_table_frac": sum(unknown.values()) / max(total, 1e-9),
         "top": [[k, round(v, 1)] for k, v in mix.most_common(10)]}
    if trans_per_wave is not None:
        r["transcendentals_counted_per_wave"] = trans_per_wave
    if by_counter:
        r["classes"] = by_counter
    if unknown:
        r["not_in_cost_table"] = [[k, round(v, 1)] for k, v in unknown.most_common(6)]
    return r


_ALL = None


def all_kernels():
    global _ALL
    if _ALL is None:
        _ALL = {}
        for f in FILES:
            for k, v in kernels_of(compile_asm(f)).items():
                _ALL[k] = v
    return _ALL


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", default="")
    ap.add_argument("--kernel", required=True, help="substring of the demangled kernel name (pmc_report's short form)")
    ap.add_argument("--valu-per-wave", type=float, default=None)
    ap.add_argument("--trans-per-wave", type=float, default=None)
    ap.add_argument("--costs", default=COSTS)
    a = ap.parse_args()
    costs = load_costs(a.costs)
    ks = kernels_of(compile_asm(a.file)) if a.file else all_kernels()
    for name, insts in ks.items():
        if a.kernel in name:
            print(name)
            print(json.dumps(estimate(insts, costs, a.valu_per_wave, a.trans_per_wave), indent=1))


if __name__ == "__main__":
    main()
