#!/usr/bin/env python3
"""ISA-level account of a kernel's inner loops: compiles one .hip file to gfx950 assembly (device side only), finds the
loops of the requested kernel (a label that a later s_cbranch jumps back to) and prints, per loop, the instruction
histogram by class and the nominal issue slots of r01-r03's model (one 4-cycle slot per VALU instruction, two for a
transcendental).  Since r04 the ceiling DESIGN.md / bench.py quote is NOT this model: tools/valu_mix.py prices the same
histogram with the issue rate of every opcode's measured class (2 / 4 / 8 cycles, profiles/valu_costs.json) and scales it to
the hardware's instruction-class counters; this tool stays for the per-loop view (instructions per unit, LDS / SALU / nop share).

    python3 tools/isa_slots.py --file jbf_fast.hip --kernel 'jbf_pk_kernelILi11ELi2ELi16ELi16ELb0ELb1ELb0' \
            --units-per-trip 22 > profiles/k1_w11_isa_slots.txt

--units-per-trip: "units" one loop trip processes (K1: window x pixel pairs per thread = one row of the window for each
pair; a unit is one tap for each pixel of a pair), to print slots per unit."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kinectdepthmapenhancement_amd", "csrc")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(m):
    if m.startswith(TRANS):
        return "valu-transcendental"
    if m.startswith("v_pk_"):
        return "valu-packed-f32"
    if m.startswith("v_dot"):
        return "valu-dot4"
    if m.startswith(("v_lshl_add", "v_add_u32", "v_sub_u32", "v_lshlrev", "v_and_", "v_or_", "v_mad_u", "v_add_co", "v_bfe", "v_perm")):
        return "valu-int"
    if m.startswith("v_"):
        return "valu-other"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt"):
        return "s_waitcnt"
    if m.startswith("s_nop"):
        return "s_nop"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", required=True, help="source under kinectdepthmapenhancement_amd/csrc")
    ap.add_argument("--kernel", required=True, help="substring of the mangled kernel name")
    ap.add_argument("--units-per-trip", type=float, default=0.0)
    ap.add_argument("--min-insts", type=int, default=40, help="ignore loops shorter than this")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-fno-gpu-rdc",
               "--cuda-device-only", "-S", "-o", asm, os.path.join(CSRC, a.file)]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    # the function body: from "<name>:" to the matching .Lfunc_end
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(a.kernel) + r"\w*:", ln))
    name = lines[start][:-1]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    meta = {}
    for key in (".vgpr_count", ".sgpr_count", ".group_segment_fixed_size"):
        for i, ln in enumerate(lines):
            if ln.strip().startswith(".name:") and name in ln:
                for l2 in lines[i - 30:i + 30]:
                    if l2.strip().startswith(key + ":"):
                        meta[key] = l2.split(":")[1].strip()
    insts = []          # (index, mnemonic, text)
    labels = {}
    for ln in body:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if re.match(r"^[a-z_0-9]+", t):
            insts.append((len(insts), t.split()[0], t))
    print(f"# {name}")
    print(f"# {a.file}: {len(insts)} instructions; vgprs {meta.get('.vgpr_count')}, sgprs {meta.get('.sgpr_count')}, "
          f"LDS {meta.get('.group_segment_fixed_size')} B")
    print("# nominal slot model of r01-r03: 1 slot (4 cycles per SIMD) per VALU instruction, 2 per transcendental.  Measured issue")
    print("# costs (profiles/valu_costs.json, r04): v_pk_* / v_dot4 / v_lshl_add / v_cndmask / v_cmp / v_cvt 4.1-4.4 cycles, v_exp / v_rcp 8.2,")
    print("# v_mul / v_add / v_fma_f32 / v_mov / v_and / v_add_u32 2.3-2.5 -- tools/valu_mix.py prices a kernel with their issue rates 4 / 8 / 2")
    loops = []
    for i, m, t in insts:
        if m.startswith("s_cbranch") or m == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    # keep innermost/outermost distinct loops long enough to matter
    for lo, hi in sorted(set(loops)):
        n = hi - lo + 1
        if n < a.min_insts:
            continue
        hist = collections.Counter(classify(m) for _, m, _ in insts[lo:hi + 1])
        mn = collections.Counter(m for _, m, _ in insts[lo:hi + 1] if m.startswith("v_"))
        valu = sum(v for k, v in hist.items() if k.startswith("valu"))
        slots = valu + hist["valu-transcendental"]
        print(f"\nloop: instructions {lo}..{hi} ({n} per trip)")
        for k in sorted(hist):
            print(f"    {k:22s} {hist[k]:6d}")
        print(f"    VALU instructions {valu}, issue slots {slots} (transcendentals twice)")
        if a.units_per_trip:
            u = a.units_per_trip
            print(f"    per unit ({u:g} units per trip): {valu / u:.2f} VALU instructions, {slots / u:.2f} slots, "
                  f"{hist['lds'] / u:.2f} LDS, {hist['salu'] / u:.2f} SALU, {(hist['s_nop'] + hist['s_waitcnt']) / u:.2f} nop/wait")
        print("    VALU mnemonics: " + ", ".join(f"{m} x{c}" for m, c in mn.most_common(14)))


if __name__ == "__main__":
    main()
