#!/usr/bin/env python3
"""ISA-level breakdown of the update kernel: static instruction counts of one k_fuse instantiation by class
(VALU / SALU / VMEM / LDS / wait) and by PHASE, attributed through the line tables of a `-gline-tables-only` device assembly:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --cuda-device-only \\
          -o /tmp/khr_dev_g.s khronos_amd/csrc/khronos_amd.hip
    python tools/isa_breakdown.py /tmp/khr_dev_g.s [--kernel SUBSTRING] [--phases-file khr_kernels_fuse.h]

Phases are source-line ranges of khronos_amd/csrc/khr_kernels_fuse.h, found by the marker comments `// isa:<phase>` in the kernel
(every line from a marker to the next one belongs to that phase); instructions from other files (khr_device.h helpers, HIP
headers) are attributed to the phase of the last kernel-file line seen before them (inlined callees).  Static counts: the item
loop is straight-line code with the ZR z-steps unrolled, so (loop-body count) / ZR is the per-z-step figure the PMC counter
SQ_INSTS_VALU integrates dynamically."""
import argparse
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_cmp", "v_cmpx")):
            return "VALU cmp"
        if op.startswith(("v_pk_",)):
            return "VALU packed"
        if op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac", "v_fmac", "v_mad")):
            return "VALU f32 arith"
        if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log")):
            return "VALU transcendental"
        if op.startswith(("v_cndmask", "v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_perm", "v_swap")):
            return "VALU move/select"
        if op.startswith(("v_cvt", "v_fract", "v_floor", "v_trunc", "v_rndne")):
            return "VALU convert"
        if op.startswith(("v_min", "v_max", "v_med3")):
            return "VALU min/max"
        return "VALU int/other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait/nop"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default="k_fuseILi16ELi4ELb1ELb1ELi12ELb0E")
    ap.add_argument("--source", default="khronos_amd/csrc/khr_kernels_fuse.h")
    ap.add_argument("--file-name", default="khr_kernels_fuse.h")
    a = ap.parse_args()
    # phase markers in the source
    marks = []
    for i, ln in enumerate(open(a.source), 1):
        m = re.search(r"//\s*isa:([a-z0-9_/ +-]+)", ln)
        if m:
            marks.append((i, m.group(1).strip()))
    if not marks:
        sys.exit("no `// isa:<phase>` markers in %s" % a.source)

    def phase_of(line):
        ph = "prologue/other"
        for ln, name in marks:
            if line >= ln:
                ph = name
        return ph
    files = {}
    in_kernel = False
    cur_phase = "prologue/other"
    counts = collections.Counter()
    by_phase = collections.defaultdict(collections.Counter)
    n_total = 0
    sym = None
    for ln in open(a.asm):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", ln)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            in_kernel = a.kernel in m.group(1) and "k_fuse2" not in m.group(1)
            sym = m.group(1) if in_kernel else sym
            continue
        if not in_kernel:
            continue
        if re.match(r"\s*\.(Lfunc_end|section|end_amdhsa_kernel|size)", ln) and "Lfunc_end" in ln:
            in_kernel = False
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            # helper functions defined above the first marker (rcpRefined, divExact, interpPixels ...) and everything from
            # other files are inlined callees: they stay with the phase of the call site
            if files.get(int(m.group(1))) == a.file_name and int(m.group(2)) >= marks[0][0]:
                cur_phase = phase_of(int(m.group(2)))
            continue
        m = re.match(r"\s+([a-z_0-9]+)\s", ln)
        if not m or ln.lstrip().startswith((".", ";")):
            continue
        op = m.group(1)
        cls = classify(op)
        counts[cls] += 1
        by_phase[cur_phase][cls] += 1
        n_total += 1
    print("kernel %s: %d instructions (static)" % (sym, n_total))
    classes = sorted(counts, key=lambda c: -counts[c])
    print("%-34s %6s   %s" % ("phase", "total", "  ".join("%s" % c for c in classes)))
    order = ["prologue/other"] + [n for _, n in marks]
    seen = set()
    for ph in order:
        if ph in seen or ph not in by_phase:
            continue
        seen.add(ph)
        c = by_phase[ph]
        valu = sum(v for k, v in c.items() if k.startswith("VALU"))
        print("%-34s %6d   VALU %5d | %s" % (ph, sum(c.values()), valu, "  ".join("%s %d" % (k, c[k]) for k in classes if c[k])))
    valu = sum(v for k, v in counts.items() if k.startswith("VALU"))
    print("%-34s %6d   VALU %5d" % ("TOTAL", n_total, valu))


if __name__ == "__main__":
    main()
