#!/usr/bin/env python3
"""tools/isa_hist.py — instruction histogram of one gfx950 kernel from a `-save-temps` assembly file
(or `llvm-objdump -d` output).  Used for the ISA audits committed under profiles/ (r03_isa_*.txt).

  python tools/isa_hist.py FILE.s KERNEL_SUBSTRING [--loop] [--per N] [--dump]

--loop   restrict to the innermost backward-branch loop (micro-benchmarks); default = whole kernel
--per N  also print every count divided by N (e.g. butterflies per thread: 8 stages x 8 = 64)
--dump   print the instructions themselves after the histogram
--split-asm  two histograms: the instructions inside inline-asm statements (the hand-written butterfly cores, bracketed
         by ;;#ASMSTART / ;;#ASMEND in a -S listing) and everything the compiler wrote around them
"""
import re
import sys
from collections import Counter

CLASSES = [
    ("mad_u64_u32", r"^v_mad_u64_u32"),
    ("mul_hi/lo_u32", r"^v_mul_(hi|lo)_u32"),
    ("lshl_add_u64 (64-bit add)", r"^v_lshl_add_u64"),
    ("add/sub with carry (co)", r"^v_(add|sub|subrev)(b|c|brev)?_co_u32|^v_addc_co|^v_subb"),
    ("cndmask", r"^v_cndmask"),
    ("cmp", r"^v_cmp"),
    ("mov", r"^v_mov_b(32|64)|^v_accvgpr"),
    ("not/and/or/xor", r"^v_(not|and|or|xor)_b32"),
    ("shift/alignbit/bfe/perm", r"^v_(lshl|lshr|ashr)(rev)?_b(32|64)|^v_alignbit|^v_bfe|^v_perm|^v_lshl_or|^v_and_or"),
    ("add/sub u32 (no carry)", r"^v_(add|sub|subrev)_u32|^v_add3_u32|^v_lshl_add_u32|^v_add_lshl_u32"),
    ("other VALU", r"^v_"),
    ("ds_read", r"^ds_read"),
    ("ds_write", r"^ds_write"),
    ("global/buffer load", r"^(global|buffer|flat)_load"),
    ("global/buffer store", r"^(global|buffer|flat)_store"),
    ("s_waitcnt", r"^s_waitcnt"),
    ("s_nop", r"^s_nop"),
    ("s_barrier", r"^s_barrier"),
    ("s_load", r"^s_load|^s_buffer_load"),
    ("other SALU", r"^s_"),
]


def kernel_body(lines, needle):
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(\w+):", l)
        if m and needle in m.group(1) and not l.startswith(".L"):
            start = i
            name = m.group(1)
            break
    if start is None:
        raise SystemExit(f"kernel containing {needle!r} not found")
    body = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end") or t.startswith(".section") or t.startswith(".rodata"):
            break                                   # end of the function (blocks may follow the first s_endpgm)
        if t.startswith(";;#ASMSTART") or t.startswith(";;#ASMEND"):
            body.append(t[2:].split()[0])           # "#ASMSTART" / "#ASMEND" markers (kept for --split-asm)
            continue
        if not t or t.startswith(";") or t.startswith("//"):
            continue
        if t.startswith(".") and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        t = t.split(";")[0].strip()
        # llvm-objdump lines: "<instr> // addr: encoding"
        t = t.split("//")[0].strip()
        if t:
            body.append(t)
    return name, body


def innermost_loop(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or (span[1] - span[0]) > (best[1] - best[0]):
                best = span
    if best is None:
        raise SystemExit("no backward branch found")
    return body[best[0]:best[1] + 1]


def histogram(title, ins, per, dump_ops=True):
    hist = Counter()
    for l in ins:
        op = l.split()[0]
        for cname, pat in CLASSES:
            if re.match(pat, op):
                hist[cname] += 1
                break
        else:
            hist["(unclassified) " + op] += 1
    valu = sum(v for k, v in hist.items() if k in {c for c, p in CLASSES if p.startswith("^v_") or "v_" in p})
    print(f"{title}: {len(ins)} instructions, VALU {valu}")
    for cname, _ in CLASSES:
        if hist[cname]:
            extra = f"   {hist[cname] / per:7.2f} per unit" if per else ""
            print(f"  {cname:32s} {hist[cname]:6d}{extra}")
    for k, v in hist.items():
        if k.startswith("(unclassified)"):
            print(f"  {k:32s} {v:6d}")
    if per:
        print(f"  {'VALU total':32s} {valu:6d}   {valu / per:7.2f} per unit")
    if dump_ops:
        ops = Counter(l.split()[0] for l in ins)
        print("  opcodes " + ", ".join(f"{k} {v}" for k, v in ops.most_common()))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    per = None
    if "--per" in sys.argv:
        per = float(sys.argv[sys.argv.index("--per") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--per") + 1]]
    path, needle = args[0], args[1]
    lines = open(path).read().split("\n")
    name, body = kernel_body(lines, needle)
    if "--loop" in flags:
        body = innermost_loop(body)
    ins, inside, outside, in_asm = [], [], [], False
    for l in body:
        if l == "#ASMSTART":
            in_asm = True
            continue
        if l == "#ASMEND":
            in_asm = False
            continue
        if re.match(r"^\.LBB", l):
            continue
        ins.append(l)
        (inside if in_asm else outside).append(l)
    print(f"kernel  {name}")
    histogram(f"scope   {'innermost loop' if '--loop' in flags else 'whole kernel'}", ins, per)
    if "--split-asm" in flags:
        histogram("  -- inside inline asm (butterfly cores)", inside, per, False)
        histogram("  -- compiler-generated (addressing, exchanges, reductions, canonicalisation, control)", outside, per, False)
    body = [l for l in body if l not in ("#ASMSTART", "#ASMEND")]
    if "--dump" in flags:
        print("\n".join(body))


if __name__ == "__main__":
    main()
