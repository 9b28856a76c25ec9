#!/usr/bin/env python3
"""tools/persist_report.py <evidence dir> <tag> — profiles/<tag>_persist_A.txt and _B.txt from what
tools/refresh_evidence.sh step [8] measured: ms per step (65536 polynomials) at three settings of each variant beside the
two-pass kernels on the same box, where a workgroup's time goes (lane 0's shader-clock ticks per part of an iteration),
HBM-side traffic and instruction counters per launch."""
import os, re, sys

out, tag = sys.argv[1], sys.argv[2]
lines = open(os.path.join(out, "persist_bench.txt")).read().splitlines()
cnt = open(os.path.join(out, "persist_counters.txt")).read() if os.path.exists(os.path.join(out, "persist_counters.txt")) else ""
blocks, cur = [], None
for l in lines:
    if l.startswith("time "):
        cur = [l]
        blocks.append(cur)
    elif cur is not None and re.match(r"\s+[SC]:", l):
        cur.append(l)
two_pass = [b for b in blocks if "two-pass" in b[0]]
head = {
    "A": "# Variant A of the one-launch n = 2^16 forward transform (csrc/ntt_persist.hip: ntt_fwd_persist_kernel): persistent workgroups,\n"
         "# one ticket queue per XCD, tiles of T polynomials, the strided stages running L tiles ahead of the contiguous ones, the\n"
         "# intermediate in the output buffer (R = 0) — i.e. through the Infinity Cache for tiles that fit it.  Settings: A:T,L,R.\n",
    "B": "# Variant B (ntt_fwd_team_kernel): teams of sixteen workgroups of one XCD take ONE polynomial through both halves; the\n"
         "# intermediate lives in a ring of R polynomial slots per group of sixteen and is read back a few microseconds after it\n"
         "# was written.  Settings: B:R,s (s = start-up stagger between the groups of an XCD): four workgroups per CU; D:R,s: the\n"
         "# same at TWO workgroups per CU (256 registers, resident twiddle tiles, the next part prefetched into a second register set);\n"
         "# E:R,s (ntt_fwd_flow_kernel): the teams without the meeting — a FIFO of pending parts per workgroup, ONE half per iteration, the\n"
         "# next two halves' loads in flight.  E rows: 'poll' = deciding which half runs (both S and C iterations, charged to S);\n"
         "# S 'store-barrier' column = how often nothing was decided ahead, S 'next-loads' = ... because the ring slot was still being\n"
         "# read, C 'poll' = ... because no strided half was fetched, C 'hand-over' = ... and a C half followed (counts per part).\n",
}
for v in ("A", "B"):
    with open(os.path.join("profiles", f"{tag}_persist_{v}.txt"), "w") as f:
        f.write(head[v])
        f.write("# tools/persist_bench.py 65536 ... (one MI355X; 65536 polynomials = one bench step; parity of every setting against the\n"
                "# two-pass kernels in the same run: " + next((l for l in lines if l.startswith("parity:")), "parity: ?") + ")\n")
        f.write("# ticks = shader clock of lane 0 of every workgroup, per work item (S = strided stages of 4096 coefficients, C = contiguous):\n"
                "#   top-barrier / xchg-barrier / store-barrier = waiting for the workgroup's other waves; poll = resolving a ticket by polling\n"
                "#   (A) ; half1 = first four stages incl. the wait for the coefficients; look-ahead = (A) next ticket's control words, (B, S\n"
                "#   rows) THE TEAM WAIT; gather / round1 / epi-a / next-loads / half2-rest = second half: LDS gather, last four stages,\n"
                "#   stores or canonical scatter, issuing the next item's loads, store loop; hand-over = ticket bookkeeping.\n\n")
        for b in two_pass + [b for b in blocks if f" {v}:" in b[0] or (v == "B" and (" D:" in b[0] or " E:" in b[0]))]:
            f.write("\n".join(b) + "\n")
        f.write("\n# ---- HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled on gfx950) and\n"
                "# ---- instruction counters (SQ_*, GRBM_GUI_ACTIVE) per launch: tools/persist_one.py <setting> 8192 3 ----\n")
        keep, on = [], False
        for l in cnt.splitlines():
            if l.startswith("== "):
                name = l[3:].split(" ")[0].rstrip(":")          # "two-pass", "A:64,1,0", "B:2,1", "D:1,1"
                on = name == "two-pass" or name[0] == v or (v == "B" and name[0] in "DE")
            if on:
                keep.append(l[:400])
        f.write("\n".join(keep) + "\n")
print("wrote profiles/%s_persist_A.txt, _B.txt" % tag)
