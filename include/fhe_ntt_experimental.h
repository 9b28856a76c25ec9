/* fhe_ntt_experimental.h — entry points that are NOT part of the drop-in boundary (include/fhe_ntt.h).
 *
 * The one-launch n = 2^16 forward transform (fhe-study_amd/csrc/ntt_persist.hip, round 4): persistent workgroups that run
 * the strided and the contiguous stages of NTT::ntt (arith/src/ntt.rs:44-73) in ONE kernel.  Bit-exact in every test
 * (tests/test_round4_gpu.py), off by default, and 12-45 % SLOWER than the two-pass kernels on every setting measured
 * (profiles/r04_persist_*.txt; the ceiling of any such kernel: profiles/r05_single_pass_bound.txt).
 * Kept as the measured record of that design and as a test bed; nothing in the reference binds to it, a caller of
 * arith::NTT never needs it, and these declarations may change or disappear.
 *
 * Hand-off inside the kernels (why it is sound, and what it is NOT measured for): a workgroup draws tickets from the
 * queue of its OWN XCD only, so producer and consumer of every intermediate share one L2.  The producer's waves store
 * plainly, each waits `s_waitcnt vmcnt(0)`, the workgroup meets at a barrier and ONE lane adds to an agent-scope counter;
 * the consumer polls that counter with an `sc1` load, meets at a barrier and reads the intermediate with 8-byte `sc1`
 * loads, which bypass the CU's L1 and are served by that same L2.  /opt/skills/guides/MI355X_MICROARCH.md's table of
 * measured hand-offs has no row for plain stores + 8-byte sc1 loads at 2-4 workgroups per CU (its plain-store row lists
 * dword / dwordx4 loads at one workgroup per CU): the evidence here is the comparison of every word in 14 settings x 5
 * batches x 7 grid sizes, and tests/test_round5.py::test_persistent_under_uneven_load (a copy kernel streaming on a
 * second stream, consumers L1-warm).  That is a test record, not an architectural guarantee — one more reason these
 * kernels stay outside the boundary. */
#ifndef FHE_NTT_EXPERIMENTAL_H
#define FHE_NTT_EXPERIMENTAL_H

#include "fhe_ntt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Persistent workgroups draw tickets from one queue per XCD; every wait inside the kernel is bounded.
 *   mode 0  off: the two-pass kernels.
 *   mode 1  "A": tiles of tile_polys polynomials (power of two <= 1024); the strided stages run `lag` tiles ahead of
 *           the contiguous ones; the intermediate lives in a ring of `ringslots` (>= lag + 1) tile slots per XCD, or in
 *           the output buffer (ringslots = 0).
 *   mode 2  "B": teams — sixteen workgroups of one XCD take ONE polynomial through both halves and meet in between; the
 *           intermediate lives in a ring of `ringslots` (>= 1) polynomial slots per XCD and is read back out of the L2.
 *   mode 3  "D": mode 2 at two workgroups per CU — 256 registers per lane, both twiddle tiles in LDS, the next part's
 *           coefficients prefetched into a second register set; `lag` = start-up stagger between groups, as for mode 2.
 *   mode 4  "E": the teams without the meeting, two workgroups per CU: a workgroup keeps a FIFO of the parts whose
 *           contiguous half is still to come and runs ONE half per iteration — the contiguous half of its oldest part once
 *           that polynomial's sixteen strided halves are in, a strided half of the next ticket otherwise — with the
 *           coefficients of the next two halves in flight; it waits (bounded) only when it can do neither.  ringslots >= 2.
 * Environment: FHE_NTT_PERSIST=A:T,L,R, B:R[,s], D:R[,s] or E:R[,s].  fhe_ntt_persist_status() (after synchronising) returns FHE_E_HIP if a
 * bounded wait ran out, and clears the flag. */
int fhe_ntt_set_persist(unsigned mode, unsigned tile_polys, unsigned lag, unsigned ringslots);
int fhe_ntt_persist_status(void);
/* workgroups a persistent launch uses (default 0: as many as the chip holds).  Any number >= 1 gives the same words:
 * nothing in the kernels assumes that workgroups are resident together (tests run 1, 8, 20, ...). */
int fhe_ntt_set_persist_grid(int workgroups);
/* diagnostic: d_words26 = device buffer of 26 uint64_t (zeroed by the caller), or NULL to stop; see tools/persist_bench.py */
int fhe_ntt_persist_profile(void *d_words26);

#ifdef __cplusplus
}
#endif

#endif /* FHE_NTT_EXPERIMENTAL_H */
