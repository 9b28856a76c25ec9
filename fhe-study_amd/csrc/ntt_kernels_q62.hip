// ntt_kernels_q62.hip — the second translation unit of ntt_kernels.hip: the same kernel templates (NTT::ntt / intt,
// arith/src/ntt.rs:44-110; Rq x Rq, ring_nq.rs:586-607) instantiated for the arithmetics of the moduli at the top of the
// reference's range — AR = 0 (2^61 <= q < 2^62: Harvey's [0, 4q)) and AR = 3 (2^62 <= q < 2^63: strict, every value
// canonical) — behind the five q62_* entry points declared in ntt_kernels.hip.  Split off so that the two units compile
// in parallel (one unit for every arithmetic was the library's longest compile by far).
#define FHE_NTT_TU_Q62
#include "ntt_kernels.hip"
