#!/bin/bash
# tools/ab_r04.sh — same-box A/B builds of round 4's switches in ntt_kernels.hip (csrc/ntt_kernels.hip: FHE_MID_ONE_TILE,
# FHE_MID_EARLY_FETCH, FHE_INV_TLOAD): fhe-study_amd/build/abl/libfhe_ntt_<tag>.so, loaded with FHE_NTT_LIB=...
# (the switches act on the headline arithmetics of ntt_kernels.hip; the second unit, ntt_kernels_q62.o, is the default build's)
set -e
cd "$(dirname "$0")/.."
B=fhe-study_amd/build; mkdir -p $B/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=off"
OBJS="$B/capi.o $B/ntt_persist.o $B/digit_mac.o $B/digit32.o $B/bfv32.o $B/smallq.o $B/generic63.o $B/ntt_kernels_q62.o $B/zring.o $B/glue.o"
build() {   # tag, defines
  /opt/rocm/bin/hipcc $F $2 -c -o $B/abl/ntt_kernels_$1.o fhe-study_amd/csrc/ntt_kernels.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $B/abl/libfhe_ntt_$1.so $B/abl/ntt_kernels_$1.o $OBJS
}
build base "-DFHE_MID_ONE_TILE=0 -DFHE_MID_EARLY_FETCH=0 -DFHE_INV_TLOAD=0" &
build tile "-DFHE_MID_ONE_TILE=1 -DFHE_MID_EARLY_FETCH=0 -DFHE_INV_TLOAD=0" &
build fetch "-DFHE_MID_ONE_TILE=0 -DFHE_MID_EARLY_FETCH=1 -DFHE_INV_TLOAD=0" &
build tload "-DFHE_MID_ONE_TILE=0 -DFHE_MID_EARLY_FETCH=0 -DFHE_INV_TLOAD=1" &
wait
ls -la $B/abl/libfhe_ntt_{base,tile,fetch,tload}.so
