#!/bin/bash
# tools/ab_bfv_r04.sh — same-box A/B builds of bfv32.hip's shape switches (FHE_B32_INV_T / FHE_B32_INV_R: how the block kernels run
# their inverse rounds — 0 twiddles preloaded ahead of the exchanges, 1 read as they go, 2 one stage at a time; FHE_B32_FWD_PRELOAD):
# fhe-study_amd/build/abl/libfhe_ntt_bfv_<tag>.so, loaded with FHE_NTT_LIB=...  (round 3 chose T = 2, R = 1 while the kernels spilled)
set -e
cd "$(dirname "$0")/.."
B=fhe-study_amd/build; mkdir -p $B/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=off"
OBJS="$B/capi.o $B/ntt_kernels.o $B/ntt_kernels_q62.o $B/ntt_persist.o $B/digit_mac.o $B/digit32.o $B/smallq.o $B/generic63.o $B/zring.o $B/glue.o"
build() {   # tag, defines
  /opt/rocm/bin/hipcc $F $2 -c -o $B/abl/bfv32_$1.o fhe-study_amd/csrc/bfv32.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $B/abl/libfhe_ntt_bfv_$1.so $B/abl/bfv32_$1.o $OBJS
}
for t in 0 1 2; do for r in 0 1 2; do build t${t}r${r} "-DFHE_B32_INV_T=$t -DFHE_B32_INV_R=$r" & done; done
build fwd0 "-DFHE_B32_FWD_PRELOAD=0" &
wait
ls $B/abl/libfhe_ntt_bfv_*.so
