#!/bin/bash
# tools/abl_build.sh — timing-only builds of the fused digit kernels (never loaded by the product or the tests):
#   fhe-study_amd/build/abl/libfhe_ntt_nomac.so   digit_mac32_kernel without its multiply phase   (-DFHE_D32_ABLATE_MAC)
#   fhe-study_amd/build/abl/libfhe_ntt_nontt.so   ... without its transforms                      (-DFHE_D32_ABLATE_NTT)
#   fhe-study_amd/build/abl/libfhe_ntt_b32noinv.so / _b32noepi.so   bfv32.hip's inverse kernels without their transforms / f64 epilogues
# Use: FHE_NTT_LIB=fhe-study_amd/build/abl/libfhe_ntt_nomac.so python tools/abl_key_switch.py   (results are wrong by design)
set -e
cd "$(dirname "$0")/.."
B=fhe-study_amd/build; mkdir -p $B/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=off"
for v in MAC NTT; do
  lc=$(echo $v | tr A-Z a-z)
  /opt/rocm/bin/hipcc $F -DFHE_D32_ABLATE_$v -DFHE_DM_ABLATE_$v -c -o $B/abl/digit32_no$lc.o fhe-study_amd/csrc/digit32.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $B/abl/libfhe_ntt_no$lc.so $B/capi.o $B/ntt_kernels.o $B/ntt_kernels_q62.o $B/ntt_persist.o $B/generic63.o $B/digit_mac.o $B/abl/digit32_no$lc.o $B/bfv32.o $B/smallq.o $B/zring.o $B/glue.o
done
for v in INV EPI; do
  lc=$(echo $v | tr A-Z a-z)
  /opt/rocm/bin/hipcc $F -DFHE_B32_ABLATE_$v -c -o $B/abl/bfv32_no$lc.o fhe-study_amd/csrc/bfv32.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $B/abl/libfhe_ntt_b32no$lc.so $B/capi.o $B/ntt_kernels.o $B/ntt_kernels_q62.o $B/ntt_persist.o $B/generic63.o $B/digit_mac.o $B/digit32.o $B/abl/bfv32_no$lc.o $B/smallq.o $B/zring.o $B/glue.o
done
ls -la $B/abl/*.so
