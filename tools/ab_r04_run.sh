for rep in 1 2; do
for v in base tile fetch tload all; do
  if [ $v = all ]; then L=fhe-study_amd/libfhe_ntt.so; else L=fhe-study_amd/build/abl/libfhe_ntt_$v.so; fi
  echo "== $v rep $rep"
  FHE_NTT_LIB=$L python tools/kbench.py 16 16384 2>/dev/null | grep inv
  FHE_NTT_LIB=$L python tools/mulbench.py 16:4096 14:16384 2>/dev/null | grep rq_mul
done; done
