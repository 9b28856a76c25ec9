set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a/traffic; mkdir -p $O
for cfg in two-pass B:2,1 A:16,1,0; do
  tag=$(echo $cfg | tr ':,' '__')
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f_$tag -o run --output-format csv -- python3 $R/tools/persist_one.py $cfg 8192 3 > $O/f_$tag.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w_$tag -o run --output-format csv -- python3 $R/tools/persist_one.py $cfg 8192 3 > $O/w_$tag.log 2>&1
  echo "== $cfg (8192 polynomials per launch: 4295 MB in, 4295 MB out)"
  python3 $R/tools/pmc_kernels.py $O/f_$tag $O/w_$tag 100
done
