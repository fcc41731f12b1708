cd $GRAFT_REPO_ROOT
for m in base 1 2 4 6 8 16 32; do
  if [ $m = base ]; then unset ANIP_LIB; else export ANIP_LIB=aniportrait_amd/lib/libanip_abl$m.so; fi
  for z in 0 1; do
    if [ $z = 1 ]; then export ATTN_ZERO=1; else unset ATTN_ZERO; fi
    echo -n "ablate=$m zero=$z  "; timeout 120 python tools/bench_kernels.py --only=attn 2>/dev/null | grep '"64^2 d40' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.1f us  %.0f TF'%(r['us'],r['tflops']))"
  done
done
