for rep in 1 2; do
for L in variants/libgemx_noprep.so variants/libgemx_prep.so; do
  echo "== $L"
  GEMX_LIBRARY=$PWD/$L python tools/ab_full_variant.py rc rinit 2>&1 | grep "GEMX_PIPE=1\|GEMX_PIPE=2" | cut -c1-80
done
done
