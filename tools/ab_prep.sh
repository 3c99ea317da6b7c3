# same-box A/B of variant libraries on the random-initialiser rows: bash tools/ab_prep.sh variants/a.so variants/b.so ...
for rep in 1 2; do
for L in "$@"; do
  echo "== $L"
  GEMX_LIBRARY=$PWD/$L python tools/ab_full_variant.py rinit rinit_scim 2>&1 | grep "GEMX_PIPE=1" | cut -c1-80
done
done
