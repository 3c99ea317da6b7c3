V=variants/r05
for P in 0 5600 6000 6400 6800 7200 7600; do
  echo "## GEMX_PACE_GBPS=$P"
  GEMX_PACE_GBPS=$P python tools/ab_libs.py Cont-SC-SCIM-v0 default 65536 $V/libgemx_base.so $V/libgemx_pk.so | grep -v "^| library\|^|---\|product"
done
for S in 1 2; do
  echo "## GEMX_PIPE_SHAPE=$S (default pace)"
  GEMX_PIPE_SHAPE=$S python tools/ab_libs.py Cont-SC-SCIM-v0 default 65536 $V/libgemx_base.so $V/libgemx_pk.so | grep -v "^| library\|^|---\|product"
done
