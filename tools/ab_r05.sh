#!/bin/bash
# round 5 same-box A/B of the config-4 levers (tools/dev_build.py variants under variants/r05): product library first, then each variant
V=variants/r05
python tools/ab_libs.py Cont-SC-SCIM-v0 default 65536,16384,131072 $V/libgemx_base.so $V/libgemx_pl.so $V/libgemx_pre.so $V/libgemx_both.so $V/libgemx_nopart.so
python tools/ab_libs.py Cont-SC-SCIM-v0 default 65536 $V/libgemx_base.so $V/libgemx_both.so
python tools/ab_libs.py Cont-SC-PMSM-v0 default 16384,65536 $V/libgemx_base.so $V/libgemx_pl.so $V/libgemx_pre.so $V/libgemx_both.so
python tools/ab_libs.py Cont-CC-PMSM-v0 rk4 16384,65536,131072 $V/libgemx_base.so $V/libgemx_pl.so $V/libgemx_pre.so $V/libgemx_both.so
python tools/ab_libs.py Finite-CC-PMSM-v0 rk4 16384,32768 $V/libgemx_base.so $V/libgemx_nopart.so
python tools/ab_libs.py Finite-CC-PMSM-v0 rk4 16384 $V/libgemx_base.so $V/libgemx_nopart.so
