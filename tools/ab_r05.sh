#!/bin/bash
# round 5 same-box A/B: the packed-fp32 right-hand sides (variants/r05/libgemx_pk.so) against the array code (libgemx_base.so)
V=variants/r05
python tools/ab_libs.py Cont-SC-SCIM-v0 default 65536,16384,131072,32768 $V/libgemx_base.so $V/libgemx_pk.so
python tools/ab_libs.py Cont-SC-PMSM-v0 default 16384,65536 $V/libgemx_base.so $V/libgemx_pk.so
python tools/ab_libs.py Finite-SC-SCIM-v0 default 16384,65536 $V/libgemx_base.so $V/libgemx_pk.so
