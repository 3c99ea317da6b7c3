#!/bin/bash
# round 5 same-box A/B: the shallow shapes' loader two blocks ahead (product) against one block ahead (variants/r05/libgemx_ahead1.so)
V=variants/r05
python tools/ab_libs.py Cont-CC-PMSM-v0 rk4 32768,65536,131072 $V/libgemx_ahead1.so
python tools/ab_libs.py Cont-SC-SCIM-v0 default 32768,65536,131072 $V/libgemx_ahead1.so
python tools/ab_libs.py Cont-SC-PMSM-v0 default 65536 $V/libgemx_ahead1.so
python tools/ab_libs.py Cont-CC-PermExDc-v0 rk4 65536,131072 $V/libgemx_ahead1.so
python tools/ab_libs.py Finite-CC-EESM-v0 rk4 65536,131072 $V/libgemx_ahead1.so
python tools/ab_libs.py Finite-CC-PMSM-v0 rk4 32768,65536,131072 $V/libgemx_ahead1.so
