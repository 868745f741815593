#!/bin/bash
# parity subset under kernel build variants: tools/ab_parity.sh <pytest -k expr> <variant>...
K=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset LRHIP_LIB; else export LRHIP_LIB=$PWD/luisarender_amd/lib/variants/liblrhip_$v.so; fi
  echo "== $v"; timeout 600 python -m pytest tests -m gpu -q -k "$K" 2>&1 | tail -4
done
