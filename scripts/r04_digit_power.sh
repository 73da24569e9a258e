#!/bin/bash
# How much of the bits GEMM's time is the DATA: the same kernel, same X, digit planes overwritten with chosen distributions
for d in "" full pos7 neg7 pos6 pos4 one m1 zero "" full pos7; do
  echo "== DIGITS='$d'"
  DIGITS=$d timeout -k 10 120 python scripts/gemm_i8_microbench.py 40 3 2>&1 | grep -v "^\[stamps\|^   " | tail -3
done
