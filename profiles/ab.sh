#!/bin/bash
# A/B of library variants in ONE process-sequence on ONE box (device-to-device spread is several %).
# usage: bash profiles/ab.sh "<bench args>" lib1.so lib2.so ...   (two interleaved rounds)
ARGS=$1; shift
for round in 1 2; do
  for L in "$@"; do
    CRT_AMD_LIB=$PWD/$L timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
