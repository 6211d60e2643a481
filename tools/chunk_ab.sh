#!/bin/bash
# Run ON THE GPU BOX: A/B of TOYNI_CHUNK_ELEMS at n = 2^20 / 2^21 (profiles/r05_ab_chunk.txt)
for ch in 0 8388608 16777216 33554432 67108864; do
  echo "== TOYNI_CHUNK_ELEMS=$ch"
  TOYNI_CHUNK_ELEMS=$ch SWEEP_RANGE=20:22 timeout -k 10 120 python3 tools/sweep.py 2>&1 | grep "n=2"
done
