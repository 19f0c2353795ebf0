#!/bin/bash
# developer probe: compares alternative builds of the library (paths relative to the repo root)
for lib in "$@"; do
  echo "== $lib"
  LSD_HIP_LIB=$PWD/$lib timeout 120 python tools/scale_probe.py 2048 1 512 2>&1 | grep -E "^(1|512) " | cut -c1-200
done
