#!/bin/bash
# GPU box: tools/kernel_choice.py over launch lengths (1 spp) -- the data behind PT_POOL_MIN_TRIANGLES / PT_POOL_MIN_FRAMES.
for f in ${FRAME_LIST:-16 24 32 48 64 96}; do
  echo "== $f frames per launch"
  FRAMES=$f timeout -k 10 280 python tools/kernel_choice.py 4 8 16 27 48 76 2>/dev/null || exit 1
done
