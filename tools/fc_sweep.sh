#!/bin/bash
# Tile / split-K sweep of the FC-layer GEMMs (tools/opbench.py with the ACKTR_TILE / *_SLICES overrides of linear.hip)
for t in 2 3 4; do
  for f in 0 2 3 4 6 8 12; do
    echo "== fwd tile $t slices $f"; ACKTR_TILE=$t ACKTR_FWD_SLICES=$f timeout -k 5 60 python tools/opbench.py --ops fc_fwd --iters 100 | grep fc_ || exit 1
  done
  echo "== dgrad tile $t"; ACKTR_TILE=$t timeout -k 5 60 python tools/opbench.py --ops fc_dgrad --iters 100 | grep fc_ || exit 1
  for f in 0 1 2 4 6 8; do
    echo "== wgrad tile $t slices $f"; ACKTR_TILE=$t ACKTR_WGRAD_SLICES=$f timeout -k 5 60 python tools/opbench.py --ops fc_wgrad --iters 100 | grep fc_ || exit 1
  done
done
