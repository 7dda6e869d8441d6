#!/bin/bash
# brick spread against tile spread at the reference's default operating point, several sizes: bash tools/spread_ab.sh
export DP_ONLY=spme
for ns in 18 32 40 50; do
  for arm in 0 1; do
    echo "n_side $ns MDX_PME_SPREAD_BRICK=$arm: $(MDX_PME_SPREAD_BRICK=$arm python3 tools/default_point_time.py $ns 2>/dev/null | grep OPC)"
  done
done
