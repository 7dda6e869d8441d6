#!/bin/bash
# clusters by interaction kind at the reference's default operating point: off / list only / list + kind-aware pair kernel
# bash tools/kind_ab.sh [N_SIDE...]
export DP_ONLY=spme
for ns in "${@:-64}"; do
  echo "n_side $ns off:           $(MDX_KIND_CLUSTERS=0 python3 tools/default_point_time.py $ns 2>/dev/null | grep OPC)"

  echo "n_side $ns on:  $(MDX_KIND_CLUSTERS=1 python3 tools/default_point_time.py $ns 2>/dev/null | grep OPC)"
done
