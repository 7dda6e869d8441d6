#!/bin/bash
# pair-kernel LDS padding (caps its workgroups per CU so that the reciprocal-space chain can run beside it) and the
# multi-wave pruning pass at the reference's default operating point: bash tools/lds_pad_ab.sh [N_SIDE]
export DP_ONLY=spme
ns=${1:-64}
for pad in 0 12000 26000 40000; do
  echo "MDX_NB_LDS_PAD=$pad: $(MDX_NB_LDS_PAD=$pad python3 tools/default_point_time.py $ns 2>/dev/null | grep OPC)"
done
echo "MDX_PRUNE_MW_BELOW=30000: $(MDX_PRUNE_MW_BELOW=30000 python3 tools/default_point_time.py $ns 2>/dev/null | grep OPC)"
echo "MDX_CONS_SORT=0: $(MDX_CONS_SORT=0 python3 tools/default_point_time.py $ns 2>/dev/null | grep OPC)"
