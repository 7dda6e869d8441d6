# arrangements of the decomposed step for one rank of 8 (tools/one_rank_profile.py), with and without the fold of pack / add into the drift pass
export ONE_RANK_TRACE=1 ONE_RANK_WIRE="0,25"
for fold in 1 0; do
  echo "== plain, fold $fold"; MDX_HALO_FOLD=$fold ONE_RANK_SPLIT=0 timeout 200 python tools/one_rank_profile.py 8 96 2>&1 | grep "^world"
  echo "== split, fold $fold"; MDX_HALO_FOLD=$fold ONE_RANK_SPLIT=1 timeout 200 python tools/one_rank_profile.py 8 96 2>&1 | grep "^world"
done
echo "== single launch of drawn tiles (MDX_HALO_PIPE=1)"; MDX_HALO_PIPE=1 ONE_RANK_SPLIT=pipe timeout 200 python tools/one_rank_profile.py 8 96 2>&1 | grep "^world"
ONE_RANK_WIRE=25 timeout 200 bash tools/kt_step.sh ktsplit 8 1 2>&1 | cut -c1-160 | head -40
