# usage: bash tools/ab_seq.sh "VAR=val" ...  -- tools/seq_rate.py (one so_icp_register_sequence call over 48 scans, last of 3 repetitions) per environment, three rounds, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for cfg in "$@"; do
  echo "$cfg: $(env $cfg python tools/seq_rate.py --count 48 --reps 3 2>&1 | grep '^\[sequence' | tail -1)"
done; done
