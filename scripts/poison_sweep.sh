#!/bin/bash
# Which working buffer does a test read before anything writes it?  One run of the test per buffer, that buffer poisoned.
#   scripts/poison_sweep.sh <pytest node id> [more pytest arguments]
out=gpurun_out/poison
mkdir -p $out
: > $out/sweep.txt
for name in none active basis cand cand_la cand_next cmin colp colv crow is_basic list_full list_part obj part_i part_v pass pr_stamp prb_c prb_cache prb_elist prb_ep_i prb_ep_v prb_erow prb_escore prb_i prb_info prb_top_l prb_top_u prb_ub prb_ucand prb_v ptr_full rec_price rec_ratio rec_shard rec_tail rhs_next rowp rsl_full rsl_part st steps  all; do
  LPAMD_POISON=$name timeout -k 10 300 python -u -m pytest "$@" -x -q -m gpu --timeout 120 --timeout-method thread > $out/$name.log 2>&1
  echo "$name rc $? $(tail -1 $out/$name.log)" | tee -a $out/sweep.txt
done
