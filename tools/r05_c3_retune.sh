#!/bin/bash
# C3 (inner-product prover, n = 2^20) per-round times under the round's new window table, and the fold thresholds re-tuned
for cfg in "" "ipa_small_m=2048" "ipa_small_m=8192" "ipa_small_m=16384" "ipa_big_m=131072" "ipa_big_m=524288" "ipa_small_m=1"; do
  echo "#### $cfg"
  timeout 300 python tools/c3_round_times.py 20 $cfg 2>&1 | tail -23
done
