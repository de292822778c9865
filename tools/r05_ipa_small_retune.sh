#!/bin/bash
# the inner-product prover at n = 2^13 (config C4's argument) and 2^12 / 2^14 / 2^16: where to fold the bases once through products (ipa_small_m)
for logn in 13 12 14 16; do
  for cfg in "" "ipa_small_m=4096" "ipa_small_m=2048" "ipa_small_m=1024" "ipa_small_m=512" "ipa_small_m=256" "ipa_small_m=128"; do
    echo "logn=$logn [$cfg] $(timeout 120 python tools/c3_round_times.py $logn $cfg 2>&1 | tail -1)"
  done
done
