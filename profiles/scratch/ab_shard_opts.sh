#!/bin/bash
# shard frame (an eighth of the garden frame) under run-time options
R=$PWD
for rep in 1 2; do
for opt in "" "PNR_NO_HOSTED_TAIL=1" "PNR_MARCH_BUDGET=4" "PNR_MARCH_BUDGET=8"; do
  echo "$opt: $(env $opt python3 $R/profiles/shard_profile.py 8 2>/dev/null | grep '^shards')"
done
done
