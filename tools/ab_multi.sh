#!/bin/bash
# A/B of library variants over the bench shapes: tools/ab_multi.sh "name1 name2" reps
names=$1; reps=${2:-3}
for cfg in "" "--kv-heads 8" "--kv int8 --seqlen 65536" "--kv-heads 8 --kv int8 --seqlen 65536" "--kv-heads 8 --seqlen 131072 --topk 64"; do
  echo "### cfg: $cfg"
  tools/ab.sh "$names" $reps $cfg || exit 1
done
