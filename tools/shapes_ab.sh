#!/bin/bash
# A/B of library variants (tools/build_variants.sh) on launch shapes around the planner's choices:
# tools/shapes_ab.sh "name1 name2" [reps]
names=${1:?variant names}; reps=${2:-2}
for cfg in "--batch 128" "--seqlen 16384" "--batch 128 --seqlen 16384" "--batch 32" "--seqlen 8192 --batch 256" "--batch 100"; do
  echo "### cfg: $cfg"
  tools/ab.sh "$names" $reps $cfg || exit 1
done
