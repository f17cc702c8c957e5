#!/bin/bash
for cfg in "--batch 128" "--seqlen 16384" "--batch 128 --seqlen 16384" "--batch 32" "--seqlen 8192 --batch 256" "--batch 100"; do
  echo "### cfg: $cfg"
  tools/ab.sh "base d2 base d2" 1 $cfg || exit 1
done
