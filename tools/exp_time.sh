#!/bin/bash
# GPU box: times library variants with tools/smx_time on a list of shapes.
#   tools/exp_time.sh "<lib1> <lib2> ..." "<R K sigma>;<R K sigma>;..."
set -u
IFS=';' read -ra SHAPES <<< "$2"
for lib in $1; do
  for sh in "${SHAPES[@]}"; do
    echo "== $lib  $sh"
    timeout -k 5 60 tools/smx_time $lib $sh 20 | grep -v "^x " || exit 1
  done
done
