#!/bin/bash
# Runs the steps of one gpurun call: each line of the file given as $1 is "<timeout seconds> <output file> <command...>".
# A step that fails (a test assertion, a non-zero exit) does not stop the following ones; a step that is KILLED at its
# time limit or dies from a signal does: nothing further touches the GPU after a hang or a fault.
set -u
while IFS= read -r line; do
  [ -z "$line" ] && continue
  case "$line" in \#*) continue;; esac
  t=${line%% *}; rest=${line#* }; out=${rest%% *}; cmd=${rest#* }
  mkdir -p "$(dirname "$out")"
  echo "== [$t s] $cmd" | tee -a "$out.meta"
  timeout -k 10 "$t" bash -c "$cmd" > "$out" 2>&1
  rc=$?
  echo "rc=$rc" | tee -a "$out.meta"
  tail -n 3 "$out"
  if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping here"; exit $rc; fi
done < "$1"
