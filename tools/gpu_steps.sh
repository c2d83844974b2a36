#!/bin/bash
# Runs GPU steps one after another; an assertion failure (exit 1) lets the next step run, anything
# else (fault, abort, time limit) ends the script there.  usage: gpu_steps.sh "<secs> <cmd>" ...
for step in "$@"; do
  secs=${step%% *}; cmd=${step#* }
  echo "=== [$secs s] $cmd"
  timeout -k 10 "$secs" bash -c "$cmd"
  rc=$?
  echo "=== exit $rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
done
exit 0
