#!/bin/bash
# Copies one measurement pass (gpurun_out/<tag>/ as left by tools/r03_round.sh) into profiles/ under per-round names.
# usage: bash tools/collect_profiles.sh r03a
set -eu
cd "$(dirname "$0")/.."
TAG=$1
SRC=gpurun_out/$TAG
for cfg in sintel kitti_alt things_train 1080p; do
  [ -f "$SRC/bench_$cfg.json" ] && cp "$SRC/bench_$cfg.json" "profiles/${TAG}_bench_$cfg.json"
  f=$(ls "$SRC"/prof_$cfg/*/*_kernel_stats.csv 2>/dev/null | head -1 || true)
  [ -n "$f" ] && cp "$f" "profiles/${TAG}_${cfg}_kernel_stats.csv"
done
[ -f "$SRC/parity_errors.json" ] && cp "$SRC/parity_errors.json" "profiles/${TAG}_parity_errors.json"
ls profiles | grep "^$TAG"
