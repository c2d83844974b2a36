#!/bin/bash
# Round-3 PMC passes (separate runs per counter group, kernel trace off) for the four bench shapes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for cfg in ${CONFIGS:-sintel kitti_alt things_train 1080p}; do
  bash "$ROOT/tools/pmc_passes.sh" 2 $cfg r03pmc_$cfg || { echo "pmc $cfg failed"; exit 1; }
done
