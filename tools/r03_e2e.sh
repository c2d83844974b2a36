#!/bin/bash
# Round-3 end-to-end refresh: RAFT-basic inference with each correlation block, and the training step (1 GPU).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03e2e
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python tools/raft_e2e.py 32 > "$OUT/raft_e2e.json" 2> "$OUT/raft_e2e.err" || { echo "raft_e2e failed"; tail -5 "$OUT/raft_e2e.err"; exit 1; }
echo "raft_e2e ok"
for c in ours stock alt; do
  timeout -k 10 300 python tools/train_step.py --corr $c --steps 10 > "$OUT/train_$c.json" 2> "$OUT/train_$c.err" || { echo "train_step $c failed"; tail -5 "$OUT/train_$c.err"; exit 1; }
  echo "train_step $c ok"
done
