#!/bin/bash
# Round-3 measurement pass: GPU tests, the four bench configs, the kernel trace of each bench command.
# Every GPU step runs under its own time limit; a failing step ends the script.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r03$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest.log" 2>&1 || { echo "pytest failed"; tail -30 "$OUT/pytest.log"; exit 1; }
tail -3 "$OUT/pytest.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
fi
for cfg in ${CONFIGS:-sintel kitti_alt things_train 1080p}; do
  timeout -k 10 300 python bench.py --config $cfg > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err" || { echo "bench $cfg failed"; tail -20 "$OUT/bench_$cfg.err"; exit 1; }
  echo "bench $cfg ok"
done
cd /tmp && export TMPDIR=/tmp
for cfg in ${CONFIGS:-sintel kitti_alt things_train 1080p}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$cfg" -- python3 "$ROOT/bench.py" --config $cfg --no-extras --no-cpu-baseline > "$OUT/prof_$cfg.log" 2>&1 || { echo "prof $cfg failed"; tail -20 "$OUT/prof_$cfg.log"; exit 1; }
  echo "prof $cfg ok"
done
