#!/bin/bash
# Things config: the backward tests, the bench line and the kernel trace of the same command.  usage: bash tools/gpu_things.sh <outdir-name>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-things}
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "deferred or t1_things or golden or bit_reproducible or accumulates or fused_with_convc1_backward or record or other_radii or partial_backward or 16bit" > "$OUT/pytest.log" 2>&1 || { echo "pytest failed"; tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
timeout -k 10 200 python bench.py --config things_train --no-cpu-baseline > "$OUT/things.json" 2> "$OUT/things.err" || { echo "bench failed"; tail -20 "$OUT/things.err"; exit 1; }
python - "$OUT/things.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("things_train: %.0f pairs/s, %.3f ms/step, fwd %.0f us, bwd %.0f us" % (d["value"], d["ms_per_step"], d["us_fwd"], d["us_bwd"]))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$ROOT/bench.py" --config things_train --no-extras --no-cpu-baseline > "$OUT/prof.log" 2>&1 || { echo "prof failed"; tail -20 "$OUT/prof.log"; exit 1; }
cut -d, -f1-4 "$OUT"/prof/*/*kernel_stats.csv | cut -c1-150 | head -14
