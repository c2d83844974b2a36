#!/bin/bash
# On-the-fly backward: tests, lab timings and a kernel trace of the lab.  usage: bash tools/gpu_alt.sh <outdir-name> [notest]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-alt}
mkdir -p "$OUT"
cd "$ROOT"
if [ "${2:-}" != "notest" ]; then
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "alternate or on_the_fly or k1_kitti or t1_things" > "$OUT/pytest.log" 2>&1 || { echo "pytest failed"; tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
fi
timeout -k 10 120 python tools/alt_bwd_lab.py || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof"
WINDOW_ONLY=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$ROOT/tools/alt_bwd_lab.py" > "$OUT/prof.log" 2>&1 || { echo "prof failed"; tail -20 "$OUT/prof.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:12]:
        import re
        name = re.sub(r"raftcorr::|\(anonymous namespace\)::|void ", "", r["Name"])
        print("%-50s calls %5s avg %9.1f us" % (name[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
