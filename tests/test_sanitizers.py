"""Sanitizer runs of the code that runs on the host (the reference has none; GPU sanitizers are not
available on this pool, so device code is covered by the parity tests and the fuzz run instead).

* the C oracle (oracle/alt_corr_oracle.c) under gcc's AddressSanitizer + UBSan at edge-case shapes and coordinates;
* the library's host layer (argument checks, layout and workspace arithmetic of raft_amd/csrc/*.hip) built with
  `-Xarch_host -fsanitize=address,undefined`, driven without a GPU by tests/sanitize/host_driver.c.
Both need only the compilers; no GPU, no torch."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_c_oracle_under_asan_ubsan(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_asan")
    subprocess.run(["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fsanitize=float-cast-overflow",
                    "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", os.path.join(HERE, "sanitize", "oracle_driver.c"),
                    os.path.join(ROOT, "oracle", "alt_corr_oracle.c"), "-lm", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 bad" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_library_host_layer_under_asan_ubsan():
    if not os.path.exists("/opt/rocm/bin/hipcc") or os.environ.get("RAFTCORR_SKIP_HOST_ASAN") == "1":
        pytest.skip("no hipcc (or RAFTCORR_SKIP_HOST_ASAN=1)")
    r = subprocess.run(["bash", os.path.join(HERE, "sanitize", "build_host_asan.sh")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "host layer under asan+ubsan: 0 failures" in r.stdout, r.stdout[-2000:]
    assert "all refused" in r.stdout or "bad-argument section skipped" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
