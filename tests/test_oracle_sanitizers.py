"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no
sanitizer leg; this build's is on the CPU side -- GPU sanitizers are not available on the pool).

``make -C oracle asan`` builds ``librl8_oracle_asan.so`` from the same source; a CHILD interpreter with the ASan
runtime preloaded (``LD_PRELOAD`` set for the child only) then runs the oracle's golden-vector tests
(tests/test_oracle_golden.py: every entry point of ``oracle/rl8_oracle.c`` on the reference's fixtures, ragged sizes
included) and the end-to-end oracle traces against that library.  Any out-of-bounds access, use after free, signed
overflow, misaligned access or invalid shift aborts the child (``-fno-sanitize-recover``).  Never run on the GPU box.
"""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name: str) -> str:
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True, check=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else ""


@pytest.mark.skipif(not _runtime("libasan.so"), reason="gcc's libasan.so is not installed")
def test_oracle_golden_vectors_are_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B", "asan"], check=True)
    lib = os.path.join(ROOT, "oracle", "_build", "librl8_oracle_asan.so")
    assert os.path.exists(lib)
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": _runtime("libasan.so"),
        # the interpreter itself leaks by design; everything else aborts with a report
        "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",
        "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
        "RL8_ORACLE_LIB": lib,
    })
    run = subprocess.run(
        [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "not gpu",
         os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_traces.py")],
        env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    report = run.stdout[-4000:] + run.stderr[-4000:]
    assert run.returncode == 0, report
    assert "AddressSanitizer" not in report and "runtime error" not in report, report
    assert " passed" in run.stdout and "failed" not in run.stdout, report
    # the child really ran the sanitized library
    probe = subprocess.run(
        [sys.executable, "-c", "from oracle import oracle; oracle.lib(); print(open('/proc/self/maps').read().count('librl8_oracle_asan.so') > 0)"],
        env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr
