"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer ("race
detection / sanitizers"): `make -C oracle asan` builds liblpr_oracle_asan.so and the oracle's own
CPU tests run against it in a child interpreter with the sanitizer runtime preloaded.  Sanitizers
run on the CPU build only -- GPU AddressSanitizer / XNACK runs are not available on the pool."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True,
                          text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), "libasan.so not found next to gcc"
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True,
                   capture_output=True)
    env = dict(os.environ)
    env.update({"LD_PRELOAD": asan, "LPR_ORACLE_SANITIZE": "1",
                # CPython "leaks" by design; everything else stops the run with a report
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    tests = ["tests/test_oracle_primal.py", "tests/test_oracle_revised.py",
             "tests/test_oracle_bb.py", "tests/test_oracle_cut.py", "tests/test_properties.py"]
    proc = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                           "-m", "not gpu",
                           "-k", "not host_formatter", *tests],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (proc.stdout + proc.stderr)[-3000:]
    assert proc.returncode == 0, tail
    assert "passed" in proc.stdout and "ERROR: AddressSanitizer" not in tail \
        and "runtime error" not in tail, tail
