"""CPU sanitizer leg (SURVEY.md §5 "race detection / sanitizers"): the oracle and the HIP library's host-side
arithmetic are built with -fsanitize=address,undefined and run on small, ragged and degenerate inputs.
(GPU AddressSanitizer is not available on this pool; device code is covered by the parity tests.)"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _run(cmd_build, exe):
    subprocess.check_call(cmd_build)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    return r.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_driver")
    out = _run(["gcc", "-std=c11", "-ffp-contract=off", *SAN, "-o", exe, os.path.join(ROOT, "tests", "sanitize", "oracle_driver.c"),
                os.path.join(ROOT, "oracle", "kde_oracle.c"), "-lm"], exe)
    assert "oracle driver ok" in out


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_threshold_code_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_driver")
    obj = str(tmp_path / "oracle.o")
    subprocess.check_call(["gcc", "-std=c11", "-ffp-contract=off", *SAN, "-c", "-o", obj, os.path.join(ROOT, "oracle", "kde_oracle.c")])
    out = _run(["g++", "-std=c++17", "-ffp-contract=off", *SAN, "-o", exe, os.path.join(ROOT, "tests", "sanitize", "host_driver.cpp"),
                obj, "-lm"], exe)
    assert "host driver ok" in out
