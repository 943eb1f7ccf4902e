"""Sanitizer tier: the reducer core and the loopback fabric (mutex, condition variables, threads as ranks, hook
calls from several threads) under AddressSanitizer, ThreadSanitizer and UndefinedBehaviorSanitizer.  Host code
only - the stress driver is a plain executable, no GPU and no Python in the instrumented process."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.sanitize


def _build(kind):
    spec = importlib.util.spec_from_file_location("_pdt_build", os.path.join(ROOT, "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return str(mod.build_sanitized(kind))


@pytest.mark.parametrize("kind", ["address", "thread", "undefined"])
def test_reducer_core_and_loopback_are_clean_under(kind):
    exe = _build(kind)
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=1 exitcode=66"
    env["ASAN_OPTIONS"] = "detect_leaks=1 exitcode=67"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    out = subprocess.run([exe, "4", "12"], env=env, capture_output=True, text=True, timeout=300)
    report = out.stdout + out.stderr
    assert out.returncode == 0, report[-4000:]
    assert "reducer_stress: ok" in out.stdout
    for needle in ("ThreadSanitizer", "AddressSanitizer", "LeakSanitizer", "runtime error"):
        assert needle not in report, report[-4000:]
