"""The library's host orchestration (the api_*.hip translation units: 3 700 lines of argument checks, grouping, in-flight state machines, staging
ring, shard plans) under AddressSanitizer + UndefinedBehaviorSanitizer, in the CPU container: `make asan-host` compiles the
library's own translation units for the host only and links them against tools/fakehip (no device; allocation bookkeeping),
tests/host_asan_driver.py drives every entry point's orderings and refusals.  Any sanitizer report fails the test."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = '/opt/rocm/lib/llvm/lib/clang'


def _asan_runtime():
    for d, _, files in os.walk(RT):
        if 'libclang_rt.asan-x86_64.so' in files:
            return os.path.join(d, 'libclang_rt.asan-x86_64.so')
    return None


@pytest.mark.skipif(not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')) or _asan_runtime() is None,
                    reason='needs hipcc and clang\'s sanitizer runtime (the CPU container has both)')
def test_host_orchestration_under_asan_and_ubsan():
    r = subprocess.run(['make', '-C', ROOT, '-j6', 'asan-host'], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    env = dict(os.environ, SNK_LIBRARY=os.path.join(ROOT, 'build', 'asan', 'libsnkhip_host_asan.so'),
               LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'host_asan_driver.py')], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    report = r.stdout[-3000:] + '\n' + r.stderr[-6000:]
    assert r.returncode == 0 and 'HOST-ASAN-OK' in r.stdout, report
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error:' not in r.stderr, report
