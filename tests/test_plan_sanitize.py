"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side plan compiler (csrc/plan.cpp): random
radial / meshed / complete graphs through plan creation, every read-back array and destruction
(tests/native/plan_sanitize.cpp).  GPU sanitizers are not available on the pool; the plan compiler is where
the index arithmetic of the kernels' descriptors is made."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_plan_compiler_is_clean_under_asan_and_ubsan():
    r = subprocess.run(['bash', os.path.join(ROOT, 'scripts', 'sanitize_plan.sh')], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and 'clean under ASan/UBSan' in r.stdout, r.stdout[-3000:]
    assert 'runtime error' not in r.stdout and 'AddressSanitizer' not in r.stdout
