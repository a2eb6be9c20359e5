"""Live checks against the UNMODIFIED reference where it is present (the build container: /root/reference; absent on the GPU box,
where these tests skip): a short run of the randomized campaigns whose full logs are committed under profiles/r05/ -
``scripts/r5/fuzz_oracle_vs_reference.py`` (the oracle's forwards and chains equal the reference's bit for bit) and
``scripts/r5/fuzz_glue_vs_reference.py`` (the PRODUCT's collate / templates / DDPM.sample_chain glue, noise schedules and per-step
scalars equal the reference's bit for bit).  The committed golden fixtures (tests/golden/) stay the portable pin."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason='the reference checkout is not on this machine')


def _run(script, *args):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'r5', script), *args], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_oracle_equals_the_reference_on_random_cases():
    log = _run('fuzz_oracle_vs_reference.py', '--cases', '60', '--seed', '11')
    last = log.strip().splitlines()[-1]
    assert '60 cases' in last and ' 0 failures' in last, last
    assert 'FAIL' not in log


def test_host_glue_equals_the_reference_on_random_batches():
    log = _run('fuzz_glue_vs_reference.py', '--cases', '60', '--seed', '11')
    lines = log.strip().splitlines()
    assert '0 differences from the reference' in lines[0], lines[0]
    assert '60 random batches' in lines[-1] and ' 0 failures' in lines[-1], lines[-1]
    assert 'FAIL' not in log
