"""GPU: the phased three-stream schedule of HotPath against its one-stream results on odd configurations (batch sizes that
are not multiples of 8, int64 indices, Shape B with its multi-scale [features, xyz] layout, the FPS identity shortcut):
tools/hotpath_check.py runs five back-to-back pipelined steps over alternating inputs per configuration and compares
every output tensor bit for bit."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_phased_schedule_equals_one_stream_on_odd_configurations(dev):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "hotpath_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "TOTAL mismatches 0" in r.stdout
