"""The C ABI's multi-device host path on ONE GPU (-m gpu): lz4hip_init accepts a device list with repeats, so [0, 0] / [0, 0, 0]
drive csrc/api.cpp's D > 1 branch -- contiguous block ranges per listed device, one thread + staging set each (SURVEY.md 8(e);
block independence: /root/reference/src/java/net/jpountz/lz4/LZ4FrameOutputStream.java:361-363) -- which no 1-GPU run reaches
otherwise.  Each case runs in its own process (lz4hip_init is once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("D", [2, 3])
def test_host_batch_over_repeated_device_list(D):
    r = subprocess.run([sys.executable, os.path.join(HERE, "multidev_child.py"), str(D)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ("multidev ok D=%d" % D) in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
