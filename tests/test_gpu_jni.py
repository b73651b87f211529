"""The JNI shim executed on the GPU box without a JVM: tests/jni_stub/fake_jni.c provides the JNIEnv function table (arrays =
malloc'd buffers with pin accounting, direct buffers = pointers) and drives every Java_net_jpountz_* entry point of
lz4-java_amd/jni/net_jpountz_lz4_LZ4HIPJNI.c against liblz4hip.so: LZ4Factory's 20-byte constructor self-test vector
(LZ4Factory.java:176-220), heap / direct / mixed arguments with offsets (AbstractLZ4Test.java:66-116), the `out cannot be pinned`
path the reference leaks on (LZ4JNI.c:59-73), the batch entry, xxhash one-shot / batch / streaming with SURVEY App. D's known
answers, and a library failure inside a hash call (exception, not hash 0)."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_jni_shim_full_scenarios():
    d = os.path.join(ROOT, "tests", "jni_stub")
    subprocess.check_call(["bash", os.path.join(d, "build.sh")])
    out = subprocess.check_output([os.path.join(d, "fake_jni")], timeout=300).decode()
    assert "checks ok" in out and "no device" not in out, out
