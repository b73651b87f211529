#!/usr/bin/env python3
"""Extracts the regression input of LZ4Test.testRoundtripIssue12 (/root/reference/src/test/net/jpountz/lz4/LZ4Test.java:487-541,
https://github.com/jpountz/lz4-java/issues/12) into tests/golden/issue12.bin.  Run in the build container (the reference is not on
the GPU box); the test round-trips bytes [9:] as the reference does (`testRoundTrip(data, 9, data.length - 9)`)."""
import os, re
here = os.path.dirname(os.path.abspath(__file__))
src = open("/root/reference/src/test/net/jpountz/lz4/LZ4Test.java").read()
m = re.search(r"testRoundtripIssue12\(\)\s*\{\s*byte\[\]\s*data\s*=\s*new\s*byte\[\]\s*\{(.*?)\};", src, re.S)
vals = [int(x) & 0xFF for x in re.findall(r"-?\d+", m.group(1))]
open(os.path.join(here, "issue12.bin"), "wb").write(bytes(vals))
print(len(vals), "bytes")
