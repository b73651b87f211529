"""A bounded slice of the volume campaigns in the driver-run suite (-m gpu): the hand-scheduled finder loop (csrc/lz4_fast_v2_asm.h)
is ISA and has no CPU simulator -- its one real bug (tests/golden/regress/r03_binary_alphabet_*.bin) was found by tools/gpu_fuzz.py
at volume, not by the suite.  These run the same scripts with fixed seeds that differ from every other test's: every compress core
with full and tight capacities, every decoder variant, HC levels on subsets (gpu_fuzz.py); long valid and damaged streams through the
deep / windowed decoder loops at 4 / 8 / 16 lanes (gpu_fuzz_deep.py); blocks of 65547 bytes and more -- the 64-bit-entry form of
the hand-scheduled loop, csrc/lz4_fast_v2_asm32.h -- with full and tight capacities (gpu_fuzz_u32.py).  Everything is compared with
the reference library."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def run(script, *args, env=None):
    r = subprocess.run([sys.executable, os.path.join(TOOLS, script)] + [str(a) for a in args], capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("seed", [31337, 424242])
def test_gpu_fuzz_slice(seed):
    out = run("gpu_fuzz.py", 2500, seed)
    assert "core 3: 2500 inputs bit-exact" in out and "decode safe/fast" in out


@pytest.mark.parametrize("pipe,ring", [(2, 0), (3, 2048), (4, 0), (4, 8192), (5, 0), (5, 8192), (7, 0), (7, 16384), (8, 0), (8, 8192), (8, 32768)])
def test_gpu_fuzz_deep_slice(pipe, ring):
    """long streams, valid and damaged, through the deep loop, the ring loop with the 2 KiB ring the routed default uses, and the wave
    loop (its default ring for the batch and the smallest one)"""
    out = run("gpu_fuzz_deep.py", 400, 2026 + pipe, env={"FUZZ_PIPE": str(pipe), "FUZZ_RING": str(ring)})
    assert "deep fuzz ok" in out


@pytest.mark.parametrize("seed", [90210, 1729])
def test_gpu_fuzz_u32_slice(seed):
    out = run("gpu_fuzz_u32.py", 300, seed, 700000, env={"U32_TIMING": "0"})
    assert "core 3: 300 byU32 inputs" in out and "core 5: 300 byU32 inputs" in out and "byU32 fuzz ok" in out
