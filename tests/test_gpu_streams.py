"""The stream/container twins on the product path (-m gpu): same cases as tests/test_streams_host.py, with every
compress / decompress / xxh32 served by liblz4hip batches; the oracle is only the checker inside the cases."""
import importlib
import io

import pytest

import streams_common as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(amd):
    return importlib.import_module("lz4-java_amd.streams")


@pytest.fixture(scope="module")
def engine(S):
    return S.HIPEngine()


@pytest.fixture(scope="module")
def data(corpus, O):
    return sc.payload(corpus, O)


def test_frame_layout_and_roundtrip(S, engine, port, data):
    sc.case_frame_layout_and_roundtrip(S, engine, port, data)


def test_frame_size_sweep(S, engine, data):
    sc.case_frame_size_sweep(S, engine, data)


def test_frame_known_header_bytes(S, engine):
    sc.case_frame_known_header_bytes(S, engine)


def test_frame_flush_and_bytewise(S, engine, port, data):
    sc.case_frame_flush_and_bytewise(S, engine, port, data)


def test_frame_concat_skippable_single(S, engine, data):
    sc.case_frame_concat_skippable_single(S, engine, data)


def test_frame_errors(S, engine, data):
    sc.case_frame_errors(S, engine, data)


@pytest.mark.skipif(sc.LZ4_CLI is None, reason="lz4 CLI not installed")
def test_frame_cli_interop(S, engine, data):
    sc.case_frame_cli_interop(S, engine, data)


def test_frame_hc(S, port, data):
    """HC level 9 behind the frame container: blocks are bit-exact LZ4_compress_HC output"""
    sink = io.BytesIO()
    f = S.LZ4FrameOutputStream(sink, S.BLOCKSIZE.SIZE_256KB, -1, engine=S.HIPEngine(hcLevel=9))
    f.write(data)
    f.close()
    _, _, _, _, blocks, _, _ = sc.parse_frame(sink.getvalue())
    for i, (stored_raw, body, _) in enumerate(blocks):
        raw = data[i << 18:(i + 1) << 18]
        comp = port.compress_hc(raw, 9)
        assert (stored_raw and body == raw) if len(comp) >= len(raw) else (not stored_raw and body == comp)
    assert S.LZ4FrameInputStream(io.BytesIO(sink.getvalue())).read() == data


def test_block_stream(S, engine, port, data):
    sc.case_block_stream(S, engine, port, data)


def test_with_length(S, engine, port, data):
    sc.case_with_length(S, engine, port, data, hc_engine=S.HIPEngine(hcLevel=9))


def test_cpp_stream_mirror_runs_and_interoperates(S, engine):
    """the C++ twins (lz4-java_amd/host/lz4hip_streams.hpp): their own checks pass, and the containers they write are the
    Python twin's byte for byte, decode with the lz4 CLI, and a CLI frame decodes with them"""
    import os
    import subprocess
    import tempfile
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "stream_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "stream_mirror_test.cpp"),
                           "-L" + os.path.join(ROOT, "lz4-java_amd"), "-llz4hip", "-Wl,-rpath," + os.path.join(ROOT, "lz4-java_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    with tempfile.TemporaryDirectory() as d:
        cli_payload = bytes(range(256)) * 3000 + b"tail"
        if sc.LZ4_CLI:
            open(os.path.join(d, "cli_payload.bin"), "wb").write(cli_payload)
            open(os.path.join(d, "cli.lz4"), "wb").write(sc.cli(["-1", "-B5", "-BX"], cli_payload))
        assert subprocess.call([exe, d]) == 0
        data = open(os.path.join(d, "cpp_payload.bin"), "rb").read()
        B = S.FLG.Bits
        for name, bs, bits, known in (("cpp_frame_default.lz4", S.BLOCKSIZE.SIZE_4MB, (B.BLOCK_INDEPENDENCE,), -1),
                                      ("cpp_frame_all.lz4", S.BLOCKSIZE.SIZE_64KB,
                                       (B.BLOCK_INDEPENDENCE, B.BLOCK_CHECKSUM, B.CONTENT_CHECKSUM, B.CONTENT_SIZE), len(data)),
                                      ("cpp_frame_256k_cc.lz4", S.BLOCKSIZE.SIZE_256KB, (B.BLOCK_INDEPENDENCE, B.CONTENT_CHECKSUM), -1)):
            fr = open(os.path.join(d, name), "rb").read()
            assert fr == sc.frame_bytes(S, data, engine, bs, bits, known), name
            assert S.LZ4FrameInputStream(io.BytesIO(fr), engine=engine).read() == data
            if sc.LZ4_CLI:
                assert sc.cli(["-d"], fr) == data
        blk = open(os.path.join(d, "cpp_stream.blk"), "rb").read()
        assert blk == sc.block_stream_bytes(S, data, engine, 1 << 16)
        assert S.LZ4BlockInputStream(io.BytesIO(blk), engine=engine).read() == data
