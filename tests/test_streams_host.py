"""Container logic of the stream twins (lz4-java_amd/streams.py) on CPU: headers, flags, checksums, batching
cadence, error order and messages -- with the oracle standing in for the GPU engine (test infrastructure only;
tests/test_gpu_streams.py runs the same cases through liblz4hip).  Reference tests of the same classes:
src/test/net/jpountz/lz4/LZ4FrameIOStreamTest.java, LZ4BlockStreamingTest.java, LZ4Test.java (WithLength)."""
import importlib

import pytest

import streams_common as sc


@pytest.fixture(scope="module")
def S(amd):
    return importlib.import_module("lz4-java_amd.streams")


@pytest.fixture(scope="module")
def engine(port, O):
    return sc.OracleEngine(port, O)


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


def test_block_stream(S, engine, port, data):
    sc.case_block_stream(S, engine, port, data)


def test_with_length(S, engine, port, O, data):
    sc.case_with_length(S, engine, port, data, hc_engine=sc.OracleEngine(port, O, hcLevel=9))


@pytest.fixture(scope="module")
def dev_engine(port, O):
    return sc.OracleDeviceEngine(port, O)


def test_device_read_path_logic_on_the_cpu(S, dev_engine, port, data):
    """the readers' DEVICE-PATH logic (a chunk of container bytes to engine.containerDecode, stop reasons -> the reference's
    exceptions, unconsumed bytes handed back) with a host restatement of the device walk standing in for the GPU: the frame and
    LZ4Block cases of this file once more, through that path"""
    c0 = dev_engine.calls
    sc.case_frame_layout_and_roundtrip(S, dev_engine, port, data)
    sc.case_frame_concat_skippable_single(S, dev_engine, data)
    sc.case_frame_errors(S, dev_engine, data)
    sc.case_block_stream(S, dev_engine, port, data)
    assert dev_engine.calls - c0 > 50          # (the device path did the reading)


def test_device_read_path_advisor_findings(S, dev_engine, data):
    sc.case_device_read_path_advisor_findings(S, dev_engine, data)


def test_default_engine_is_the_gpu_batch_engine_and_fails_loudly_without_a_gpu(S, amd):
    import io
    import torch
    e = S.HIPEngine()
    assert e.decompressSafe == amd.LZ4HIPBatch.decompressSafe and e.xxh32 == amd.LZ4HIPBatch.xxh32
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_streams.py")
    with pytest.raises(amd.LZ4HIPError):   # no CPU fallback behind the streams either
        S.LZ4FrameOutputStream(io.BytesIO())
    with pytest.raises(amd.LZ4HIPError):
        S.LZ4CompressorWithLength().compress(b"abcdabcdabcdabcd")
