package net.jpountz.lz4;

import static net.jpountz.util.ByteBufferUtils.checkNotReadOnly;
import static net.jpountz.util.ByteBufferUtils.checkRange;
import static net.jpountz.util.SafeUtils.checkRange;

import java.nio.ByteBuffer;

/**
 * Fast {@link LZ4Compressor} of the "HIP" family: byte-identical output to {@code LZ4JNICompressor}
 * (liblz4 1.9.3 LZ4_compress_default), computed on the GPU.  Same checks, same order and same
 * exception as LZ4JNICompressor.java:35-82.
 */
final class LZ4HIPCompressor extends LZ4Compressor {

  public static final LZ4Compressor INSTANCE = new LZ4HIPCompressor();
  private static LZ4Compressor SAFE_INSTANCE;

  @Override
  public int compress(byte[] src, int srcOff, int srcLen, byte[] dest, int destOff, int maxDestLen) {
    checkRange(src, srcOff, srcLen);
    checkRange(dest, destOff, maxDestLen);
    final int result = LZ4HIPJNI.LZ4HIP_compress_fast(src, null, srcOff, srcLen, dest, null, destOff, maxDestLen);
    if (result <= 0) {
      throw new LZ4Exception(result == 0 ? "maxDestLen is too small" : "liblz4hip: " + LZ4HIPJNI.lastError());
    }
    return result;
  }

  @Override
  public int compress(ByteBuffer src, int srcOff, int srcLen, ByteBuffer dest, int destOff, int maxDestLen) {
    checkNotReadOnly(dest);
    checkRange(src, srcOff, srcLen);
    checkRange(dest, destOff, maxDestLen);

    if ((src.hasArray() || src.isDirect()) && (dest.hasArray() || dest.isDirect())) {
      byte[] srcArr = null, destArr = null;
      ByteBuffer srcBuf = null, destBuf = null;
      if (src.hasArray()) {
        srcArr = src.array();
        srcOff += src.arrayOffset();
      } else {
        srcBuf = src;
      }
      if (dest.hasArray()) {
        destArr = dest.array();
        destOff += dest.arrayOffset();
      } else {
        destBuf = dest;
      }
      final int result = LZ4HIPJNI.LZ4HIP_compress_fast(srcArr, srcBuf, srcOff, srcLen, destArr, destBuf, destOff, maxDestLen);
      if (result <= 0) {
        throw new LZ4Exception(result == 0 ? "maxDestLen is too small" : "liblz4hip: " + LZ4HIPJNI.lastError());
      }
      return result;
    } else {
      // neither array-backed nor direct: same escape hatch as the JNI family (LZ4JNICompressor.java:75-80)
      LZ4Compressor safeInstance = SAFE_INSTANCE;
      if (safeInstance == null) {
        safeInstance = SAFE_INSTANCE = LZ4Factory.safeInstance().fastCompressor();
      }
      return safeInstance.compress(src, srcOff, srcLen, dest, destOff, maxDestLen);
    }
  }
}
