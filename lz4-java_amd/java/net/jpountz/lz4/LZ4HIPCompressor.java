package net.jpountz.lz4;

/*
 * Adapted from lz4-java (src/java/net/jpountz/lz4/LZ4JNICompressor.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
 * for the "HIP" implementation family: the argument checks, their order and the exceptions are the original's; the native
 * call goes to liblz4hip instead of liblz4.
 *
 * Licensed under the Apache License, Version 2.0 (the "License");
 * you may not use this file except in compliance with the License.
 * You may obtain a copy of the License at
 *
 *     http://www.apache.org/licenses/LICENSE-2.0
 *
 * Unless required by applicable law or agreed to in writing, software
 * distributed under the License is distributed on an "AS IS" BASIS,
 * WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or implied.
 * See the License for the specific language governing permissions and
 * limitations under the License.
 */

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
