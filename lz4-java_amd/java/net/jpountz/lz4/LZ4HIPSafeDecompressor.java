package net.jpountz.lz4;

/*
 * Adapted from lz4-java (src/java/net/jpountz/lz4/LZ4JNISafeDecompressor.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

import java.nio.ByteBuffer;

import net.jpountz.util.ByteBufferUtils;
import net.jpountz.util.SafeUtils;

/**
 * {@link LZ4SafeDecompressor} of the "HIP" family; return codes and the exception message equal
 * LZ4JNISafeDecompressor.java:34-43 (liblz4 1.9.3 LZ4_decompress_safe), including on malformed input.
 */
final class LZ4HIPSafeDecompressor extends LZ4SafeDecompressor {

  public static final LZ4HIPSafeDecompressor INSTANCE = new LZ4HIPSafeDecompressor();

  @Override
  public final int decompress(byte[] src, int srcOff, int srcLen, byte[] dest, int destOff, int maxDestLen) {
    SafeUtils.checkRange(src, srcOff, srcLen);
    SafeUtils.checkRange(dest, destOff, maxDestLen);
    final int result = LZ4HIPJNI.LZ4HIP_decompress_safe(src, null, srcOff, srcLen, dest, null, destOff, maxDestLen);
    if (result < 0) {
      throw new LZ4Exception(result <= Integer.MIN_VALUE + 63 ? "liblz4hip: " + LZ4HIPJNI.lastError()
                                                               : "Error decoding offset " + (srcOff - result) + " of input buffer");
    }
    return result;
  }

  @Override
  public int decompress(ByteBuffer src, int srcOff, int srcLen, ByteBuffer dest, int destOff, int maxDestLen) {
    ByteBufferUtils.checkNotReadOnly(dest);
    ByteBufferUtils.checkRange(src, srcOff, srcLen);
    ByteBufferUtils.checkRange(dest, destOff, maxDestLen);
    if ((src.hasArray() || src.isDirect()) && (dest.hasArray() || dest.isDirect())) {
      final byte[] srcArr = src.hasArray() ? src.array() : null;
      final byte[] destArr = dest.hasArray() ? dest.array() : null;
      final int so = srcArr != null ? srcOff + src.arrayOffset() : srcOff;
      final int dof = destArr != null ? destOff + dest.arrayOffset() : destOff;
      final int result = LZ4HIPJNI.LZ4HIP_decompress_safe(srcArr, srcArr == null ? src : null, so, srcLen,
                                                          destArr, destArr == null ? dest : null, dof, maxDestLen);
      if (result < 0) {
        throw new LZ4Exception("Error decoding offset " + (srcOff - result) + " of input buffer");
      }
      return result;
    }
    return LZ4Factory.safeInstance().safeDecompressor().decompress(src, srcOff, srcLen, dest, destOff, maxDestLen);
  }
}
