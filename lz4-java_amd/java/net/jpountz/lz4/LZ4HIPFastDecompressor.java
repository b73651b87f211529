package net.jpountz.lz4;

/*
 * Adapted from lz4-java (src/java/net/jpountz/lz4/LZ4JNIFastDecompressor.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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
 * {@link LZ4FastDecompressor} of the "HIP" family (twin of LZ4JNIFastDecompressor.java:35-82).
 * Unlike liblz4's LZ4_decompress_fast the native side is told how many source bytes are readable
 * (src.length - srcOff) and never reads past them; on valid streams the result is identical.
 */
final class LZ4HIPFastDecompressor extends LZ4FastDecompressor {

  public static final LZ4HIPFastDecompressor INSTANCE = new LZ4HIPFastDecompressor();

  @Override
  public final int decompress(byte[] src, int srcOff, byte[] dest, int destOff, int destLen) {
    SafeUtils.checkRange(src, srcOff);
    SafeUtils.checkRange(dest, destOff, destLen);
    final int result = LZ4HIPJNI.LZ4HIP_decompress_fast(src, null, srcOff, src.length - srcOff, dest, null, destOff, destLen);
    if (result < 0) {
      throw new LZ4Exception("Error decoding offset " + (srcOff - result) + " of input buffer");
    }
    return result;
  }

  @Override
  public int decompress(ByteBuffer src, int srcOff, ByteBuffer dest, int destOff, int destLen) {
    ByteBufferUtils.checkNotReadOnly(dest);
    ByteBufferUtils.checkRange(src, srcOff);
    ByteBufferUtils.checkRange(dest, destOff, destLen);
    if ((src.hasArray() || src.isDirect()) && (dest.hasArray() || dest.isDirect())) {
      final byte[] srcArr = src.hasArray() ? src.array() : null;
      final byte[] destArr = dest.hasArray() ? dest.array() : null;
      final int so = srcArr != null ? srcOff + src.arrayOffset() : srcOff;
      final int dof = destArr != null ? destOff + dest.arrayOffset() : destOff;
      final int result = LZ4HIPJNI.LZ4HIP_decompress_fast(srcArr, srcArr == null ? src : null, so, src.capacity() - srcOff,
                                                          destArr, destArr == null ? dest : null, dof, destLen);
      if (result < 0) {
        throw new LZ4Exception("Error decoding offset " + (srcOff - result) + " of input buffer");
      }
      return result;
    }
    return LZ4Factory.safeInstance().fastDecompressor().decompress(src, srcOff, dest, destOff, destLen);
  }
}
