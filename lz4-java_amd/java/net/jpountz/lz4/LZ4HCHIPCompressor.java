package net.jpountz.lz4;

/*
 * Adapted from lz4-java (src/java/net/jpountz/lz4/LZ4HCJNICompressor.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

import static net.jpountz.lz4.LZ4Constants.DEFAULT_COMPRESSION_LEVEL;
import static net.jpountz.util.ByteBufferUtils.checkNotReadOnly;
import static net.jpountz.util.ByteBufferUtils.checkRange;
import static net.jpountz.util.SafeUtils.checkRange;

import java.nio.ByteBuffer;

/**
 * High-compression {@link LZ4Compressor} of the "HIP" family (twin of LZ4HCJNICompressor.java:30-89).
 * LZ4Factory needs the INSTANCE field and the declared (int) constructor (LZ4Factory.java:193-202).
 */
final class LZ4HCHIPCompressor extends LZ4Compressor {

  public static final LZ4HCHIPCompressor INSTANCE = new LZ4HCHIPCompressor();

  private final int compressionLevel;

  LZ4HCHIPCompressor() { this(DEFAULT_COMPRESSION_LEVEL); }
  LZ4HCHIPCompressor(int compressionLevel) {
    this.compressionLevel = compressionLevel;
  }

  @Override
  public int compress(byte[] src, int srcOff, int srcLen, byte[] dest, int destOff, int maxDestLen) {
    checkRange(src, srcOff, srcLen);
    checkRange(dest, destOff, maxDestLen);
    final int result = LZ4HIPJNI.LZ4HIP_compressHC(src, null, srcOff, srcLen, dest, null, destOff, maxDestLen, compressionLevel);
    if (result <= 0) {
      throw new LZ4Exception();
    }
    return result;
  }

  @Override
  public int compress(ByteBuffer src, int srcOff, int srcLen, ByteBuffer dest, int destOff, int maxDestLen) {
    checkNotReadOnly(dest);
    checkRange(src, srcOff, srcLen);
    checkRange(dest, destOff, maxDestLen);
    if ((src.hasArray() || src.isDirect()) && (dest.hasArray() || dest.isDirect())) {
      final byte[] srcArr = src.hasArray() ? src.array() : null;
      final byte[] destArr = dest.hasArray() ? dest.array() : null;
      final int so = srcArr != null ? srcOff + src.arrayOffset() : srcOff;
      final int dof = destArr != null ? destOff + dest.arrayOffset() : destOff;
      final int result = LZ4HIPJNI.LZ4HIP_compressHC(srcArr, srcArr == null ? src : null, so, srcLen,
                                                     destArr, destArr == null ? dest : null, dof, maxDestLen, compressionLevel);
      if (result <= 0) {
        throw new LZ4Exception();
      }
      return result;
    }
    return LZ4Factory.safeInstance().highCompressor(compressionLevel).compress(src, srcOff, srcLen, dest, destOff, maxDestLen);
  }
}
