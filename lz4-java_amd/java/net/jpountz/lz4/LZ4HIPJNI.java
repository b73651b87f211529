package net.jpountz.lz4;

/*
 * Adapted from lz4-java (src/java/net/jpountz/lz4/LZ4JNI.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

/**
 * JNI bindings to liblz4hip (include/lz4hip.h), the MI355X LZ4 block engine.
 *
 * Same shape as the reference's {@code LZ4JNI} enum (src/java/net/jpountz/lz4/LZ4JNI.java:27-43):
 * a constant-less enum whose static initialiser loads the native library, and static native
 * methods taking {@code (array | direct buffer, offset, length)} pairs.  The single-block methods
 * back the {@link LZ4Compressor}/{@link LZ4FastDecompressor}/{@link LZ4SafeDecompressor}
 * signatures; the {@code *Batch} methods are the new entry point (many independent blocks per
 * HIP launch) used by {@link LZ4HIPBatch}.
 *
 * NOT COMPILED in the build image (no JDK there); see INTEGRATION.md.
 */
enum LZ4HIPJNI {
  ;

  static {
    // mirrors net.jpountz.util.Native.load() (Native.java:98-162): java.library.path first, then the
    // copy bundled under /net/jpountz/util/linux/amd64/liblz4hip-java.so
    NativeHIP.load();
    init();
  }

  static native void init();

  /* single block: return conventions of liblz4 (0 / negative = failure), or <= Integer.MIN_VALUE + 63 for a
   * library failure (no GPU, HIP error), which the callers turn into an LZ4Exception as well */
  static native int LZ4HIP_compress_fast(byte[] srcArray, ByteBuffer srcBuffer, int srcOff, int srcLen,
                                         byte[] destArray, ByteBuffer destBuffer, int destOff, int maxDestLen);
  static native int LZ4HIP_compressHC(byte[] srcArray, ByteBuffer srcBuffer, int srcOff, int srcLen,
                                      byte[] destArray, ByteBuffer destBuffer, int destOff, int maxDestLen, int compressionLevel);
  static native int LZ4HIP_decompress_fast(byte[] srcArray, ByteBuffer srcBuffer, int srcOff, int srcCap,
                                           byte[] destArray, ByteBuffer destBuffer, int destOff, int destLen);
  static native int LZ4HIP_decompress_safe(byte[] srcArray, ByteBuffer srcBuffer, int srcOff, int srcLen,
                                           byte[] destArray, ByteBuffer destBuffer, int destOff, int maxDestLen);
  static native int LZ4HIP_compressBound(int len);

  /* batches over DIRECT buffers (so the shim never pins the Java heap across a kernel):
   * op 0 = fast compress, 1 = safe decompress, 2 = fast decompress, 3 = HC compress(level).
   * Returns 0 or a negative lz4hip_status; per-block results land in outLen. */
  static native int LZ4HIP_batch(int op, int level, ByteBuffer src, long[] srcOff, int[] srcLen,
                                 ByteBuffer dest, long[] destOff, int[] destCap, int[] outLen, int nBlocks);

  /** Container blocks assembled on the device (LZ4HIPBatch.containerBlocks); returns bytes written or the negative lz4hip_status. */
  static native long LZ4HIP_containerBlocks(int kind, int flags, int level, ByteBuffer src, long srcOff, long len, int blockSize,
      ByteBuffer dest, long destOff, long destCap);

  /** The read side on the device (LZ4HIPBatch.containerDecode); returns 0 or the negative lz4hip_status. */
  static native int LZ4HIP_containerDecode(int kind, int flags, ByteBuffer src, long srcOff, long len, int maxBlock, int nMax,
      ByteBuffer dest, long destOff, long destCap, int[] sizes, long[] info);

  /** Destination bytes a containerDecode call can need (host-side walk of the headers); blocks[0] = whole blocks present. */
  static native long LZ4HIP_containerDecodeBound(int kind, int flags, ByteBuffer src, long srcOff, long len, int maxBlock, int nMax,
      int[] blocks);

  static native String lastError();
}
