package net.jpountz.lz4;

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

  static native String lastError();
}
