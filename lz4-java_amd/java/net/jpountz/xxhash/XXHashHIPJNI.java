package net.jpountz.xxhash;

import java.nio.ByteBuffer;

/** JNI bindings to liblz4hip's one-shot hashes (twin of XXHashJNI.java; streaming state is out of scope). */
enum XXHashHIPJNI {
  ;

  static {
    net.jpountz.lz4.LZ4HIPLoader.load();
  }

  static native int XXH32(byte[] input, int offset, int len, int seed);
  static native int XXH32BB(ByteBuffer input, int offset, int len, int seed);
  static native long XXH64(byte[] input, int offset, int len, long seed);
  static native long XXH64BB(ByteBuffer input, int offset, int len, long seed);
  /** n buffers of a direct ByteBuffer in one launch: out32/out64 receive the hashes */
  static native int XXH32Batch(ByteBuffer input, long[] off, int[] len, int seed, int[] out32, int n);
  static native int XXH64Batch(ByteBuffer input, long[] off, int[] len, long seed, long[] out64, int n);
}
