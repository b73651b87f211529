package net.jpountz.xxhash;

/*
 * Adapted from lz4-java (src/java/net/jpountz/xxhash/XXHashJNI.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

/** JNI bindings to liblz4hip's hashes (twin of XXHashJNI.java:24-48): one-shot, streaming state, and the batch calls. */
enum XXHashHIPJNI {
  ;

  static {
    net.jpountz.lz4.LZ4HIPLoader.load();
  }

  static native int XXH32(byte[] input, int offset, int len, int seed);
  static native int XXH32BB(ByteBuffer input, int offset, int len, int seed);
  /** streaming state (XXHashJNI.java:33-36, :40-43): a native handle; 0 + a pending OutOfMemoryError when it cannot be created */
  static native long XXH32_init(int seed);
  static native void XXH32_reset(long state, int seed);
  static native void XXH32_update(long state, byte[] input, int offset, int len);
  static native int XXH32_digest(long state);
  static native long XXH64_init(long seed);
  static native void XXH64_reset(long state, long seed);
  static native void XXH64_update(long state, byte[] input, int offset, int len);
  static native long XXH64_digest(long state);
  static native void XXH_free(long state);
  static native long XXH64(byte[] input, int offset, int len, long seed);
  static native long XXH64BB(ByteBuffer input, int offset, int len, long seed);
  /** n buffers of a direct ByteBuffer in one launch: out32/out64 receive the hashes */
  static native int XXH32Batch(ByteBuffer input, long[] off, int[] len, int seed, int[] out32, int n);
  static native int XXH64Batch(ByteBuffer input, long[] off, int[] len, long seed, long[] out64, int n);
}
