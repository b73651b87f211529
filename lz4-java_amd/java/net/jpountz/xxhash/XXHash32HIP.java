package net.jpountz.xxhash;

/*
 * Adapted from lz4-java (src/java/net/jpountz/xxhash/XXHash32JNI.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

import static net.jpountz.util.ByteBufferUtils.checkRange;
import static net.jpountz.util.SafeUtils.checkRange;

import java.nio.ByteBuffer;

/** "HIP" family member of {@link XXHash32} (twin of XXHash32JNI.java:24-50); {@code XXHashFactory.hipInstance()} finds it via INSTANCE. */
final class XXHash32HIP extends XXHash32 {

  public static final XXHash32 INSTANCE = new XXHash32HIP();

  @Override
  public int hash(byte[] buf, int off, int len, int seed) {
    checkRange(buf, off, len);
    return XXHashHIPJNI.XXH32(buf, off, len, seed);
  }

  @Override
  public int hash(ByteBuffer buf, int off, int len, int seed) {
    if (buf.isDirect()) {
      checkRange(buf, off, len);
      return XXHashHIPJNI.XXH32BB(buf, off, len, seed);
    } else if (buf.hasArray()) {
      return hash(buf.array(), off + buf.arrayOffset(), len, seed);
    }
    return XXHashFactory.safeInstance().hash32().hash(buf, off, len, seed);
  }
}
