package net.jpountz.xxhash;

/*
 * Adapted from lz4-java (src/java/net/jpountz/xxhash/StreamingXXHash32JNI.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

/**
 * "HIP" family member of {@link StreamingXXHash32} (twin of StreamingXXHash32JNI.java:28-104).  The state lives in a device
 * record behind a native handle; every update continues it with one launch.  {@code XXHashFactory.instance("HIP")} finds the
 * nested Factory by name (XXHashFactory.java:179-182).  Methods are synchronized for the same reason as in the JNI twin:
 * finalize() may free the native state concurrently.
 */
final class StreamingXXHash32HIP extends StreamingXXHash32 {

  static class Factory implements StreamingXXHash32.Factory {

    public static final StreamingXXHash32.Factory INSTANCE = new Factory();

    @Override
    public StreamingXXHash32 newStreamingHash(int seed) {
      return new StreamingXXHash32HIP(seed);
    }

  }

  private long state;

  StreamingXXHash32HIP(int seed) {
    super(seed);
    state = XXHashHIPJNI.XXH32_init(seed);
  }

  private void checkState() {
    if (state == 0) {
      throw new AssertionError("Already finalized");
    }
  }

  @Override
  public synchronized void reset() {
    checkState();
    XXHashHIPJNI.XXH32_reset(state, seed);   // keeps the device record, restarts it with the same seed
  }

  @Override
  public synchronized int getValue() {
    checkState();
    return XXHashHIPJNI.XXH32_digest(state);
  }

  @Override
  public synchronized void update(byte[] bytes, int off, int len) {
    checkState();
    net.jpountz.util.SafeUtils.checkRange(bytes, off, len);
    XXHashHIPJNI.XXH32_update(state, bytes, off, len);
  }

  @Override
  public synchronized void close() {
    if (state != 0) {
      super.close();
      XXHashHIPJNI.XXH_free(state);
      state = 0;
    }
  }

  @Override
  protected synchronized void finalize() throws Throwable {
    super.finalize();
    if (state != 0) {
      XXHashHIPJNI.XXH_free(state);
      state = 0;
    }
  }

}
