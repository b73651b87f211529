package net.jpountz.xxhash;

/*
 * Adapted from lz4-java (src/java/net/jpountz/xxhash/StreamingXXHash64JNI.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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
 * "HIP" family member of {@link StreamingXXHash64} (twin of StreamingXXHash64JNI.java:28-104).  The state lives in a device
 * record behind a native handle; every update continues it with one launch.  {@code XXHashFactory.instance("HIP")} finds the
 * nested Factory by name (XXHashFactory.java:179-182).  Methods are synchronized for the same reason as in the JNI twin:
 * finalize() may free the native state concurrently.
 */
final class StreamingXXHash64HIP extends StreamingXXHash64 {

  static class Factory implements StreamingXXHash64.Factory {

    public static final StreamingXXHash64.Factory INSTANCE = new Factory();

    @Override
    public StreamingXXHash64 newStreamingHash(long seed) {
      return new StreamingXXHash64HIP(seed);
    }

  }

  private long state;

  StreamingXXHash64HIP(long seed) {
    super(seed);
    state = XXHashHIPJNI.XXH64_init(seed);
  }

  private void checkState() {
    if (state == 0) {
      throw new AssertionError("Already finalized");
    }
  }

  @Override
  public synchronized void reset() {
    checkState();
    XXHashHIPJNI.XXH64_reset(state, seed);   // keeps the device record, restarts it with the same seed
  }

  @Override
  public synchronized long getValue() {
    checkState();
    return XXHashHIPJNI.XXH64_digest(state);
  }

  @Override
  public synchronized void update(byte[] bytes, int off, int len) {
    checkState();
    net.jpountz.util.SafeUtils.checkRange(bytes, off, len);
    XXHashHIPJNI.XXH64_update(state, bytes, off, len);
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
