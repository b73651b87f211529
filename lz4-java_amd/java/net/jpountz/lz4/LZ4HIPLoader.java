package net.jpountz.lz4;

/*
 * Adapted from lz4-java (src/java/net/jpountz/util/Native.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

/** Public hook so net.jpountz.xxhash can trigger the one-time load of liblz4hip-java.so. */
public final class LZ4HIPLoader {
  private LZ4HIPLoader() {}
  public static void load() { NativeHIP.load(); }
}
