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

import java.io.File;
import java.io.FileOutputStream;
import java.io.IOException;
import java.io.InputStream;

/** Loader for liblz4hip-java.so; the extraction logic follows net.jpountz.util.Native (Native.java:98-162). */
final class NativeHIP {
  private static boolean loaded = false;

  private NativeHIP() {}

  static synchronized void load() {
    if (loaded) {
      return;
    }
    try {
      System.loadLibrary("lz4hip-java");
      loaded = true;
      return;
    } catch (UnsatisfiedLinkError e) {
      // fall through to the bundled copy
    }
    final String resource = "/net/jpountz/util/linux/amd64/liblz4hip-java.so";
    try (InputStream is = NativeHIP.class.getResourceAsStream(resource)) {
      if (is == null) {
        throw new UnsupportedOperationException("liblz4hip-java.so is not bundled for this platform (linux/amd64 + MI355X only)");
      }
      File tmp = File.createTempFile("liblz4hip-java-", ".so");
      tmp.deleteOnExit();
      try (FileOutputStream out = new FileOutputStream(tmp)) {
        byte[] buf = new byte[8192];
        for (int n; (n = is.read(buf)) > 0; ) {
          out.write(buf, 0, n);
        }
      }
      System.load(tmp.getAbsolutePath());
      loaded = true;
    } catch (IOException e) {
      throw new ExceptionInInitializerError("Cannot unpack liblz4hip-java.so: " + e);
    }
  }
}
