package net.jpountz.lz4;

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
