package net.jpountz.lz4;

/** Public hook so net.jpountz.xxhash can trigger the one-time load of liblz4hip-java.so. */
public final class LZ4HIPLoader {
  private LZ4HIPLoader() {}
  public static void load() { NativeHIP.load(); }
}
