package net.jpountz.xxhash;

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
