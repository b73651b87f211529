package net.jpountz.xxhash;

import static net.jpountz.util.ByteBufferUtils.checkRange;
import static net.jpountz.util.SafeUtils.checkRange;

import java.nio.ByteBuffer;

/** "HIP" family member of {@link XXHash64} (twin of XXHash64JNI.java:24-50); {@code XXHashFactory.hipInstance()} finds it via INSTANCE. */
final class XXHash64HIP extends XXHash64 {

  public static final XXHash64 INSTANCE = new XXHash64HIP();

  @Override
  public long hash(byte[] buf, int off, int len, long seed) {
    checkRange(buf, off, len);
    return XXHashHIPJNI.XXH64(buf, off, len, seed);
  }

  @Override
  public long hash(ByteBuffer buf, int off, int len, long seed) {
    if (buf.isDirect()) {
      checkRange(buf, off, len);
      return XXHashHIPJNI.XXH64BB(buf, off, len, seed);
    } else if (buf.hasArray()) {
      return hash(buf.array(), off + buf.arrayOffset(), len, seed);
    }
    return XXHashFactory.safeInstance().hash64().hash(buf, off, len, seed);
  }
}
