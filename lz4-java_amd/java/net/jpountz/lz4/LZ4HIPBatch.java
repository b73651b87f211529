package net.jpountz.lz4;

/*
 * Adapted from lz4-java (the batch counterpart of src/java/net/jpountz/lz4/LZ4JNI.java), Copyright 2020 Adrien Grand and the lz4-java contributors,
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

/**
 * Many independent blocks per HIP launch -- the entry point the one-block-per-call API of lz4-java
 * lacks (LZ4Compressor.java:59, SURVEY.md fact 9).  All buffers must be DIRECT: block i is
 * {@code src[srcOff[i], srcOff[i]+srcLen[i])} and owns the slot {@code dest[destOff[i], destOff[i]+destCap[i])}.
 * Results follow liblz4's conventions per block (see include/lz4hip.h).
 */
public final class LZ4HIPBatch {
  private LZ4HIPBatch() {}

  private static void check(ByteBuffer src, ByteBuffer dest, long[] srcOff, int[] srcLen, long[] destOff, int[] destCap, int[] outLen) {
    if (!src.isDirect() || !dest.isDirect()) {
      throw new IllegalArgumentException("LZ4HIPBatch needs direct ByteBuffers");
    }
    if (dest.isReadOnly()) {
      throw new java.nio.ReadOnlyBufferException();
    }
    final int n = srcOff.length;
    if (srcLen.length != n || destOff.length != n || destCap.length != n || outLen.length < n) {
      throw new IllegalArgumentException("per-block arrays must have the same length");
    }
    for (int i = 0; i < n; i++) {
      if (srcLen[i] < 0 || destCap[i] < 0 || srcOff[i] < 0 || destOff[i] < 0
          || srcOff[i] + srcLen[i] > src.capacity() || destOff[i] + destCap[i] > dest.capacity()) {
        throw new ArrayIndexOutOfBoundsException(i);
      }
    }
  }

  private static void run(int op, int level, ByteBuffer src, long[] srcOff, int[] srcLen, ByteBuffer dest, long[] destOff, int[] destCap, int[] outLen) {
    check(src, dest, srcOff, srcLen, destOff, destCap, outLen);
    final int rc = LZ4HIPJNI.LZ4HIP_batch(op, level, src, srcOff, srcLen, dest, destOff, destCap, outLen, srcOff.length);
    if (rc != 0) {
      throw new LZ4Exception("liblz4hip status " + rc + ": " + LZ4HIPJNI.lastError());
    }
  }

  /** outLen[i] &gt; 0: compressed size; 0: destCap[i] too small. */
  public static void compress(ByteBuffer src, long[] srcOff, int[] srcLen, ByteBuffer dest, long[] destOff, int[] destCap, int[] outLen) {
    run(0, 0, src, srcOff, srcLen, dest, destOff, destCap, outLen);
  }

  /** outLen[i] &gt;= 0: decompressed size; &lt; 0: -(input position)-1. */
  public static void decompressSafe(ByteBuffer src, long[] srcOff, int[] srcLen, ByteBuffer dest, long[] destOff, int[] destCap, int[] outLen) {
    run(1, 0, src, srcOff, srcLen, dest, destOff, destCap, outLen);
  }

  /** destLen[i] is the exact decompressed size; outConsumed[i] &gt; 0: bytes read from src. */
  public static void decompressFast(ByteBuffer src, long[] srcOff, int[] srcCap, ByteBuffer dest, long[] destOff, int[] destLen, int[] outConsumed) {
    run(2, 0, src, srcOff, srcCap, dest, destOff, destLen, outConsumed);
  }

  public static void compressHC(int level, ByteBuffer src, long[] srcOff, int[] srcLen, ByteBuffer dest, long[] destOff, int[] destCap, int[] outLen) {
    run(3, level, src, srcOff, srcLen, dest, destOff, destCap, outLen);
  }

  /** Container kinds of {@link #containerBlocks}. */
  public static final int FRAME_BLOCKS = 0, LZ4BLOCK_BLOCKS = 1;

  /**
   * The data blocks of an LZ4 Frame ({@link #FRAME_BLOCKS}: what {@code LZ4FrameOutputStream.writeBlock} emits per block;
   * {@code blockChecksum} adds the XXH32 of each stored payload) or of the LZ4Block container ({@link #LZ4BLOCK_BLOCKS}: what
   * {@code LZ4BlockOutputStream.flushBufferedData} emits) for {@code src[srcOff, srcOff+len)} cut into {@code blockSize} pieces,
   * assembled on the device: compression, raw fallback, headers, payload compaction and checksums happen behind one another
   * there, only the finished bytes come back.  {@code level} 0 = fast, 1..17 = HC.  Returns the bytes written at {@code dest[destOff..)}.
   */
  public static long containerBlocks(int kind, boolean blockChecksum, int level, ByteBuffer src, long srcOff, long len, int blockSize,
                                     ByteBuffer dest, long destOff) {
    if (!src.isDirect() || !dest.isDirect()) {
      throw new IllegalArgumentException("LZ4HIPBatch needs direct ByteBuffers");
    }
    if (dest.isReadOnly()) {
      throw new java.nio.ReadOnlyBufferException();
    }
    if (srcOff < 0 || len < 0 || destOff < 0 || srcOff + len > src.capacity() || destOff > dest.capacity()) {
      throw new ArrayIndexOutOfBoundsException();
    }
    final long r = LZ4HIPJNI.LZ4HIP_containerBlocks(kind, blockChecksum ? 1 : 0, level, src, srcOff, len, blockSize, dest, destOff,
        dest.capacity() - destOff);
    if (r < 0) {
      throw new LZ4Exception("liblz4hip status " + r + ": " + LZ4HIPJNI.lastError());
    }
    return r;
  }

  /** Stop reasons of {@link #containerDecode} ({@code info[2]}; include/lz4hip.h). */
  public static final int CR_END = 0, CR_MORE = 1, CR_TRUNCATED = 2, CR_BLOCK_TOO_BIG = 3, CR_BLOCK_CHECKSUM = 4, CR_DECODE = 5,
      CR_CORRUPT = 6, CR_SLOTS = 7;

  /**
   * Destination bytes a {@link #containerDecode} call over {@code src[srcOff, srcOff+len)} can need: a host-side walk of the size
   * words / headers (frame: {@code maxBlock} per compressed block, the stored size of a raw one; LZ4Block: the headers' original
   * lengths).  {@code blocks[0]} receives the number of whole blocks present (at most {@code nMax}).
   */
  public static long containerDecodeBound(int kind, boolean blockChecksum, ByteBuffer src, long srcOff, long len, int maxBlock, int nMax,
                                          int[] blocks) {
    if (!src.isDirect()) {
      throw new IllegalArgumentException("LZ4HIPBatch needs direct ByteBuffers");
    }
    if (blocks == null) {
      throw new NullPointerException("blocks");
    }
    if (srcOff < 0 || len < 0 || srcOff + len > src.capacity() || blocks.length < 1) {
      throw new ArrayIndexOutOfBoundsException();
    }
    final long r = LZ4HIPJNI.LZ4HIP_containerDecodeBound(kind, blockChecksum ? 1 : 0, src, srcOff, len, maxBlock, nMax, blocks);
    if (r < 0) {
      throw new LZ4Exception("liblz4hip status " + r + ": " + LZ4HIPJNI.lastError());
    }
    return r;
  }

  /**
   * The READ side of the container formats on the device: the data blocks of an LZ4 Frame body ({@link #FRAME_BLOCKS}: {@code src}
   * from the first block's size word on; {@code maxBlock} = the frame's block maximum size) or of an LZ4Block stream
   * ({@link #LZ4BLOCK_BLOCKS}; {@code maxBlock} = an upper bound of the original lengths, {@code 1 << (10 + level nibble)}) are
   * walked, verified and decoded there -- what {@code LZ4FrameInputStream.readBlock} / {@code LZ4BlockInputStream.refill} do block
   * by block, for up to {@code nMax} blocks in one call.  The decoded blocks land back to back at {@code dest[destOff..)},
   * {@code sizes[k]} = decoded size of block k, {@code info} = {blocks delivered, bytes of src consumed, stop reason (CR_*), decoded
   * bytes, liblz4's code of a failed decode}.  The reader maps the stop reason to the reference's exception and goes on from
   * {@code srcOff + info[1]}.
   */
  public static void containerDecode(int kind, boolean blockChecksum, ByteBuffer src, long srcOff, long len, int maxBlock, int nMax,
                                     ByteBuffer dest, long destOff, int[] sizes, long[] info) {
    if (!src.isDirect() || !dest.isDirect()) {
      throw new IllegalArgumentException("LZ4HIPBatch needs direct ByteBuffers");
    }
    if (dest.isReadOnly()) {
      throw new java.nio.ReadOnlyBufferException();
    }
    if (srcOff < 0 || len < 0 || destOff < 0 || srcOff + len > src.capacity() || destOff > dest.capacity() || sizes.length < nMax
        || info.length < 5) {
      throw new ArrayIndexOutOfBoundsException();
    }
    final int r = LZ4HIPJNI.LZ4HIP_containerDecode(kind, blockChecksum ? 1 : 0, src, srcOff, len, maxBlock, nMax, dest, destOff,
        dest.capacity() - destOff, sizes, info);
    if (r != 0) {
      throw new LZ4Exception("liblz4hip status " + r + ": " + LZ4HIPJNI.lastError());
    }
  }
}
