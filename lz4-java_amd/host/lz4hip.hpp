// lz4hip.hpp -- C++ host-side mirror of lz4-java's plugin interface for the "HIP" family, header-only
// over the C ABI (include/lz4hip.h).  The reference's host side is compiled (JVM) code; this image has
// no JDK, so this header is the compiled-language mirror (the Java twins live in ../java):
//
//   net::jpountz::lz4::LZ4Factory::hipInstance()     <- LZ4Factory.java:91-126 (new fourth accessor), :229-291
//   LZ4Compressor::compress / maxCompressedLength   <- LZ4Compressor.java:36,59 ; LZ4JNICompressor.java:35-43
//   LZ4SafeDecompressor::decompress                 <- LZ4SafeDecompressor.java:45 ; LZ4JNISafeDecompressor.java:34-43
//   LZ4FastDecompressor::decompress                 <- LZ4FastDecompressor.java:48 ; LZ4JNIFastDecompressor.java:35-44
//   LZ4Exception                                    <- LZ4Exception.java
//   net::jpountz::xxhash::XXHashFactory::hipInstance().hash32()/hash64()  <- XXHashFactory.java:80,211,220
// Same names, argument meaning (byte array + offset + length) and error behaviour: range violations throw
// std::out_of_range / std::invalid_argument (ArrayIndexOutOfBounds / IllegalArgument, SafeUtils.java:24-42),
// codec failures throw LZ4Exception with the reference's messages.  No CPU path: a missing GPU throws.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/lz4hip.h"

namespace net { namespace jpountz {

using bytes = std::vector<uint8_t>;

namespace util {
inline void checkLength(int len) { if (len < 0) throw std::invalid_argument("lengths must be >= 0"); }
inline void checkRange(const bytes& buf, int off) { if (off < 0 || off >= (int)buf.size()) throw std::out_of_range(std::to_string(off)); }
inline void checkRange(const bytes& buf, int off, int len) {
  checkLength(len);
  if (len > 0) { checkRange(buf, off); checkRange(buf, off + len - 1); }
}
}  // namespace util

namespace lz4 {

struct LZ4Exception : std::runtime_error { using std::runtime_error::runtime_error; };

inline int libCheck(int ret) {
  if (LZ4HIP_IS_LIB_ERROR(ret)) throw LZ4Exception(std::string("liblz4hip: ") + lz4hip_last_error());
  return ret;
}

class LZ4Compressor {
 public:
  virtual ~LZ4Compressor() {}
  int maxCompressedLength(int length) const {  // LZ4Utils.java:34-41
    if (length < 0) throw std::invalid_argument("length must be >= 0, got " + std::to_string(length));
    if (length >= 0x7E000000) throw std::invalid_argument("length must be < 0x7E000000");
    return length + length / 255 + 16;
  }
  virtual int compress(const bytes& src, int srcOff, int srcLen, bytes& dest, int destOff, int maxDestLen) const = 0;
  int compress(const bytes& src, int srcOff, int srcLen, bytes& dest, int destOff) const {
    return compress(src, srcOff, srcLen, dest, destOff, (int)dest.size() - destOff);
  }
  int compress(const bytes& src, bytes& dest) const { return compress(src, 0, (int)src.size(), dest, 0); }
  bytes compress(const bytes& src, int srcOff, int srcLen) const {
    bytes out((size_t)maxCompressedLength(srcLen));
    out.resize((size_t)compress(src, srcOff, srcLen, out, 0, (int)out.size()));
    return out;
  }
  bytes compress(const bytes& src) const { return compress(src, 0, (int)src.size()); }
};

class LZ4HIPCompressor final : public LZ4Compressor {
 public:
  using LZ4Compressor::compress;
  int compress(const bytes& src, int srcOff, int srcLen, bytes& dest, int destOff, int maxDestLen) const override {
    util::checkRange(src, srcOff, srcLen);
    util::checkRange(dest, destOff, maxDestLen);
    const int result = libCheck(lz4hip_compress_fast(src.data() + srcOff, srcLen, dest.data() + destOff, maxDestLen));
    if (result <= 0) throw LZ4Exception("maxDestLen is too small");
    return result;
  }
};

class LZ4HCHIPCompressor final : public LZ4Compressor {
  int level_;
 public:
  using LZ4Compressor::compress;
  explicit LZ4HCHIPCompressor(int compressionLevel = 9) : level_(compressionLevel) {}
  int compress(const bytes& src, int srcOff, int srcLen, bytes& dest, int destOff, int maxDestLen) const override {
    util::checkRange(src, srcOff, srcLen);
    util::checkRange(dest, destOff, maxDestLen);
    const int result = libCheck(lz4hip_compress_hc(src.data() + srcOff, srcLen, dest.data() + destOff, maxDestLen, level_));
    if (result <= 0) throw LZ4Exception("");
    return result;
  }
};

class LZ4SafeDecompressor {
 public:
  int decompress(const bytes& src, int srcOff, int srcLen, bytes& dest, int destOff, int maxDestLen) const {
    util::checkRange(src, srcOff, srcLen);
    util::checkRange(dest, destOff, maxDestLen);
    const int result = libCheck(lz4hip_decompress_safe(src.data() + srcOff, srcLen, dest.data() + destOff, maxDestLen));
    if (result < 0) throw LZ4Exception("Error decoding offset " + std::to_string(srcOff - result) + " of input buffer");
    return result;
  }
  int decompress(const bytes& src, bytes& dest) const { return decompress(src, 0, (int)src.size(), dest, 0, (int)dest.size()); }
  bytes decompress(const bytes& src, int maxDestLen) const {
    bytes out((size_t)maxDestLen);
    out.resize((size_t)decompress(src, 0, (int)src.size(), out, 0, maxDestLen));
    return out;
  }
};

class LZ4FastDecompressor {
 public:
  int decompress(const bytes& src, int srcOff, bytes& dest, int destOff, int destLen) const {
    if (!src.empty() || srcOff) util::checkRange(src, srcOff);
    util::checkRange(dest, destOff, destLen);
    const int result = libCheck(lz4hip_decompress_fast(src.data() + srcOff, (int)src.size() - srcOff, dest.data() + destOff, destLen));
    if (result < 0) throw LZ4Exception("Error decoding offset " + std::to_string(srcOff - result) + " of input buffer");
    return result;
  }
  bytes decompress(const bytes& src, int destLen) const {
    bytes out((size_t)destLen);
    decompress(src, 0, out, 0, destLen);
    return out;
  }
};

class LZ4Factory {
  LZ4HIPCompressor fast_;
  LZ4HCHIPCompressor high_;
  LZ4FastDecompressor fastDec_;
  LZ4SafeDecompressor safeDec_;
  LZ4Factory() {
    // LZ4Factory.java:204-220: round-trip a 20-byte vector through the members before handing them out
    const bytes original = {'a','b','c','d',' ',' ',' ',' ',' ',' ','a','b','c','d','e','f','g','h','i','j'};
    const LZ4Compressor* both[2] = {&fast_, &high_};
    for (const LZ4Compressor* c : both) {
      const bytes compressed = c->compress(original);
      if (fastDec_.decompress(compressed, (int)original.size()) != original) throw std::logic_error("AssertionError");
      if (safeDec_.decompress(compressed, (int)original.size()) != original) throw std::logic_error("AssertionError");
    }
  }
 public:
  static LZ4Factory& hipInstance() { static LZ4Factory f; return f; }
  const LZ4Compressor& fastCompressor() const { return fast_; }
  std::unique_ptr<LZ4Compressor> highCompressor(int compressionLevel = 9) const {  // clamp: LZ4Factory.java:263-270
    if (compressionLevel > 17) compressionLevel = 17; else if (compressionLevel < 1) compressionLevel = 9;
    return std::unique_ptr<LZ4Compressor>(new LZ4HCHIPCompressor(compressionLevel));
  }
  const LZ4FastDecompressor& fastDecompressor() const { return fastDec_; }
  const LZ4SafeDecompressor& safeDecompressor() const { return safeDec_; }
  std::string toString() const { return "LZ4Factory:HIP"; }
};

}  // namespace lz4

namespace xxhash {
struct XXHash32 {
  int32_t hash(const bytes& buf, int off, int len, int32_t seed) const {
    util::checkRange(buf, off, len);
    uint32_t h = 0;
    if (lz4hip_xxh32(buf.data() + off, len, (uint32_t)seed, &h) != 0) throw std::runtime_error(std::string("liblz4hip: ") + lz4hip_last_error());
    return (int32_t)h;
  }
};
struct XXHash64 {
  int64_t hash(const bytes& buf, int off, int len, int64_t seed) const {
    util::checkRange(buf, off, len);
    uint64_t h = 0;
    if (lz4hip_xxh64(buf.data() + off, len, (uint64_t)seed, &h) != 0) throw std::runtime_error(std::string("liblz4hip: ") + lz4hip_last_error());
    return (int64_t)h;
  }
};
// StreamingXXHash32.java:38-129 / StreamingXXHash64.java (JNI twins StreamingXXHash32JNI.java:28-104): the state is a device
// record behind a lz4hip_xxh_stream handle; update() continues it with one launch.
namespace detail {
inline void xchk(int rc) { if (rc != 0) throw std::runtime_error(std::string("liblz4hip: ") + lz4hip_last_error()); }
class StreamingBase {
 public:
  StreamingBase(const StreamingBase&) = delete;
  StreamingBase& operator=(const StreamingBase&) = delete;
  ~StreamingBase() { close(); }
  void update(const bytes& buf, int off, int len) {
    checkState();
    util::checkRange(buf, off, len);
    xchk(lz4hip_xxh_stream_update(st_, buf.data() + off, len));
  }
  void update(const uint8_t* p, size_t n) {  // raw overload for the stream classes (pieces of <= 1 GiB)
    checkState();
    while (n) { const size_t part = n < (1u << 30) ? n : (1u << 30); xchk(lz4hip_xxh_stream_update(st_, p, (int)part)); p += part; n -= part; }
  }
  void close() { if (st_) { lz4hip_xxh_stream_free(st_); st_ = nullptr; } }
 protected:
  StreamingBase() = default;
  void checkState() const { if (!st_) throw std::logic_error("Already finalized"); }  // StreamingXXHash32JNI.java:47-51
  lz4hip_xxh_stream* st_ = nullptr;
};
}  // namespace detail
class StreamingXXHash32 final : public detail::StreamingBase {
 public:
  explicit StreamingXXHash32(int32_t seed) : seed_(seed) { detail::xchk(lz4hip_xxh32_stream_create((uint32_t)seed, &st_)); }
  int32_t getValue() { checkState(); uint32_t h = 0; detail::xchk(lz4hip_xxh32_stream_digest(st_, &h)); return (int32_t)h; }
  void reset() { checkState(); detail::xchk(lz4hip_xxh_stream_reset(st_, (uint32_t)seed_)); }
  int64_t checksumValue() { return (int64_t)((uint32_t)getValue() & 0xFFFFFFFu); }  // asChecksum().getValue(): 28 bits (:101-107)
 private:
  int32_t seed_;
};
class StreamingXXHash64 final : public detail::StreamingBase {
 public:
  explicit StreamingXXHash64(int64_t seed) : seed_(seed) { detail::xchk(lz4hip_xxh64_stream_create((uint64_t)seed, &st_)); }
  int64_t getValue() { checkState(); uint64_t h = 0; detail::xchk(lz4hip_xxh64_stream_digest(st_, &h)); return (int64_t)h; }
  void reset() { checkState(); detail::xchk(lz4hip_xxh_stream_reset(st_, (uint64_t)seed_)); }
 private:
  int64_t seed_;
};
struct XXHashFactory {
  static XXHashFactory& hipInstance() { static XXHashFactory f; return f; }
  XXHash32 hash32() const { return XXHash32(); }
  XXHash64 hash64() const { return XXHash64(); }
  std::unique_ptr<StreamingXXHash32> newStreamingHash32(int32_t seed) const { return std::unique_ptr<StreamingXXHash32>(new StreamingXXHash32(seed)); }  // XXHashFactory.java:230
  std::unique_ptr<StreamingXXHash64> newStreamingHash64(int64_t seed) const { return std::unique_ptr<StreamingXXHash64>(new StreamingXXHash64(seed)); }  // :240
};
}  // namespace xxhash

}}  // namespace net::jpountz
