// lz4hip_streams.hpp -- C++ twins of the reference's stream / container classes over the batch C ABI:
//
//   net::jpountz::lz4::LZ4FrameOutputStream / LZ4FrameInputStream   <- LZ4FrameOutputStream.java, LZ4FrameInputStream.java
//   net::jpountz::lz4::LZ4BlockOutputStream / LZ4BlockInputStream   <- LZ4BlockOutputStream.java, LZ4BlockInputStream.java
//   net::jpountz::lz4::LZ4CompressorWithLength / LZ4DecompressorWithLength  <- the classes of the same name
//
// Same container bytes, flags, checks, error messages and error ORDER as the reference classes; the cadence differs: the
// reference calls compressor.compress / decompressor.decompress once per block (LZ4FrameOutputStream.java:199-235,
// LZ4BlockOutputStream.java:203-227, LZ4FrameInputStream.java:258-322, LZ4BlockInputStream.java:191-264), these queue
// `batchBlocks` independent blocks and hand them to ONE lz4hip_*_batch call (and one lz4hip_xxh32_batch call for the per-block
// checksums).  A defect in block k of a batch surfaces once the bytes of blocks < k have been delivered, as in the reference.
// std::ostream / std::istream stand in for java.io.OutputStream / InputStream.  The Python twin with the same structure is
// lz4-java_amd/streams.py.  No CPU codec anywhere: every data byte goes through liblz4hip.
#pragma once
#include <cstring>
#include <deque>
#include <functional>
#include <istream>
#include <ostream>
#include "lz4hip.hpp"

namespace net { namespace jpountz { namespace lz4 {

struct IOException : std::runtime_error { using std::runtime_error::runtime_error; };
struct EOFException : IOException { using IOException::IOException; };
struct IllegalStateException : std::logic_error { using std::logic_error::logic_error; };

// the batch engine behind the streams: fast or HC compressor, both decompressors, xxh32
struct BatchEngine {
  int hcLevel = 0;  // 0: LZ4_compress_default; 1..9: LZ4_compress_HC at that level
  static void chk(int rc) { if (rc != 0) throw LZ4Exception(std::string("liblz4hip: ") + lz4hip_last_error()); }
  std::vector<int32_t> compress(const uint8_t* src, const std::vector<uint64_t>& so, const std::vector<int32_t>& sl, uint8_t* dst,
                                const std::vector<uint64_t>& d_o, const std::vector<int32_t>& dc) const {
    std::vector<int32_t> out(so.size());
    if (so.empty()) return out;
    if (hcLevel) chk(lz4hip_compress_hc_batch(src, so.data(), sl.data(), dst, d_o.data(), dc.data(), out.data(), (uint32_t)so.size(), hcLevel));
    else chk(lz4hip_compress_fast_batch(src, so.data(), sl.data(), dst, d_o.data(), dc.data(), out.data(), (uint32_t)so.size()));
    return out;
  }
  std::vector<int32_t> decompressSafe(const uint8_t* src, const std::vector<uint64_t>& so, const std::vector<int32_t>& sl, uint8_t* dst,
                                      const std::vector<uint64_t>& d_o, const std::vector<int32_t>& dc) const {
    std::vector<int32_t> out(so.size());
    if (!so.empty()) chk(lz4hip_decompress_safe_batch(src, so.data(), sl.data(), dst, d_o.data(), dc.data(), out.data(), (uint32_t)so.size()));
    return out;
  }
  std::vector<int32_t> decompressFast(const uint8_t* src, const std::vector<uint64_t>& so, const std::vector<int32_t>& scap, uint8_t* dst,
                                      const std::vector<uint64_t>& d_o, const std::vector<int32_t>& dl) const {
    std::vector<int32_t> out(so.size());
    if (!so.empty()) chk(lz4hip_decompress_fast_batch(src, so.data(), scap.data(), dst, d_o.data(), dl.data(), out.data(), (uint32_t)so.size()));
    return out;
  }
  std::vector<uint32_t> xxh32(const uint8_t* buf, const std::vector<uint64_t>& off, const std::vector<int32_t>& len, uint32_t seed) const {
    std::vector<uint32_t> out(off.size());
    if (!off.empty()) chk(lz4hip_xxh32_batch(buf, off.data(), len.data(), seed, out.data(), (uint32_t)off.size()));
    return out;
  }
  // round 3: the container bytes of a batch of blocks laid out on the DEVICE behind the compress launch (lz4hip_container_blocks:
  // raw fallback, size scan, headers, payload compaction, checksums); hostAssembly = true keeps the round-1/2 path (compress batch,
  // then assembly on the host) -- same bytes either way
  bool hostAssembly = false;
  bytes containerBlocks(int kind, const bytes& data, int blockSize, bool blockChecksum) const {
    const size_t n = (data.size() + (size_t)blockSize - 1) / (size_t)blockSize;
    bytes o(data.size() + n * (kind == 0 ? 8u : 21u) + 1u);
    uint64_t got = 0;
    chk(lz4hip_container_blocks(kind, blockChecksum ? 1 : 0, hcLevel, data.data(), data.size(), (uint32_t)blockSize, o.data(), o.size(), &got));
    o.resize((size_t)got);
    return o;
  }
  // rounds 4 / 5: the READ side on the device (lz4hip_container_decode: the size words / LZ4Block headers of a run of blocks walked,
  // checksums verified, blocks decoded there; what LZ4FrameInputStream.readBlock / LZ4BlockInputStream.refill do block by block).
  // hostWalk = true keeps the readers' host walk around the batch launches (rounds 1-3) -- same bytes, same exceptions either way
  bool hostWalk = false;
  enum { CR_END = 0, CR_MORE = 1, CR_TRUNCATED = 2, CR_BLOCK_TOO_BIG = 3, CR_BLOCK_CHECKSUM = 4, CR_DECODE = 5, CR_CORRUPT = 6, CR_SLOTS = 7 };
  struct ContainerRun { bytes decoded; std::vector<int32_t> sizes; uint64_t consumed = 0; int why = CR_MORE; int64_t code = 0; };
  ContainerRun containerDecode(int kind, const uint8_t* body, size_t len, uint32_t maxBlock, uint32_t nMax, bool blockChecksum) const {
    ContainerRun r;
    static const uint8_t dummy = 0;
    uint32_t nb = 0;
    uint64_t need = 0;   // the destination by what the body holds (a host-side walk of the headers), not by nMax x maxBlock
    chk(lz4hip_container_decode_bound(kind, blockChecksum ? 1 : 0, len ? body : &dummy, len, maxBlock, nMax, &nb, &need));
    r.decoded.resize(need ? (size_t)need : 1u);
    r.sizes.assign(nMax, 0);
    uint64_t info[5] = {0, 0, 0, 0, 0};
    chk(lz4hip_container_decode(kind, blockChecksum ? 1 : 0, len ? body : &dummy, len, maxBlock, nMax, r.decoded.data(), r.decoded.size(), r.sizes.data(), info));
    r.sizes.resize((size_t)info[0]);
    r.consumed = info[1]; r.why = (int)info[2]; r.decoded.resize((size_t)info[3]); r.code = (int64_t)info[4];
    return r;
  }
  uint32_t xxh32_one(const uint8_t* buf, size_t n, uint32_t seed) const {
    static const uint8_t dummy = 0;
    return xxh32(n ? buf : &dummy, {0}, {(int32_t)n}, seed)[0];
  }
};

namespace detail {
inline void putLE32(bytes& b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
inline void putLE64(bytes& b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
inline uint32_t getLE32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t getLE64(const uint8_t* p) { return (uint64_t)getLE32(p) | ((uint64_t)getLE32(p + 4) << 32); }
inline int maxCompressedLength(int n) { return n + n / 255 + 16; }
// reads up to n bytes; returns how many arrived
inline size_t readUpTo(std::istream& in, uint8_t* p, size_t n) {
  size_t got = 0;
  while (got < n && in.good()) {
    in.read((char*)p + got, (std::streamsize)(n - got));
    const std::streamsize k = in.gcount();
    if (k <= 0) break;
    got += (size_t)k;
  }
  return got;
}
// the readers' input: an istream plus the bytes the device read path took from it and did not consume (they are read again first)
class PushbackReader {
 public:
  explicit PushbackReader(std::istream& in) : in_(in) {}
  size_t readUpTo(uint8_t* p, size_t n) {
    size_t got = std::min(n, back_.size());
    std::copy(back_.begin(), back_.begin() + (std::ptrdiff_t)got, p);
    back_.erase(back_.begin(), back_.begin() + (std::ptrdiff_t)got);
    if (got < n) got += detail::readUpTo(in_, p + got, n - got);
    return got;
  }
  // up to `want` bytes, in pieces: a reader that asks for a whole batch of maximum-size blocks does not allocate them for a short stream
  bytes readChunk(size_t want) {
    bytes c;
    while (c.size() < want) {
      const size_t piece = std::min<size_t>(want - c.size(), (size_t)4 << 20), at = c.size();
      c.resize(at + piece);
      const size_t got = readUpTo(c.data() + at, piece);
      c.resize(at + got);
      if (got < piece) break;
    }
    return c;
  }
  void unread(const uint8_t* p, size_t n) { back_.insert(back_.begin(), p, p + n); }
  // (a read that reached the end of the stream leaves eofbit AND failbit set: both are cleared for the probe -- a chunk that was cut
  // short by the end of the input is the usual case here, not an error)
  bool seekable() { in_.clear(); return in_.tellg() != std::istream::pos_type(-1); }
  // the container ended and the caller reads on behind it: the surplus goes back to the stream (the reference's readers consume exactly
  // the container: LZ4FrameInputStream.java:258-322, LZ4BlockInputStream.java:191-264).  Only for seekable inputs.
  void giveBack() {
    if (back_.empty()) return;
    in_.clear();
    in_.seekg(-(std::istream::off_type)back_.size(), std::ios::cur);
    back_.clear();
  }
 private:
  std::istream& in_;
  std::deque<uint8_t> back_;
};
// compress data[i*blockSize ..] for all i in one launch
struct Compressed { bytes dst; int bound; std::vector<int32_t> lens, sizes; };
inline Compressed compressBlocks(const BatchEngine& e, const bytes& data, int blockSize) {
  Compressed c;
  const size_t n = (data.size() + (size_t)blockSize - 1) / (size_t)blockSize;
  c.bound = maxCompressedLength(blockSize);
  c.dst.resize(n * (size_t)c.bound);
  std::vector<uint64_t> so(n), d_o(n);
  std::vector<int32_t> dc(n, c.bound);
  c.lens.resize(n);
  for (size_t i = 0; i < n; i++) {
    so[i] = i * (size_t)blockSize;
    c.lens[i] = (int32_t)std::min<size_t>((size_t)blockSize, data.size() - so[i]);
    d_o[i] = i * (size_t)c.bound;
  }
  c.sizes = e.compress(data.data(), so, c.lens, c.dst.data(), d_o, dc);
  for (int32_t s : c.sizes) if (s <= 0) throw LZ4Exception("maxDestLen is too small");  // LZ4JNICompressor.java:39-41
  return c;
}
}  // namespace detail

// =====================================================================================================
// LZ4 Frame format (v1.5.1)
// =====================================================================================================
namespace frame {
enum BLOCKSIZE { SIZE_64KB = 4, SIZE_256KB = 5, SIZE_1MB = 6, SIZE_4MB = 7 };  // LZ4FrameOutputStream.java:65-83
enum Bits { RESERVED_0 = 0, RESERVED_1 = 1, CONTENT_CHECKSUM = 2, CONTENT_SIZE = 3, BLOCK_CHECKSUM = 4, BLOCK_INDEPENDENCE = 5 };
constexpr uint32_t MAGIC = 0x184D2204u, MAGIC_SKIPPABLE_BASE = 0x184D2A50u, INCOMPRESSIBLE_MASK = 0x80000000u;
constexpr const char* PREMATURE_EOS = "Stream ended prematurely";
constexpr const char* NOT_SUPPORTED = "Stream unsupported";
constexpr const char* BLOCK_HASH_MISMATCH = "Block checksum mismatch";
constexpr const char* DESCRIPTOR_HASH_MISMATCH = "Stream frame descriptor corrupted";
constexpr const char* CLOSED_STREAM = "The stream is already closed";

struct FLG {  // LZ4FrameOutputStream.java:296-372
  int version, bits;
  FLG(int v, int b) : version(v), bits(b & 0x3F) {
    if (isEnabled(RESERVED_0)) throw std::runtime_error("Reserved0 field must be 0");
    if (isEnabled(RESERVED_1)) throw std::runtime_error("Reserved1 field must be 0");
    if (!isEnabled(BLOCK_INDEPENDENCE)) throw std::runtime_error("Dependent block stream is unsupported (BLOCK_INDEPENDENCE must be set)");
    if (version != 1) throw std::runtime_error("Version " + std::to_string(version) + " is unsupported");
  }
  static FLG fromByte(uint8_t f) { return FLG((f >> 6) & 3, f & 0x3F); }
  uint8_t toByte() const { return (uint8_t)(bits | ((version & 3) << 6)); }
  bool isEnabled(int bit) const { return (bits >> bit) & 1; }
};
struct BD {  // LZ4FrameOutputStream.java:374-404
  int blockSizeValue;
  explicit BD(int v) : blockSizeValue(v) {
    if (v < 4 || v > 7) throw std::invalid_argument("Block size must be 4-7. Cannot use value of [" + std::to_string(v) + "]");
  }
  static BD fromByte(uint8_t b) { if (b & 0x8F) throw std::runtime_error("Reserved fields must be 0"); return BD((b >> 4) & 7); }
  int getBlockMaximumSize() const { return 1 << (2 * blockSizeValue + 8); }
  uint8_t toByte() const { return (uint8_t)((blockSizeValue & 7) << 4); }
};
}  // namespace frame

class LZ4FrameOutputStream {
 public:
  LZ4FrameOutputStream(std::ostream& out, frame::BLOCKSIZE blockSize = frame::SIZE_4MB, int64_t knownSize = -1,
                       std::initializer_list<frame::Bits> bits = {frame::BLOCK_INDEPENDENCE}, BatchEngine engine = BatchEngine(),
                       size_t batchBlocks = 64)
      : out_(out), e_(engine), flg_(1, mask(bits)), bd_(blockSize), knownSize_(knownSize), batch_(batchBlocks ? batchBlocks : 1) {
    maxBlockSize_ = bd_.getBlockMaximumSize();
    if (flg_.isEnabled(frame::CONTENT_SIZE) && knownSize < 0)
      throw std::invalid_argument("Known size must be greater than zero in order to use the known size feature");
    bytes head;   // LZ4FrameOutputStream.java:172-190
    detail::putLE32(head, frame::MAGIC);
    head.push_back(flg_.toByte());
    head.push_back(bd_.toByte());
    if (flg_.isEnabled(frame::CONTENT_SIZE)) detail::putLE64(head, (uint64_t)knownSize_);
    head.push_back((uint8_t)((e_.xxh32_one(head.data() + 4, head.size() - 4, 0) >> 8) & 0xFF));
    out_.write((const char*)head.data(), (std::streamsize)head.size());
  }
  ~LZ4FrameOutputStream() { try { close(); } catch (...) {} }
  void write(const uint8_t* p, size_t n) {
    if (finished_) throw IllegalStateException(frame::CLOSED_STREAM);
    buf_.insert(buf_.end(), p, p + n);
    if (buf_.size() >= batch_ * (size_t)maxBlockSize_) writeBlocks(buf_.size() / (size_t)maxBlockSize_ * (size_t)maxBlockSize_);
  }
  void write(const bytes& b) { write(b.data(), b.size()); }
  void flush() { if (!finished_) { writeBlocks(buf_.size()); out_.flush(); } }  // :279-286: flush() writes the pending (possibly short) block
  void close() {                                                                  // :289-298
    if (finished_) return;
    writeBlocks(buf_.size());
    bytes tail;
    detail::putLE32(tail, 0);
    if (flg_.isEnabled(frame::CONTENT_CHECKSUM)) detail::putLE32(tail, (uint32_t)content().getValue());
    out_.write((const char*)tail.data(), (std::streamsize)tail.size());
    out_.flush();
    finished_ = true;
  }

 private:
  static int mask(std::initializer_list<frame::Bits> bits) { int m = 0; for (auto b : bits) m |= 1 << b; return m; }
  void writeBlocks(size_t nbytes) {  // writeBlock (:199-235) for every block of buf_[:nbytes] at once
    if (nbytes == 0) return;
    bytes data(buf_.begin(), buf_.begin() + (std::ptrdiff_t)nbytes);
    buf_.erase(buf_.begin(), buf_.begin() + (std::ptrdiff_t)nbytes);
    if (flg_.isEnabled(frame::CONTENT_CHECKSUM)) content().update(data.data(), data.size());  // :211-213
    const bool bc = flg_.isEnabled(frame::BLOCK_CHECKSUM);
    if (!e_.hostAssembly) {
      const bytes blocks = e_.containerBlocks(0, data, maxBlockSize_, bc);
      out_.write((const char*)blocks.data(), (std::streamsize)blocks.size());
      return;
    }
    const detail::Compressed c = detail::compressBlocks(e_, data, maxBlockSize_);
    bytes o;
    std::vector<uint64_t> spanOff;
    std::vector<int32_t> spanLen;
    for (size_t i = 0; i < c.lens.size(); i++) {
      const uint8_t* payload;
      int32_t n;
      if (c.sizes[i] >= c.lens[i]) {  // store uncompressed if compression does not gain
        payload = data.data() + i * (size_t)maxBlockSize_; n = c.lens[i];
        detail::putLE32(o, (uint32_t)n | frame::INCOMPRESSIBLE_MASK);
      } else {
        payload = c.dst.data() + i * (size_t)c.bound; n = c.sizes[i];
        detail::putLE32(o, (uint32_t)n);
      }
      spanOff.push_back(o.size()); spanLen.push_back(n);
      o.insert(o.end(), payload, payload + n);
      if (bc) detail::putLE32(o, 0);
    }
    if (bc) {
      const std::vector<uint32_t> h = e_.xxh32(o.data(), spanOff, spanLen, 0);
      for (size_t i = 0; i < h.size(); i++)
        for (int k = 0; k < 4; k++) o[spanOff[i] + (size_t)spanLen[i] + (size_t)k] = (uint8_t)(h[i] >> (8 * k));
    }
    out_.write((const char*)o.data(), (std::streamsize)o.size());
  }
  std::ostream& out_;
  BatchEngine e_;
  frame::FLG flg_;
  frame::BD bd_;
  int64_t knownSize_;
  size_t batch_;
  int maxBlockSize_ = 0;
  xxhash::StreamingXXHash32& content() { if (!content_) content_.reset(new xxhash::StreamingXXHash32(0)); return *content_; }
  bytes buf_;
  std::unique_ptr<xxhash::StreamingXXHash32> content_;  // LZ4FrameOutputStream.java:116
  bool finished_ = false;
};

class LZ4FrameInputStream {
 public:
  explicit LZ4FrameInputStream(std::istream& in, bool readSingleFrame = false, BatchEngine engine = BatchEngine(), size_t batchBlocks = 64)
      : r_(in), e_(engine), single_(readSingleFrame), batch_(batchBlocks ? batchBlocks : 1) {}
  // up to n bytes; 0 at the end of the stream
  size_t read(uint8_t* p, size_t n) {
    if (n == 0 || !fill()) return 0;
    const size_t k = std::min(n, ready_.size());
    std::copy(ready_.begin(), ready_.begin() + (std::ptrdiff_t)k, p);
    ready_.erase(ready_.begin(), ready_.begin() + (std::ptrdiff_t)k);
    return k;
  }
  bytes readAll() {
    bytes all;
    while (fill()) { all.insert(all.end(), ready_.begin(), ready_.end()); ready_.clear(); }
    return all;
  }
  // InputStream.skip / available: at most what is decoded and waiting (one batch of blocks); mark/reset are not supported
  int64_t skip(int64_t n) {
    if (n <= 0 || !fill()) return 0;
    const size_t k = std::min((size_t)n, ready_.size());
    ready_.erase(ready_.begin(), ready_.begin() + (std::ptrdiff_t)k);
    return (int64_t)k;
  }
  size_t available() const { return ready_.size(); }
  static bool markSupported() { return false; }
  int64_t getExpectedContentSize() { if (!headerRead_) fill(); return expectedContentSize_; }
  bool isExpectedContentSizeDefined() { return getExpectedContentSize() >= 0; }

 private:
  void readFully(uint8_t* p, size_t n) { if (r_.readUpTo(p, n) < n) throw IOException(frame::PREMATURE_EOS); }
  bool nextFrameInfo() {  // LZ4FrameInputStream.java:124-160
    for (;;) {
      uint8_t h[4];
      const size_t got = r_.readUpTo(h, 4);
      if (got == 0 && headerRead_) return false;  // clean end between frames
      if (got < 4) throw IOException(frame::PREMATURE_EOS);
      const uint32_t magic = detail::getLE32(h);
      if (magic == frame::MAGIC) { readHeader(); return true; }
      if ((magic >> 4) == (frame::MAGIC_SKIPPABLE_BASE >> 4)) {
        uint8_t s[4];
        readFully(s, 4);
        bytes skip(detail::getLE32(s));
        readFully(skip.data(), skip.size());
        headerRead_ = true;
      } else {
        throw IOException(frame::NOT_SUPPORTED);
      }
    }
  }
  void readHeader() {  // :180-224
    bytes header(2);
    readFully(header.data(), 2);
    flgBits_ = frame::FLG::fromByte(header[0]).bits;
    maxBlockSize_ = frame::BD::fromByte(header[1]).getBlockMaximumSize();
    if ((flgBits_ >> frame::CONTENT_SIZE) & 1) {
      uint8_t cs[8];
      readFully(cs, 8);
      expectedContentSize_ = (int64_t)detail::getLE64(cs);
      header.insert(header.end(), cs, cs + 8);
    }
    totalContentSize_ = 0;
    const uint8_t h = (uint8_t)((e_.xxh32_one(header.data(), header.size(), 0) >> 8) & 0xFF);
    uint8_t expected;
    readFully(&expected, 1);
    if (h != expected) throw IOException(frame::DESCRIPTOR_HASH_MISMATCH);
    content_.reset();
    headerRead_ = true;
    frameFinished_ = false;
    inFrame_ = true;
  }
  bool bit(int b) const { return (flgBits_ >> b) & 1; }
  // behind the end mark (LZ4FrameInputStream.java:264-276): content checksum, content size
  void endMark() {
    try {
      if (bit(frame::CONTENT_CHECKSUM)) {
        uint8_t w4[4];
        readFully(w4, 4);
        if (detail::getLE32(w4) != (uint32_t)content().getValue()) throw IOException("Content checksum mismatch");
      }
      if (bit(frame::CONTENT_SIZE) && expectedContentSize_ != totalContentSize_) throw IOException("Size check mismatch");
    } catch (const IOException& e) { pending_ = e.what(); return; }
    frameFinished_ = true;
  }
  // readBlock for a run of blocks ON THE DEVICE (BatchEngine::containerDecode): one chunk of container bytes goes over, the decoded
  // blocks and a stop reason come back; what the device did not consume is read again.  Same checks in the same order, same messages;
  // a defect in block k surfaces after the bytes of the blocks before it
  void readBlocksDevice() {
    const bool bc = bit(frame::BLOCK_CHECKSUM);
    const size_t want = batch_ * ((size_t)maxBlockSize_ + 8u) + 4u;
    const bytes chunk = r_.readChunk(want);
    const bool atEof = chunk.size() < want;
    const BatchEngine::ContainerRun run = e_.containerDecode(0, chunk.data(), chunk.size(), (uint32_t)maxBlockSize_, (uint32_t)batch_, bc);
    const uint8_t* rest = chunk.data() + run.consumed;
    const size_t nrest = chunk.size() - (size_t)run.consumed;
    if (!run.decoded.empty()) {
      ready_.insert(ready_.end(), run.decoded.begin(), run.decoded.end());
      totalContentSize_ += (int64_t)run.decoded.size();
      if (bit(frame::CONTENT_CHECKSUM)) content().update(run.decoded.data(), run.decoded.size());  // one update per batch
    }
    r_.unread(rest, nrest);
    uint8_t w4[4];
    switch (run.why) {
      case BatchEngine::CR_END: endMark(); break;
      case BatchEngine::CR_BLOCK_TOO_BIG:
        (void)r_.readUpTo(w4, 4);
        pending_ = "Block size " + std::to_string(detail::getLE32(rest) & ~frame::INCOMPRESSIBLE_MASK) + " exceeded max: " + std::to_string(maxBlockSize_);
        break;
      case BatchEngine::CR_BLOCK_CHECKSUM: pending_ = frame::BLOCK_HASH_MISMATCH; break;
      case BatchEngine::CR_DECODE: pending_ = "Error decoding offset " + std::to_string(-run.code) + " of input buffer"; break;
      case BatchEngine::CR_TRUNCATED:
      case BatchEngine::CR_MORE:
        if (atEof) { bytes drop(nrest); (void)r_.readUpTo(drop.data(), nrest); pending_ = frame::PREMATURE_EOS; }
        break;   // (else: the next call goes on from the bytes handed back)
      default: break;
    }
  }
  void readBlocks() {  // readBlock (:258-322) for up to batch_ blocks
    // (the device path takes a whole chunk from the stream: with readSingleFrame the caller reads on behind the frame, so it needs a
    // stream that can take the surplus back; any other gets the host walk, which consumes exactly the frame)
    if (!e_.hostWalk && (!single_ || r_.seekable())) {
      readBlocksDevice();
      if (single_ && frameFinished_) r_.giveBack();
      return;
    }
    struct Blk { bool compressed; bytes payload; uint32_t stored; };
    std::vector<Blk> blocks;
    bool endMark = false;
    std::string exc;
    const bool bc = bit(frame::BLOCK_CHECKSUM);
    try {
      while (blocks.size() < batch_) {
        uint8_t w4[4];
        readFully(w4, 4);
        const uint32_t word = detail::getLE32(w4);
        const bool compressed = (word & frame::INCOMPRESSIBLE_MASK) == 0;
        const uint32_t size = word & ~frame::INCOMPRESSIBLE_MASK;
        if (size == 0) { endMark = true; break; }
        if (size > (uint32_t)maxBlockSize_)
          throw IOException("Block size " + std::to_string(size) + " exceeded max: " + std::to_string(maxBlockSize_));
        Blk b{compressed, bytes(size), 0};
        readFully(b.payload.data(), size);
        if (bc) { readFully(w4, 4); b.stored = detail::getLE32(w4); }
        blocks.push_back(std::move(b));
      }
    } catch (const IOException& e) { exc = e.what(); }
    const size_t n = blocks.size();
    if (n) {
      bytes src;
      std::vector<uint64_t> offs(n);
      std::vector<int32_t> lens(n);
      for (size_t i = 0; i < n; i++) { offs[i] = src.size(); lens[i] = (int32_t)blocks[i].payload.size(); src.insert(src.end(), blocks[i].payload.begin(), blocks[i].payload.end()); }
      size_t bad = n;
      std::string badExc;
      if (bc) {
        const std::vector<uint32_t> h = e_.xxh32(src.data(), offs, lens, 0);
        for (size_t i = 0; i < n; i++) if (h[i] != blocks[i].stored) { bad = i; badExc = frame::BLOCK_HASH_MISMATCH; break; }
      }
      std::vector<size_t> cidx;
      for (size_t i = 0; i < bad; i++) if (blocks[i].compressed) cidx.push_back(i);
      std::vector<bytes> outs(bad);
      if (!cidx.empty()) {
        bytes dst(cidx.size() * (size_t)maxBlockSize_);
        std::vector<uint64_t> so(cidx.size()), d_o(cidx.size());
        std::vector<int32_t> sl(cidx.size()), dc(cidx.size(), maxBlockSize_);
        for (size_t k = 0; k < cidx.size(); k++) { so[k] = offs[cidx[k]]; sl[k] = lens[cidx[k]]; d_o[k] = k * (size_t)maxBlockSize_; }
        const std::vector<int32_t> res = e_.decompressSafe(src.data(), so, sl, dst.data(), d_o, dc);
        for (size_t k = 0; k < cidx.size(); k++) {
          if (res[k] < 0) {  // LZ4JNISafeDecompressor.java:39-41, wrapped in IOException (:307-311)
            bad = cidx[k]; badExc = "Error decoding offset " + std::to_string(-res[k]) + " of input buffer";
            break;
          }
          outs[cidx[k]].assign(dst.begin() + (std::ptrdiff_t)d_o[k], dst.begin() + (std::ptrdiff_t)d_o[k] + res[k]);
        }
      }
      bytes fresh;
      for (size_t i = 0; i < bad; i++) {
        const bytes& piece = blocks[i].compressed ? outs[i] : blocks[i].payload;
        fresh.insert(fresh.end(), piece.begin(), piece.end());
      }
      ready_.insert(ready_.end(), fresh.begin(), fresh.end());
      totalContentSize_ += (int64_t)fresh.size();
      if (bit(frame::CONTENT_CHECKSUM) && !fresh.empty()) content().update(fresh.data(), fresh.size());  // one update per batch
      if (!badExc.empty()) { pending_ = badExc; return; }
    }
    if (!exc.empty()) { pending_ = exc; return; }
    if (endMark) this->endMark();
  }
  bool fill() {  // false at the end of the stream
    while (ready_.empty()) {
      if (!pending_.empty()) { const std::string m = pending_; pending_.clear(); throw IOException(m); }
      if (!headerRead_ || frameFinished_) {
        if (headerRead_ && frameFinished_ && single_ && inFrame_) return false;
        if (!nextFrameInfo()) return false;
        if (frameFinished_) continue;  // only skippable frames so far
      }
      readBlocks();
    }
    return true;
  }
  detail::PushbackReader r_;
  BatchEngine e_;
  bool single_;
  size_t batch_;
  bool headerRead_ = false, frameFinished_ = true, inFrame_ = false;
  int flgBits_ = 0, maxBlockSize_ = 0;
  int64_t expectedContentSize_ = -1, totalContentSize_ = 0;
  std::deque<uint8_t> ready_;
  xxhash::StreamingXXHash32& content() { if (!content_) content_.reset(new xxhash::StreamingXXHash32(0)); return *content_; }
  std::unique_ptr<xxhash::StreamingXXHash32> content_;
  std::string pending_;
};

// =====================================================================================================
// lz4-java "LZ4Block" container
// =====================================================================================================
namespace blockstream {
constexpr int MAGIC_LENGTH = 8, HEADER_LENGTH = MAGIC_LENGTH + 1 + 4 + 4 + 4;  // LZ4BlockOutputStream.java:42-47
constexpr int COMPRESSION_LEVEL_BASE = 10, MIN_BLOCK_SIZE = 64, MAX_BLOCK_SIZE = 1 << (COMPRESSION_LEVEL_BASE + 0x0F);
constexpr int COMPRESSION_METHOD_RAW = 0x10, COMPRESSION_METHOD_LZ4 = 0x20;
constexpr uint32_t DEFAULT_SEED = 0x9747B28Cu, CHECK_MASK = 0x0FFFFFFFu;  // StreamingXXHash32.asChecksum keeps 28 bits
// a caller-supplied java.util.zip.Checksum (the 5-argument constructor, LZ4BlockOutputStream.java:96-101): the value of one block,
// i.e. reset(); update(block); (int) getValue().  Empty = the default XXH32 above, batched on the device.
using Checksum = std::function<uint32_t(const uint8_t*, size_t)>;
inline const uint8_t* magic() { static const uint8_t m[8] = {'L', 'Z', '4', 'B', 'l', 'o', 'c', 'k'}; return m; }
inline int compressionLevel(int blockSize) {  // LZ4BlockOutputStream.java:57-69
  if (blockSize < MIN_BLOCK_SIZE) throw std::invalid_argument("blockSize must be >= " + std::to_string(MIN_BLOCK_SIZE) + ", got " + std::to_string(blockSize));
  if (blockSize > MAX_BLOCK_SIZE) throw std::invalid_argument("blockSize must be <= " + std::to_string(MAX_BLOCK_SIZE) + ", got " + std::to_string(blockSize));
  int level = 0;
  while ((1 << level) < blockSize) level++;  // ceil(log2)
  return std::max(0, level - COMPRESSION_LEVEL_BASE);
}
}  // namespace blockstream

class LZ4BlockOutputStream {
 public:
  LZ4BlockOutputStream(std::ostream& out, int blockSize = 1 << 16, BatchEngine engine = BatchEngine(), bool syncFlush = false, size_t batchBlocks = 256,
                       blockstream::Checksum checksum = blockstream::Checksum())
      : out_(out), e_(engine), blockSize_(blockSize), level_(blockstream::compressionLevel(blockSize)), syncFlush_(syncFlush),
        batch_(batchBlocks ? batchBlocks : 1), checksum_(std::move(checksum)) {}
  ~LZ4BlockOutputStream() { try { close(); } catch (...) {} }
  void write(const uint8_t* p, size_t n) {
    if (finished_) throw IllegalStateException("This stream is already closed");
    buf_.insert(buf_.end(), p, p + n);
    if (buf_.size() >= batch_ * (size_t)blockSize_) flushBlocks(buf_.size() / (size_t)blockSize_ * (size_t)blockSize_);
  }
  void write(const bytes& b) { write(b.data(), b.size()); }
  void flush() { if (syncFlush_ && !finished_) flushBlocks(buf_.size()); out_.flush(); }  // :241-248
  void finish() {                                                                           // :256-268
    if (finished_) throw IllegalStateException("This stream is already closed");
    flushBlocks(buf_.size());
    bytes end(blockstream::magic(), blockstream::magic() + 8);
    end.push_back((uint8_t)(blockstream::COMPRESSION_METHOD_RAW | level_));
    end.resize((size_t)blockstream::HEADER_LENGTH, 0);
    out_.write((const char*)end.data(), (std::streamsize)end.size());
    finished_ = true;
    out_.flush();
  }
  void close() { if (!finished_) finish(); }

 private:
  void flushBlocks(size_t nbytes) {  // flushBufferedData (:203-227) for every block of buf_[:nbytes] at once
    if (nbytes == 0) return;
    bytes data(buf_.begin(), buf_.begin() + (std::ptrdiff_t)nbytes);
    buf_.erase(buf_.begin(), buf_.begin() + (std::ptrdiff_t)nbytes);
    if (!checksum_ && !e_.hostAssembly) {
      const bytes blocks = e_.containerBlocks(1, data, blockSize_, false);
      out_.write((const char*)blocks.data(), (std::streamsize)blocks.size());
      return;
    }
    const detail::Compressed c = detail::compressBlocks(e_, data, blockSize_);
    std::vector<uint64_t> so(c.lens.size());
    for (size_t i = 0; i < so.size(); i++) so[i] = i * (size_t)blockSize_;
    std::vector<uint32_t> checks;
    if (!checksum_) {
      checks = e_.xxh32(data.data(), so, c.lens, blockstream::DEFAULT_SEED);
      for (auto& v : checks) v &= blockstream::CHECK_MASK;
    } else {
      for (size_t i = 0; i < so.size(); i++) checks.push_back(checksum_(data.data() + so[i], (size_t)c.lens[i]));
    }
    bytes o;
    for (size_t i = 0; i < c.lens.size(); i++) {
      const bool raw = c.sizes[i] >= c.lens[i];
      const int32_t clen = raw ? c.lens[i] : c.sizes[i];
      const uint8_t* payload = raw ? data.data() + i * (size_t)blockSize_ : c.dst.data() + i * (size_t)c.bound;
      o.insert(o.end(), blockstream::magic(), blockstream::magic() + 8);
      o.push_back((uint8_t)((raw ? blockstream::COMPRESSION_METHOD_RAW : blockstream::COMPRESSION_METHOD_LZ4) | level_));
      detail::putLE32(o, (uint32_t)clen);
      detail::putLE32(o, (uint32_t)c.lens[i]);
      detail::putLE32(o, checks[i]);
      o.insert(o.end(), payload, payload + clen);
    }
    out_.write((const char*)o.data(), (std::streamsize)o.size());
  }
  std::ostream& out_;
  BatchEngine e_;
  int blockSize_, level_;
  bool syncFlush_;
  size_t batch_;
  blockstream::Checksum checksum_;
  bytes buf_;
  bool finished_ = false;
};

class LZ4BlockInputStream {
 public:
  explicit LZ4BlockInputStream(std::istream& in, bool stopOnEmptyBlock = true, BatchEngine engine = BatchEngine(), size_t batchBlocks = 256,
                               blockstream::Checksum checksum = blockstream::Checksum())
      : r_(in), e_(engine), stopOnEmpty_(stopOnEmptyBlock), batch_(batchBlocks ? batchBlocks : 1), checksum_(std::move(checksum)) {}
  size_t read(uint8_t* p, size_t n) {
    if (n == 0 || !fill()) return 0;
    const size_t k = std::min(n, ready_.size());
    std::copy(ready_.begin(), ready_.begin() + (std::ptrdiff_t)k, p);
    ready_.erase(ready_.begin(), ready_.begin() + (std::ptrdiff_t)k);
    return k;
  }
  bytes readAll() {
    bytes all;
    while (fill()) { all.insert(all.end(), ready_.begin(), ready_.end()); ready_.clear(); }
    return all;
  }
  // InputStream.skip / available: at most what is decoded and waiting (one batch of blocks); mark/reset are not supported
  int64_t skip(int64_t n) {
    if (n <= 0 || !fill()) return 0;
    const size_t k = std::min((size_t)n, ready_.size());
    ready_.erase(ready_.begin(), ready_.begin() + (std::ptrdiff_t)k);
    return (int64_t)k;
  }
  size_t available() const { return ready_.size(); }
  static bool markSupported() { return false; }

 private:
  static constexpr const char* CORRUPTED = "Stream is corrupted";
  // refill for a run of blocks ON THE DEVICE (BatchEngine::containerDecode): headers walked with the reference's rules, LZ4 blocks
  // through the fast decoder (its return value must be the header's compressed length), raw ones copied, the checksums of the decoded
  // bytes compared -- every defect is "Stream is corrupted", delivered after the blocks before it.  false: this run is for the host
  // path (the stream does not start with a header; a block bigger than the first header announced; a header whose compressedLen
  // exceeds the chunk although more input follows -- asking the device again with the same bytes would never end)
  bool refillDevice() {
    uint8_t head[blockstream::HEADER_LENGTH];
    const size_t hgot = r_.readUpTo(head, sizeof head);
    r_.unread(head, hgot);
    if (hgot < sizeof head || memcmp(head, blockstream::magic(), 8) != 0) return false;
    const uint32_t maxBlock = 1u << (blockstream::COMPRESSION_LEVEL_BASE + (head[8] & 0x0F));
    const size_t want = batch_ * ((size_t)maxBlock + sizeof head) + sizeof head;
    const bytes chunk = r_.readChunk(want);
    const bool atEof = chunk.size() < want;
    const BatchEngine::ContainerRun run = e_.containerDecode(1, chunk.data(), chunk.size(), maxBlock, (uint32_t)batch_, false);
    if (run.why == BatchEngine::CR_BLOCK_TOO_BIG ||
        (run.consumed == 0 && run.decoded.empty() && (run.why == BatchEngine::CR_TRUNCATED || run.why == BatchEngine::CR_MORE) && !atEof)) {
      r_.unread(chunk.data(), chunk.size());
      return false;
    }
    const size_t nrest = chunk.size() - (size_t)run.consumed;
    ready_.insert(ready_.end(), run.decoded.begin(), run.decoded.end());
    r_.unread(chunk.data() + run.consumed, nrest);
    if (run.why == BatchEngine::CR_END) { if (stopOnEmpty_) finished_ = true; }
    else if (run.why == BatchEngine::CR_CORRUPT) { pending_ = CORRUPTED; pendingEof_ = false; }
    else if ((run.why == BatchEngine::CR_TRUNCATED || run.why == BatchEngine::CR_MORE) && atEof) {
      bytes drop(nrest);
      (void)r_.readUpTo(drop.data(), nrest);
      if (nrest < sizeof head && !stopOnEmpty_) finished_ = true;   // (:192-199: the stream may end at -- or inside -- a header)
      else { pending_ = "Stream ended prematurely"; pendingEof_ = true; }
    }
    return true;
  }
  void refill() {  // LZ4BlockInputStream.java:191-264, for up to batch_ blocks
    // (stopOnEmptyBlock: the caller reads on behind the empty block, so the device path -- which takes whole chunks from the stream --
    // needs one that can take the surplus back)
    if (!checksum_ && !e_.hostWalk && (!stopOnEmpty_ || r_.seekable()) && refillDevice()) {
      if (stopOnEmpty_ && finished_) r_.giveBack();
      return;
    }
    struct Blk { int method; bytes payload; int32_t olen; uint32_t check; };
    std::vector<Blk> blocks;
    std::string exc;
    bool eof = false, fin = false;
    while (blocks.size() < batch_) {
      uint8_t head[blockstream::HEADER_LENGTH];
      if (r_.readUpTo(head, sizeof head) < sizeof head) {
        if (!stopOnEmpty_) fin = true; else { exc = "Stream ended prematurely"; eof = true; }
        break;
      }
      try {
        if (memcmp(head, blockstream::magic(), 8) != 0) throw IOException(CORRUPTED);
        const int token = head[8], method = token & 0xF0, level = blockstream::COMPRESSION_LEVEL_BASE + (token & 0x0F);
        if (method != blockstream::COMPRESSION_METHOD_RAW && method != blockstream::COMPRESSION_METHOD_LZ4) throw IOException(CORRUPTED);
        const int32_t clen = (int32_t)detail::getLE32(head + 9), olen = (int32_t)detail::getLE32(head + 13);
        const uint32_t check = detail::getLE32(head + 17);
        if (olen > (1 << level) || olen < 0 || clen < 0 || (olen == 0 && clen != 0) || (olen != 0 && clen == 0) ||
            (method == blockstream::COMPRESSION_METHOD_RAW && olen != clen))
          throw IOException(CORRUPTED);
        if (olen == 0 && clen == 0) {
          if (check != 0) throw IOException(CORRUPTED);
          if (!stopOnEmpty_) continue;
          fin = true;
          break;
        }
        Blk b{method, r_.readChunk((size_t)clen), olen, check};   // (in pieces: a damaged compressedLen of 2 GB allocates what the stream holds, not 2 GB)
        if (b.payload.size() < (size_t)clen) { exc = "Stream ended prematurely"; eof = true; break; }
        blocks.push_back(std::move(b));
      } catch (const IOException& e) { exc = e.what(); break; }
    }
    const size_t n = blocks.size();
    if (n) {
      size_t bad = n;
      std::string badExc;
      std::vector<bytes> raw(n);
      std::vector<size_t> lz;
      for (size_t i = 0; i < n; i++) if (blocks[i].method == blockstream::COMPRESSION_METHOD_LZ4) lz.push_back(i);
      if (!lz.empty()) {
        bytes src;
        std::vector<uint64_t> so(lz.size()), d_o(lz.size());
        std::vector<int32_t> scap(lz.size()), dl(lz.size());
        size_t dtot = 0;
        for (size_t k = 0; k < lz.size(); k++) {
          so[k] = src.size(); scap[k] = (int32_t)blocks[lz[k]].payload.size();
          src.insert(src.end(), blocks[lz[k]].payload.begin(), blocks[lz[k]].payload.end());
          d_o[k] = dtot; dl[k] = blocks[lz[k]].olen; dtot += (size_t)dl[k];
        }
        bytes dst(dtot ? dtot : 1);
        const std::vector<int32_t> res = e_.decompressFast(src.data(), so, scap, dst.data(), d_o, dl);
        for (size_t k = 0; k < lz.size(); k++) {
          if (res[k] < 0 || res[k] != scap[k]) { bad = lz[k]; badExc = CORRUPTED; break; }  // :246-253
          raw[lz[k]].assign(dst.begin() + (std::ptrdiff_t)d_o[k], dst.begin() + (std::ptrdiff_t)d_o[k] + dl[k]);
        }
      }
      for (size_t i = 0; i < bad; i++) if (blocks[i].method != blockstream::COMPRESSION_METHOD_LZ4) raw[i] = blocks[i].payload;
      if (bad) {
        bytes all;
        std::vector<uint64_t> offs(bad);
        std::vector<int32_t> lens(bad);
        for (size_t i = 0; i < bad; i++) { offs[i] = all.size(); lens[i] = (int32_t)raw[i].size(); all.insert(all.end(), raw[i].begin(), raw[i].end()); }
        std::vector<uint32_t> h;
        if (!checksum_) {
          h = e_.xxh32(all.data(), offs, lens, blockstream::DEFAULT_SEED);
          for (auto& v : h) v &= blockstream::CHECK_MASK;
        } else {
          for (size_t i = 0; i < bad; i++) h.push_back(checksum_(raw[i].data(), raw[i].size()));
        }
        for (size_t i = 0; i < bad; i++) if (h[i] != blocks[i].check) { bad = i; badExc = CORRUPTED; break; }
      }
      for (size_t i = 0; i < bad; i++) ready_.insert(ready_.end(), raw[i].begin(), raw[i].end());
      if (!badExc.empty()) { pending_ = badExc; pendingEof_ = false; return; }
    }
    if (!exc.empty()) { pending_ = exc; pendingEof_ = eof; }
    else if (fin) finished_ = true;
  }
  bool fill() {
    while (ready_.empty()) {
      if (!pending_.empty()) {
        const std::string m = pending_;
        pending_.clear();
        if (pendingEof_) throw EOFException(m);
        throw IOException(m);
      }
      if (finished_) return false;
      refill();
    }
    return true;
  }
  detail::PushbackReader r_;
  BatchEngine e_;
  bool stopOnEmpty_;
  size_t batch_;
  blockstream::Checksum checksum_;
  std::deque<uint8_t> ready_;
  std::string pending_;
  bool pendingEof_ = false, finished_ = false;
};

// =====================================================================================================
// length-prefixed blocks
// =====================================================================================================
class LZ4CompressorWithLength {  // LZ4CompressorWithLength.java: 4-byte little-endian decompressed length, then the block
 public:
  explicit LZ4CompressorWithLength(BatchEngine engine = BatchEngine()) : e_(engine) {}
  int maxCompressedLength(int length) const { return detail::maxCompressedLength(length) + 4; }
  bytes compress(const bytes& src) const { return compressMany({src})[0]; }
  std::vector<bytes> compressMany(const std::vector<bytes>& bufs) const {  // one launch for the whole list
    bytes src, dst;
    std::vector<uint64_t> so(bufs.size()), d_o(bufs.size());
    std::vector<int32_t> sl(bufs.size()), dc(bufs.size());
    size_t dtot = 0;
    for (size_t i = 0; i < bufs.size(); i++) {
      so[i] = src.size(); sl[i] = (int32_t)bufs[i].size(); src.insert(src.end(), bufs[i].begin(), bufs[i].end());
      d_o[i] = dtot; dc[i] = detail::maxCompressedLength(sl[i]); dtot += (size_t)dc[i];
    }
    dst.resize(dtot ? dtot : 1);
    static const uint8_t dummy = 0;
    const std::vector<int32_t> sizes = e_.compress(src.empty() ? &dummy : src.data(), so, sl, dst.data(), d_o, dc);
    std::vector<bytes> out;
    for (size_t i = 0; i < bufs.size(); i++) {
      if (sizes[i] <= 0) throw LZ4Exception("maxDestLen is too small");
      bytes o;
      detail::putLE32(o, (uint32_t)sl[i]);
      o.insert(o.end(), dst.begin() + (std::ptrdiff_t)d_o[i], dst.begin() + (std::ptrdiff_t)d_o[i] + sizes[i]);
      out.push_back(std::move(o));
    }
    return out;
  }

 private:
  BatchEngine e_;
};

class LZ4DecompressorWithLength {  // LZ4DecompressorWithLength.java; fast = the LZ4FastDecompressor constructor (:84-87), else safe (:94-97)
 public:
  explicit LZ4DecompressorWithLength(bool fast = true, BatchEngine engine = BatchEngine()) : fast_(fast), e_(engine) {}
  static int getDecompressedLength(const bytes& src, int srcOff = 0) { return (int32_t)detail::getLE32(src.data() + srcOff); }
  bytes decompress(const bytes& src) const { return decompressMany({src})[0]; }
  std::vector<bytes> decompressMany(const std::vector<bytes>& bufs) const {
    bytes src;
    std::vector<uint64_t> so(bufs.size()), d_o(bufs.size());
    std::vector<int32_t> sl(bufs.size()), dl(bufs.size());
    size_t dtot = 0;
    for (size_t i = 0; i < bufs.size(); i++) {
      dl[i] = getDecompressedLength(bufs[i]);
      if (dl[i] < 0) throw std::invalid_argument("lengths must be >= 0");
      so[i] = src.size() + 4; sl[i] = (int32_t)bufs[i].size() - 4; src.insert(src.end(), bufs[i].begin(), bufs[i].end());
      d_o[i] = dtot; dtot += (size_t)dl[i];
    }
    bytes dst(dtot ? dtot : 1);
    const std::vector<int32_t> res = fast_ ? e_.decompressFast(src.data(), so, sl, dst.data(), d_o, dl) : e_.decompressSafe(src.data(), so, sl, dst.data(), d_o, dl);
    std::vector<bytes> out;
    for (size_t i = 0; i < bufs.size(); i++) {
      if (res[i] < 0) throw LZ4Exception("Error decoding offset " + std::to_string(4 - res[i]) + " of input buffer");
      out.emplace_back(dst.begin() + (std::ptrdiff_t)d_o[i], dst.begin() + (std::ptrdiff_t)d_o[i] + (fast_ ? dl[i] : res[i]));
    }
    return out;
  }

 private:
  bool fast_;
  BatchEngine e_;
};

}}}  // namespace net::jpountz::lz4
