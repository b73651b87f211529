// tests/cpp/stream_mirror_test.cpp -- exercises the C++ twins of the reference's stream / container classes
// (lz4-java_amd/host/lz4hip_streams.hpp) the way LZ4FrameIOStreamTest.java / LZ4BlockStreamingTest.java do.
// Built by the CPU suite (compile + link check), run by the GPU suite:
//   stream_mirror_test <dir>    writes <dir>/cpp_*.lz4 / *.blk (the pytest side decodes them with the lz4 CLI and the Python
//                               twin and compares bytes) and reads <dir>/cli.lz4 if it exists (a frame made by the lz4 CLI)
// Exit code 0 = all good; with no GPU it must fail loudly (exit code 3).
#include <cstdio>
#include <fstream>
#include <sstream>
#include "../../lz4-java_amd/host/lz4hip_streams.hpp"

using namespace net::jpountz;
using namespace net::jpountz::lz4;

static bytes payload() {
  bytes d;
  uint32_t s = 12345;
  for (int rep = 0; rep < 40; rep++) {
    const char* t = "It was the best of times, it was the worst of times, it was the age of wisdom, it was the age of foolishness; ";
    for (int k = 0; k < 300; k++) d.insert(d.end(), t, t + strlen(t) - (size_t)(k % 7));
    for (int k = 0; k < 9000; k++) { s = s * 1664525u + 1013904223u; d.push_back((uint8_t)(s >> 24)); }   // incompressible stretch
    d.insert(d.end(), 20000, (uint8_t)rep);
  }
  d.resize(d.size() - 4321);
  return d;
}
static bytes slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); const std::string s = ss.str(); return bytes(s.begin(), s.end()); }
static void dump(const std::string& p, const std::string& s) { std::ofstream f(p, std::ios::binary); f.write(s.data(), (std::streamsize)s.size()); }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); return 1; } } while (0)

template <class F> static std::string thrown(F f) { try { f(); } catch (const std::exception& e) { return e.what(); } return ""; }

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  try {
    const bytes data = payload();
    dump(dir + "/cpp_payload.bin", std::string(data.begin(), data.end()));
    // ---- frame: default flags, and every optional field; written in odd-sized pieces with a flush in between ----
    struct Case { const char* name; frame::BLOCKSIZE bs; std::initializer_list<frame::Bits> bits; bool known; };
    const Case cases[] = {
        {"cpp_frame_default.lz4", frame::SIZE_4MB, {frame::BLOCK_INDEPENDENCE}, false},
        {"cpp_frame_all.lz4", frame::SIZE_64KB, {frame::BLOCK_INDEPENDENCE, frame::BLOCK_CHECKSUM, frame::CONTENT_CHECKSUM, frame::CONTENT_SIZE}, true},
        {"cpp_frame_256k_cc.lz4", frame::SIZE_256KB, {frame::BLOCK_INDEPENDENCE, frame::CONTENT_CHECKSUM}, false},
    };
    std::string all_frame;
    for (const Case& c : cases) {
      std::ostringstream sink;
      {
        LZ4FrameOutputStream f(sink, c.bs, c.known ? (int64_t)data.size() : -1, c.bits, BatchEngine(), 3);
        for (size_t i = 0; i < data.size(); i += 100003) f.write(data.data() + i, std::min<size_t>(100003, data.size() - i));
        f.close();
        CHECK(thrown([&] { f.write(data.data(), 1); }) == frame::CLOSED_STREAM);
      }
      dump(dir + "/" + c.name, sink.str());
      {   // the same frame with the blocks assembled on the host (round-1/2 path) instead of on the device: same bytes
        std::ostringstream sink2;
        BatchEngine he; he.hostAssembly = true;
        LZ4FrameOutputStream f2(sink2, c.bs, c.known ? (int64_t)data.size() : -1, c.bits, he, 3);
        for (size_t i = 0; i < data.size(); i += 100003) f2.write(data.data() + i, std::min<size_t>(100003, data.size() - i));
        f2.close();
        CHECK(sink2.str() == sink.str());
      }
      std::istringstream src(sink.str());
      LZ4FrameInputStream in(src, false, BatchEngine(), 2);
      if (c.known) CHECK(in.isExpectedContentSizeDefined() && in.getExpectedContentSize() == (int64_t)data.size());
      CHECK(in.readAll() == data);
      if (std::string(c.name) == "cpp_frame_all.lz4") all_frame = sink.str();
    }
    {  // known bytes of an empty frame (LZ4FrameOutputStream defaults: FLG 0x60, BD 0x70, HC 0x73)
      std::ostringstream sink;
      { LZ4FrameOutputStream f(sink); }
      const uint8_t want[] = {0x04, 0x22, 0x4d, 0x18, 0x60, 0x70, 0x73, 0, 0, 0, 0};
      CHECK(sink.str() == std::string((const char*)want, sizeof want));
    }
    {  // error paths, in the reference's words
      std::string bad = all_frame;
      bad[14] ^= 1;
      { std::istringstream s(bad); LZ4FrameInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == frame::DESCRIPTOR_HASH_MISMATCH); }
      bad = all_frame; bad[bad.size() - 1] ^= 1;
      { std::istringstream s(bad); LZ4FrameInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == "Content checksum mismatch"); }
      bad = all_frame; bad[15 + 4 + 10] ^= 0x55;   // payload of block 0
      { std::istringstream s(bad); LZ4FrameInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == frame::BLOCK_HASH_MISMATCH); }
      { std::istringstream s(all_frame.substr(0, all_frame.size() - 6)); LZ4FrameInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == frame::PREMATURE_EOS); }
      { std::istringstream s(std::string("\x01\x02\x03\x04rest")); LZ4FrameInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == frame::NOT_SUPPORTED); }
      bad = all_frame; bad[4] = (char)(0x40 | 0x1C);
      { std::istringstream s(bad); LZ4FrameInputStream in(s); CHECK(thrown([&] { in.readAll(); }).find("BLOCK_INDEPENDENCE") != std::string::npos); }
      // two frames and a skippable one in a row; readSingleFrame stops after the first
      const std::string skip = std::string("\x53\x2a\x4d\x18\x03\x00\x00\x00xyz", 11);
      { std::istringstream s(skip + all_frame + skip + all_frame); LZ4FrameInputStream in(s); bytes twice = data; twice.insert(twice.end(), data.begin(), data.end()); CHECK(in.readAll() == twice); }
      { std::istringstream s(all_frame + all_frame); LZ4FrameInputStream in(s, true); CHECK(in.readAll() == data); }
    }
    {  // a frame written by the lz4 CLI, if the pytest side provided one
      const bytes cli = slurp(dir + "/cli.lz4"), cli_in = slurp(dir + "/cli_payload.bin");
      if (!cli.empty()) {
        std::istringstream s(std::string(cli.begin(), cli.end()));
        LZ4FrameInputStream in(s);
        CHECK(in.readAll() == cli_in);
      }
    }
    // ---- lz4-java Block stream ----
    {
      std::ostringstream sink;
      { LZ4BlockOutputStream f(sink, 1 << 16, BatchEngine(), false, 5); for (size_t i = 0; i < data.size(); i += 70001) f.write(data.data() + i, std::min<size_t>(70001, data.size() - i)); }
      dump(dir + "/cpp_stream.blk", sink.str());
      {   // blocks assembled on the host instead of on the device: same bytes
        std::ostringstream sink2;
        BatchEngine he; he.hostAssembly = true;
        { LZ4BlockOutputStream f2(sink2, 1 << 16, he, false, 5); for (size_t i = 0; i < data.size(); i += 70001) f2.write(data.data() + i, std::min<size_t>(70001, data.size() - i)); }
        CHECK(sink2.str() == sink.str());
      }
      { std::istringstream s(sink.str()); LZ4BlockInputStream in(s, true, BatchEngine(), 7); CHECK(in.readAll() == data); }
      std::string bad = sink.str();
      bad[3] ^= 1;
      { std::istringstream s(bad); LZ4BlockInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == "Stream is corrupted"); }
      bad = sink.str(); bad[21 + 30] ^= 0x10;   // payload of block 0
      { std::istringstream s(bad); LZ4BlockInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == "Stream is corrupted"); }
      { std::istringstream s(sink.str().substr(0, sink.str().size() - 1)); LZ4BlockInputStream in(s); CHECK(thrown([&] { in.readAll(); }) == "Stream ended prematurely"); }
      CHECK(thrown([&] { std::ostringstream o; LZ4BlockOutputStream f(o, 63); }).find("blockSize must be >= 64") != std::string::npos);
    }
    // ---- rounds 4 / 5: the READ side on the device (BatchEngine::containerDecode; the default) against the host walk (hostWalk = true):
    // the same bytes and the same exception, at the same point of the stream -- intact streams and the same streams damaged / cut;
    // a frame / a block stream followed by other bytes leaves the input right behind it; an LZ4Block header whose compressedLen
    // exceeds the reader's chunk in the middle of a long stream ends in the reference's exception, not in a loop ----
    {
      BatchEngine hw; hw.hostWalk = true;
      auto drainF = [&](const std::string& st, const BatchEngine& e, bool single, size_t batch, std::string& exc) {
        std::istringstream s(st);
        LZ4FrameInputStream in(s, single, e, batch);
        bytes got, buf(50000);
        exc.clear();
        try { for (size_t k; (k = in.read(buf.data(), buf.size())) != 0;) got.insert(got.end(), buf.begin(), buf.begin() + (std::ptrdiff_t)k); }
        catch (const std::exception& x) { exc = x.what(); }
        return got;
      };
      auto drainB = [&](const std::string& st, const BatchEngine& e, size_t batch, std::string& exc) {
        std::istringstream s(st);
        LZ4BlockInputStream in(s, true, e, batch);
        bytes got, buf(50000);
        exc.clear();
        try { for (size_t k; (k = in.read(buf.data(), buf.size())) != 0;) got.insert(got.end(), buf.begin(), buf.begin() + (std::ptrdiff_t)k); }
        catch (const std::exception& x) { exc = x.what(); }
        return got;
      };
      std::ostringstream bs;
      { LZ4BlockOutputStream f(bs, 1 << 12, BatchEngine(), false, 64); f.write(data.data(), data.size()); }
      const std::string blk = bs.str();
      uint32_t rs = 99;
      auto rnd = [&](size_t n) { rs = rs * 1664525u + 1013904223u; return (size_t)(rs >> 8) % n; };
      int n_err = 0;
      for (int t = 0; t < 40; t++) {
        std::string f = all_frame, b = blk;
        if (t) {
          if (t % 4 == 0) { f.resize(1 + rnd(f.size() - 1)); b.resize(1 + rnd(b.size() - 1)); }
          else { f[rnd(f.size())] ^= (char)(1 << rnd(8)); b[rnd(b.size())] ^= (char)(1 << rnd(8)); }
        }
        std::string e1, e2;
        const bytes g1 = drainF(f, BatchEngine(), false, 3, e1), g2 = drainF(f, hw, false, 3, e2);
        CHECK(g1 == g2 && e1 == e2);
        const bytes h1 = drainB(b, BatchEngine(), 5, e1), h2 = drainB(b, hw, 5, e2);
        CHECK(h1 == h2 && e1 == e2);
        if (t == 0) CHECK(g1 == data && h1 == data && e1.empty());
        n_err += !e1.empty();
      }
      CHECK(n_err > 10);
      {   // a single frame / a block stream embedded in another protocol
        const std::string tail = "TRAILER-TRAILER-TRAILER";
        std::istringstream s(all_frame + tail);
        LZ4FrameInputStream in(s, true, BatchEngine(), 4);
        CHECK(in.readAll() == data);
        std::string left((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
        CHECK(left == tail);
        std::istringstream s2(blk + tail);
        LZ4BlockInputStream in2(s2, true, BatchEngine(), 16);
        CHECK(in2.readAll() == data);
        std::string left2((std::istreambuf_iterator<char>(s2)), std::istreambuf_iterator<char>());
        CHECK(left2 == tail);
      }
      {   // compressedLen of block 5: + 1 GiB (one flipped bit), and 0x7FFFFFF0
        size_t h = 0;
        for (int k = 0; k < 5; k++) h += 21u + detail::getLE32((const uint8_t*)blk.data() + h + 9);
        for (int v = 0; v < 2; v++) {
          std::string b = blk + std::string(300000, '\0');
          if (v == 0) b[h + 12] ^= 0x40; else { b[h + 9] = (char)0xF0; b[h + 10] = b[h + 11] = (char)0xFF; b[h + 12] = 0x7F; }
          std::string e1, e2;
          const bytes g1 = drainB(b, BatchEngine(), 4, e1), g2 = drainB(b, hw, 4, e2);
          CHECK(!e1.empty() && e1 == e2 && g1 == g2 && g1.size() >= 4u * 4096u);
        }
      }
    }
    // ---- WithLength ----
    {
      const std::vector<bytes> bufs = {bytes(), bytes(data.begin(), data.begin() + 13), bytes(data.begin() + 1000, data.begin() + 70000), bytes(5000, 7)};
      const std::vector<bytes> c = LZ4CompressorWithLength().compressMany(bufs);
      for (size_t i = 0; i < bufs.size(); i++) {
        CHECK(LZ4DecompressorWithLength::getDecompressedLength(c[i]) == (int)bufs[i].size());
        bytes direct = LZ4Factory::hipInstance().fastCompressor().compress(bufs[i]);
        CHECK(bytes(c[i].begin() + 4, c[i].end()) == direct);   // the block inside is the plain LZ4 block
      }
      CHECK(LZ4DecompressorWithLength(true).decompressMany(c) == bufs && LZ4DecompressorWithLength(false).decompressMany(c) == bufs);
      CHECK(LZ4CompressorWithLength().maxCompressedLength(1000) == 1000 + 1000 / 255 + 16 + 4);
      bytes bad = c[2]; bad[4] = 0xFF; bad[5] = bad[6] = bad[7] = bad[8] = 0xFF;
      CHECK(thrown([&] { LZ4DecompressorWithLength(false).decompress(bad); }).find("Error decoding offset") == 0);
      const std::vector<bytes> h = LZ4CompressorWithLength(BatchEngine{9}).compressMany({bufs[2]});
      CHECK(LZ4DecompressorWithLength().decompressMany(h)[0] == bufs[2] && h[0].size() <= c[2].size());
    }
    // ---- block stream with a caller-supplied checksum (the 5-argument constructor): a plain byte sum stands in for Adler32 / CRC32 ----
    {
      const blockstream::Checksum sum = [](const uint8_t* p, size_t n) { uint32_t v = 1; for (size_t i = 0; i < n; i++) v = v * 31u + p[i]; return v; };
      std::ostringstream os;
      { LZ4BlockOutputStream w(os, 1 << 14, BatchEngine(), false, 3, sum); w.write(data.data(), 100000); w.close(); }
      const std::string st = os.str();
      std::istringstream is(st);
      LZ4BlockInputStream r(is, true, BatchEngine(), 5, sum);
      CHECK(r.available() == 0 && r.skip(0) == 0 && !r.markSupported());
      CHECK(r.skip(12345) == 12345);
      CHECK(r.readAll() == bytes(data.begin() + 12345, data.begin() + 100000) && r.skip(5) == 0);
      std::istringstream is2(st);
      CHECK(thrown([&] { LZ4BlockInputStream r2(is2); r2.readAll(); }) == "Stream is corrupted");   // the default checksum rejects it
    }
    // ---- streaming xxhash (XXHashFactory.java:186-203 self-test shape): any split == the one-shot hash; reset(); closed state ----
    {
      auto& xf = xxhash::XXHashFactory::hipInstance();
      auto h32 = xf.newStreamingHash32((int32_t)0x9747b28c);
      auto h64 = xf.newStreamingHash64(-7);
      CHECK(h32->getValue() == xf.hash32().hash(data, 0, 0, (int32_t)0x9747b28c));
      size_t pos = 0, step = 1;
      while (pos < data.size()) {
        const size_t n = std::min(step, data.size() - pos);
        h32->update(data, (int)pos, (int)n);
        h64->update(data, (int)pos, (int)n);
        pos += n;
        step = step * 3 + 5;
      }
      CHECK(h32->getValue() == xf.hash32().hash(data, 0, (int)data.size(), (int32_t)0x9747b28c));
      CHECK(h64->getValue() == xf.hash64().hash(data, 0, (int)data.size(), -7));
      CHECK(h32->checksumValue() == (int64_t)((uint32_t)h32->getValue() & 0xFFFFFFFu));
      h32->reset();
      h32->update(data, 5, 100);
      CHECK(h32->getValue() == xf.hash32().hash(data, 5, 100, (int32_t)0x9747b28c));
      h32->close();
      CHECK(thrown([&] { h32->getValue(); }) == "Already finalized");
    }
    printf("stream mirror ok (%zu payload bytes)\n", data.size());
    return 0;
  } catch (const LZ4Exception& e) {
    fprintf(stderr, "LZ4Exception: %s\n", e.what());
    return 3;
  }
}
