// tests/cpp/host_mirror_test.cpp -- exercises the C++ host mirror (lz4-java_amd/host/lz4hip.hpp) the way
// LZ4FactoryTest.java / LZ4Test.java exercise the Java API.  Built by the CPU suite (link check), run by
// the GPU suite.  Exit code 0 = all good; with no GPU it must fail loudly (exit code 3).
#include <cstdio>
#include <cstring>
#include "../../lz4-java_amd/host/lz4hip.hpp"

using namespace net::jpountz;

static std::string hex(const bytes& b) { std::string s; char t[3]; for (uint8_t c : b) { snprintf(t, 3, "%02x", c); s += t; } return s; }

int main() {
  try {
    lz4::LZ4Factory& f = lz4::LZ4Factory::hipInstance();
    const char* txt = "abcd      abcdefghij";
    bytes in(txt, txt + 20);
    bytes c = f.fastCompressor().compress(in);
    if (hex(c) != "5161626364200100a06162636465666768696a") { fprintf(stderr, "golden mismatch %s\n", hex(c).c_str()); return 1; }
    if (f.safeDecompressor().decompress(c, 20) != in || f.fastDecompressor().decompress(c, 20) != in) return 1;
    bytes big(300000);
    for (size_t i = 0; i < big.size(); i++) big[i] = (uint8_t)((i * 2654435761u >> 13) % 7 + (i % 97 == 0 ? i : 0));
    bytes cb = f.fastCompressor().compress(big);
    if (f.safeDecompressor().decompress(cb, (int)big.size()) != big) return 1;
    bool threw = false;
    try { bytes small(cb.size() - 1); f.fastCompressor().compress(big, 0, (int)big.size(), small, 0, (int)small.size()); }
    catch (const lz4::LZ4Exception& e) { threw = std::string(e.what()) == "maxDestLen is too small"; }
    if (!threw) return 1;
    threw = false;
    try { bytes out(10); f.safeDecompressor().decompress(bytes{96, 42, 43, 44, 45, 46, 47, 5, 0}, 0, 9, out, 0, 10); }
    catch (const lz4::LZ4Exception& e) { threw = std::string(e.what()) == "Error decoding offset 2 of input buffer"; }
    if (!threw) return 1;
    threw = false;
    try { bytes out(10); f.fastCompressor().compress(in, 5, 30, out, 0, 10); } catch (const std::out_of_range&) { threw = true; }
    if (!threw) return 1;
    const char* s = "12345345234572";
    bytes sb(s, s + 14);
    if ((uint32_t)xxhash::XXHashFactory::hipInstance().hash32().hash(sb, 0, 14, (int32_t)0x9747b28c) != 0x1e34488cu) return 1;
    if ((uint64_t)xxhash::XXHashFactory::hipInstance().hash64().hash(sb, 0, 14, 0) != 0xf46bd83bde991b30ull) return 1;
    printf("host mirror ok (%s)\n", f.toString().c_str());
    return 0;
  } catch (const lz4::LZ4Exception& e) {
    fprintf(stderr, "LZ4Exception: %s\n", e.what());
    return 3;
  }
}
