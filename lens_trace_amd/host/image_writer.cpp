// image_writer.cpp -- ImageWriter::writeBufferToImage for the HIP backend's host library (reference:
// src/image_writer.cpp:7-24, which narrows value*255 to 8 bits and hands the bytes to stb_image_write at quality 100).
// stb is not part of this repository; the encoder below is a from-scratch baseline JPEG writer (ITU-T T.81):
// 8x8 forward DCT in double precision, JFIF YCbCr without chroma subsampling, the quantisation tables of Annex K scaled
// by the usual quality rule (quality 100 -> all ones), and Huffman tables built PER IMAGE from the symbol statistics
// (Annex K.2's code-size procedure with its 16-bit length limit), so nothing depends on memorised tables.
#include "lens_trace/hip/lens_trace_api.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// T.81 Annex K.1, tables K.1 (luminance) and K.2 (chrominance), row-major
const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                            69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                            81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72,  92,  95,  98,  112, 100, 103, 99};
const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                              99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

struct Huffman {
  uint64_t freq[257] = {0};
  uint8_t bits[17] = {0};      // bits[l] = number of codes of length l
  std::vector<uint8_t> vals;   // symbols in code order
  uint16_t code[256] = {0};
  uint8_t size[256] = {0};

  // T.81 Annex K.2: code sizes from frequencies (with the reserved all-ones code point), limited to 16 bits
  void build() {
    uint64_t f[257];
    int codesize[257] = {0}, others[257];
    memcpy(f, freq, sizeof f);
    for (int i = 0; i < 257; i++) others[i] = -1;
    f[256] = 1;   // guarantees that no real symbol gets the all-ones code
    for (;;) {
      int c1 = -1, c2 = -1;
      uint64_t v = ~0ull;
      for (int i = 0; i <= 256; i++) if (f[i] && f[i] <= v) { v = f[i]; c1 = i; }
      v = ~0ull;
      for (int i = 0; i <= 256; i++) if (f[i] && f[i] <= v && i != c1) { v = f[i]; c2 = i; }
      if (c2 < 0) break;
      f[c1] += f[c2];
      f[c2] = 0;
      codesize[c1]++;
      while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
      others[c1] = c2;
      codesize[c2]++;
      while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    int count[258] = {0};   // (a code can be no longer than the number of symbols)
    for (int i = 0; i <= 256; i++) if (codesize[i]) count[codesize[i]]++;
    for (int i = 257; i > 16; i--) {
      while (count[i] > 0) {
        int j = i - 2;
        while (count[j] == 0) j--;
        count[i] -= 2;
        count[i - 1]++;
        count[j + 1] += 2;
        count[j]--;
      }
    }
    int l = 16;
    while (count[l] == 0) l--;
    count[l]--;   // drop the reserved code point
    for (int i = 1; i <= 16; i++) bits[i] = (uint8_t)count[i];
    vals.clear();
    for (int len = 1; len <= 257; len++)
      for (int i = 0; i < 256; i++) if (codesize[i] == len) vals.push_back((uint8_t)i);
    uint16_t c = 0;
    size_t k = 0;
    for (int len = 1; len <= 16; len++) {
      for (int i = 0; i < bits[len]; i++, k++) { code[vals[k]] = c++; size[vals[k]] = (uint8_t)len; }
      c <<= 1;
    }
  }
};

struct BitWriter {
  std::vector<uint8_t>& out;
  uint32_t acc = 0;
  int n = 0;
  explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
  void put(uint32_t v, int len) {
    acc = (acc << len) | (v & ((1u << len) - 1u));
    n += len;
    while (n >= 8) {
      const uint8_t b = (uint8_t)(acc >> (n - 8));
      out.push_back(b);
      if (b == 0xff) out.push_back(0);   // byte stuffing
      n -= 8;
    }
  }
  void flush() { if (n) put(0x7f, 8 - n); }   // pad with ones
};

int category(int v) {
  int a = v < 0 ? -v : v, c = 0;
  while (a) { c++; a >>= 1; }
  return c;
}

// one block's symbol stream: (symbol, extra bits, extra length); table 0 = DC, 1 = AC
struct Sym { uint8_t table, symbol, len; uint16_t extra; };

void blockSymbols(const int* q, int& dcPred, std::vector<Sym>& out) {
  const int diff = q[0] - dcPred;
  dcPred = q[0];
  const int c = category(diff);
  out.push_back({0, (uint8_t)c, (uint8_t)c, (uint16_t)((diff < 0 ? diff - 1 : diff) & ((1 << c) - 1))});
  int run = 0;
  for (int k = 1; k < 64; k++) {
    const int v = q[kZigzag[k]];
    if (v == 0) { run++; continue; }
    while (run > 15) { out.push_back({1, 0xf0, 0, 0}); run -= 16; }
    const int s = category(v);
    out.push_back({1, (uint8_t)((run << 4) | s), (uint8_t)s, (uint16_t)((v < 0 ? v - 1 : v) & ((1 << s) - 1))});
    run = 0;
  }
  if (run) out.push_back({1, 0x00, 0, 0});
}

void put16(std::vector<uint8_t>& o, unsigned v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }

}  // namespace

// Encodes 8-bit interleaved pixels (components = 1: grey, >= 3: the first three are R, G, B) as a baseline JFIF file.
bool lt_encode_jpeg(const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t components, int quality, std::vector<uint8_t>& file) {
  if (!pixels || width == 0 || height == 0 || width > 65535 || height > 65535 || components == 0 || components == 2) return false;
  const int nc = components >= 3 ? 3 : 1;
  quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  uint8_t qt[2][64];
  for (int i = 0; i < 64; i++) {
    const int l = (kLumaQ[i] * scale + 50) / 100, c = (kChromaQ[i] * scale + 50) / 100;
    qt[0][i] = (uint8_t)(l < 1 ? 1 : (l > 255 ? 255 : l));
    qt[1][i] = (uint8_t)(c < 1 ? 1 : (c > 255 ? 255 : c));
  }
  double cosT[8][8];
  for (int u = 0; u < 8; u++)
    for (int x = 0; x < 8; x++) cosT[u][x] = std::cos((2 * x + 1) * u * M_PI / 16.0) * (u == 0 ? std::sqrt(0.125) : 0.5);

  // pass 1: quantised coefficients -> symbols, per component interleaved block by block (one 8x8 block of each per MCU)
  std::vector<Sym> syms[3];
  int dcPred[3] = {0, 0, 0};
  std::vector<uint32_t> order;   // component of each block, in stream order
  const uint32_t bw = (width + 7) / 8, bh = (height + 7) / 8;
  for (uint32_t by = 0; by < bh; by++) {
    for (uint32_t bx = 0; bx < bw; bx++) {
      double comp[3][64];
      for (int y = 0; y < 8; y++) {
        for (int x = 0; x < 8; x++) {
          const uint32_t px = bx * 8 + x < width ? bx * 8 + x : width - 1, py = by * 8 + y < height ? by * 8 + y : height - 1;
          const uint8_t* p = pixels + ((size_t)py * width + px) * components;
          if (nc == 1) {
            comp[0][y * 8 + x] = p[0] - 128.0;
          } else {
            const double r = p[0], g = p[1], b = p[2];
            comp[0][y * 8 + x] = 0.299 * r + 0.587 * g + 0.114 * b - 128.0;
            comp[1][y * 8 + x] = -0.168735892 * r - 0.331264108 * g + 0.5 * b;
            comp[2][y * 8 + x] = 0.5 * r - 0.418687589 * g - 0.081312411 * b;
          }
        }
      }
      for (int ci = 0; ci < nc; ci++) {
        double tmp[64];
        int q[64];
        for (int y = 0; y < 8; y++)
          for (int u = 0; u < 8; u++) {
            double s = 0;
            for (int x = 0; x < 8; x++) s += comp[ci][y * 8 + x] * cosT[u][x];
            tmp[y * 8 + u] = s;
          }
        for (int v = 0; v < 8; v++)
          for (int u = 0; u < 8; u++) {
            double s = 0;
            for (int y = 0; y < 8; y++) s += tmp[y * 8 + u] * cosT[v][y];
            q[v * 8 + u] = (int)std::lround(s / qt[ci ? 1 : 0][v * 8 + u]);
          }
        const size_t before = syms[ci].size();
        blockSymbols(q, dcPred[ci], syms[ci]);
        order.push_back((uint32_t)ci);
        order.push_back((uint32_t)(syms[ci].size() - before));
      }
    }
  }
  // Huffman tables: [0] DC luma, [1] AC luma, [2] DC chroma, [3] AC chroma
  Huffman h[4];
  for (int ci = 0; ci < nc; ci++)
    for (const Sym& s : syms[ci]) h[(ci ? 2 : 0) + s.table].freq[s.symbol]++;
  const int tables = nc == 1 ? 2 : 4;
  for (int t = 0; t < tables; t++) h[t].build();

  file.clear();
  file.push_back(0xff); file.push_back(0xd8);
  const uint8_t app0[] = {0xff, 0xe0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  file.insert(file.end(), app0, app0 + sizeof app0);
  for (int t = 0; t < (nc == 1 ? 1 : 2); t++) {
    file.push_back(0xff); file.push_back(0xdb); put16(file, 67); file.push_back((uint8_t)t);
    for (int i = 0; i < 64; i++) file.push_back(qt[t][kZigzag[i]]);
  }
  file.push_back(0xff); file.push_back(0xc0); put16(file, 8 + 3 * nc); file.push_back(8); put16(file, height); put16(file, width);
  file.push_back((uint8_t)nc);
  for (int ci = 0; ci < nc; ci++) { file.push_back((uint8_t)(ci + 1)); file.push_back(0x11); file.push_back(ci ? 1 : 0); }
  for (int t = 0; t < tables; t++) {
    file.push_back(0xff); file.push_back(0xc4); put16(file, 19 + (unsigned)h[t].vals.size());
    file.push_back((uint8_t)(((t & 1) << 4) | (t >> 1)));   // class (0 DC, 1 AC) << 4 | destination
    for (int l = 1; l <= 16; l++) file.push_back(h[t].bits[l]);
    file.insert(file.end(), h[t].vals.begin(), h[t].vals.end());
  }
  file.push_back(0xff); file.push_back(0xda); put16(file, 6 + 2 * nc); file.push_back((uint8_t)nc);
  for (int ci = 0; ci < nc; ci++) { file.push_back((uint8_t)(ci + 1)); file.push_back(ci ? 0x11 : 0x00); }
  file.push_back(0); file.push_back(63); file.push_back(0);

  // pass 2: the entropy-coded segment
  BitWriter bwr(file);
  size_t pos[3] = {0, 0, 0};
  for (size_t i = 0; i < order.size(); i += 2) {
    const uint32_t ci = order[i], n = order[i + 1];
    for (uint32_t k = 0; k < n; k++) {
      const Sym& s = syms[ci][pos[ci]++];
      const Huffman& t = h[(ci ? 2 : 0) + s.table];
      bwr.put(t.code[s.symbol], t.size[s.symbol]);
      if (s.len) bwr.put(s.extra, s.len);
    }
  }
  bwr.flush();
  file.push_back(0xff); file.push_back(0xd9);
  return true;
}

// src/image_writer.cpp:7-24: `pWriteBuffer[x] = pImageBuffer[x] * 255` into a char buffer, then a quality-100 JPEG.
void ImageWriter::writeBufferToImage(BufferToImageProperties p) {
  if (p.sType != STRUCTURE_TYPE_BUFFER_TO_IMAGE_PROPERTIES) return;
  const float* src = (const float*)p.pBuffer;
  const uint64_t W = p.imageDimensions[0], H = p.imageDimensions[1], D = p.imageDimensions[2];
  std::vector<uint8_t> bytes((size_t)(W * H * D));
  for (size_t i = 0; i < bytes.size(); i++) bytes[i] = (uint8_t)(int)(src[i] * 255);   // the reference narrows float -> char; through int this is defined
  std::vector<uint8_t> file;
  if (!lt_encode_jpeg(bytes.data(), (uint32_t)W, (uint32_t)H, (uint32_t)D, 100, file)) {
    printf("ERROR: cannot encode a %llu x %llu x %llu image as JPEG\n", (unsigned long long)W, (unsigned long long)H, (unsigned long long)D);
    return;
  }
  FILE* f = fopen(p.filename, "wb");
  if (!f) { printf("ERROR: cannot write %s\n", p.filename); return; }
  fwrite(file.data(), 1, file.size(), f);
  fclose(f);
}

extern "C" int lt_host_write_jpeg(const char* path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t components, int quality) {
  std::vector<uint8_t> file;
  if (!lt_encode_jpeg(pixels, width, height, components, quality, file)) return 1;
  FILE* f = fopen(path, "wb");
  if (!f) return 2;
  const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
  fclose(f);
  return ok ? 0 : 2;
}
