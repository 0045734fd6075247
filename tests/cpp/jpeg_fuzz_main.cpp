// jpeg_fuzz_main.cpp -- negative / fuzz driver for host/jpeg.cpp, built with -fsanitize=address,undefined by
// tests/test_dataset.py (CPU only).  The reference decodes JPEGs with the hardened cv::imdecode
// (lcm_util/decode_image.cpp:11-32); the own decoder must reject malformed input with an exception and never
// read or write out of bounds.
//   jpeg_fuzz_main <seed.jpg> <n_mutations>
// Decodes every truncation of the file, n random byte mutations of it, and a set of crafted headers.  Prints
// "ok <decoded> <rejected>"; any sanitizer report aborts the process.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <vector>

#include "dataset.hpp"

using namespace bm;

static int g_ok = 0, g_bad = 0;

static void Try(const std::vector<uint8_t>& b) {
  for (int color = 0; color < 2; ++color) {
    try {
      core::Image1b gray;
      core::Image3b bgr;
      core::DecodeJpeg(b.data(), b.size(), color != 0, &gray, &bgr);
      ++g_ok;
    } catch (const std::exception&) {
      ++g_bad;
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::ifstream in(argv[1], std::ios::binary);
  std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (file.size() < 100) return 3;
  const int n = atoi(argv[2]);
  Try(file);
  if (g_ok != 2) return 4;  // the seed itself must decode
  // every truncation
  for (size_t len = 0; len < file.size(); ++len) Try(std::vector<uint8_t>(file.begin(), file.begin() + (long)len));
  // random mutations: 1-4 bytes, biased towards the headers
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  size_t hdr = file.size();
  for (size_t i = 0; i + 1 < file.size(); ++i)
    if (file[i] == 0xFF && file[i + 1] == 0xDA) {
      hdr = i + 16;
      break;
    }
  for (int k = 0; k < n; ++k) {
    std::vector<uint8_t> b = file;
    const int nb = 1 + (int)(rnd() % 4);
    for (int j = 0; j < nb; ++j) {
      const size_t range = (rnd() % 3) ? (hdr < b.size() ? hdr : b.size()) : b.size();
      b[rnd() % range] = (uint8_t)rnd();
    }
    Try(b);
  }
  // crafted: walk the segments and poke the fields the advisor named
  for (size_t i = 2; i + 4 < file.size();) {
    if (file[i] != 0xFF) break;
    const int m = file[i + 1];
    const size_t len = ((size_t)file[i + 2] << 8) | file[i + 3];
    const size_t seg = i + 4;
    if (m == 0xDA) {
      const int ns = file[seg];
      for (int v : {0, 4, 255}) {  // component count of the scan
        std::vector<uint8_t> b = file;
        b[seg] = (uint8_t)v;
        Try(b);
      }
      for (int c = 0; c < ns; ++c)
        for (int v : {0x40, 0x04, 0xFF, 0x33, 0x3F}) {  // table selectors td / ta > 3
          std::vector<uint8_t> b = file;
          b[seg + 2 + 2 * c] = (uint8_t)v;
          Try(b);
        }
      {  // the same component twice
        std::vector<uint8_t> b = file;
        if (ns > 1) b[seg + 3] = b[seg + 1];
        Try(b);
      }
      break;
    }
    if (m == 0xC0 || m == 0xC1) {
      std::vector<uint8_t> b = file;  // a second, larger frame header behind the first scan is covered by
      b.insert(b.begin() + (long)(i + 2 + len), file.begin() + (long)i, file.begin() + (long)(i + 2 + len));
      b[i + 2 + len + 5] = 0x7F;      // ... this duplicate with another height
      Try(b);
      for (int v : {0, 2, 4, 200}) {
        std::vector<uint8_t> c = file;
        c[seg + 5] = (uint8_t)v;  // component count
        Try(c);
      }
    }
    if (m == 0xC4 || m == 0xDB) {
      for (size_t cut : {(size_t)3, (size_t)10, len - 1}) {  // segment length shorter than its tables
        if (cut < 2 || cut >= len) continue;
        std::vector<uint8_t> b = file;
        b[i + 2] = (uint8_t)(cut >> 8);
        b[i + 3] = (uint8_t)cut;
        Try(b);
      }
      std::vector<uint8_t> b = file;
      for (size_t k = seg + 1; k < seg + 17 && k < b.size(); ++k) b[k] = 0xFF;  // code counts / table bytes
      Try(b);
      // a DC table whose symbols are all 16+: categories that would shift by 31
      if (m == 0xC4 && (file[seg] >> 4) == 0) {
        std::vector<uint8_t> d = file;
        int total = 0;
        for (int k = 0; k < 16; ++k) total += d[seg + 1 + (size_t)k];
        for (int k = 0; k < total && seg + 17 + (size_t)k < d.size(); ++k) d[seg + 17 + (size_t)k] = (uint8_t)(16 + k % 200);
        Try(d);
      }
    }
    i += 2 + len;
  }
  // a file that ends inside the last table segment of the buffer
  printf("ok %d %d\n", g_ok, g_bad);
  return 0;
}
