// jpeg.cpp -- baseline / extended-sequential Huffman JPEG decoder for the ingest path (dataset.hpp):
// what cv::imread / cv::imdecode do for the reference's 8-bit JPEG inputs (test/resources/caddy_32_*.jpg,
// the LCM image_t payloads of lcm_util/decode_image.cpp:11-32).  Written from the JPEG standard (ITU T.81) and
// the published algorithms of the IJG / libjpeg-turbo decoder that OpenCV links: the accurate integer inverse
// DCT ("ISLOW", Loeffler-Ligtenberg-Moschytz, 13-bit constants), triangle-filter ("fancy") chroma upsampling
// and the 16-bit fixed-point YCbCr -> RGB tables, so that the output is bit-identical to that library's
// (checked against Pillow's libjpeg-turbo in tests/test_dataset.py).  Gray output is the luma plane itself
// (JCS_GRAYSCALE), colour output is BGR.  Not supported: progressive, arithmetic coding, 12-bit, CMYK.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "dataset.hpp"

namespace bm {
namespace core {
namespace {

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int blocks_w = 0, blocks_h = 0;  // in 8x8 blocks, padded to whole MCUs
  int width = 0, height = 0;       // downsampled real size
  std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8
  int pred = 0;
};

struct Huff {
  // canonical decoding tables (T.81 F.2.2.3)
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int nbits = 0;
  bool hit_marker = false;
  int bit() {
    if (nbits == 0) {
      int b = 0;
      if (p < end && !hit_marker) {
        b = *p++;
        if (b == 0xFF) {
          const int b2 = p < end ? *p : 0xD9;
          if (b2 == 0) ++p;          // stuffed zero
          else { hit_marker = true; --p; b = 0; }  // a marker: feed zeros from here on
        }
      }
      acc = (uint32_t)b;
      nbits = 8;
    }
    --nbits;
    return (acc >> nbits) & 1;
  }
  int bits(int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) v = (v << 1) | bit();
    return v;
  }
  void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

int DecodeSymbol(BitReader& br, const Huff& h) {
  int code = br.bit();
  int len = 1;
  while (len <= 16 && code > h.maxcode[len]) {
    code = (code << 1) | br.bit();
    ++len;
  }
  if (len > 16) throw std::runtime_error("JPEG: bad Huffman code");
  return h.vals[h.valptr[len] + code - h.mincode[len]];
}

// t is 1..16 here: callers reject larger categories (baseline allows DC <= 11, AC <= 10; the extended process 15/14)
int Extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t Clamp(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// coefficient * quantiser in 64 bits, clamped to what the inverse DCT's 32-bit arithmetic takes without overflow: a valid
// 8-bit baseline stream stays far inside (|coef| <= 2^15, the clamp never acts); a hostile one (16-bit quantiser entries
// times 15-bit values) would otherwise be signed-overflow UB
inline int Dequant(int v, int q) {
  const long long p = (long long)v * (long long)q;
  const long long lim = 1 << 20;
  return (int)(p < -lim ? -lim : (p > lim ? lim : p));
}

// Accurate integer inverse DCT (the "slow-but-accurate" one: CONST_BITS 13, PASS1_BITS 2), coefficients already
// dequantised, output level-shifted by 128 and clamped.
void IdctIslow(const int* coef, uint8_t* out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
  int ws[64];
  auto descale = [](long x, int n) -> int { return (int)((x + (1L << (n - 1))) >> n); };
  for (int c = 0; c < 8; ++c) {
    const int* in = coef + c;
    int* w = ws + c;
    if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
      const int dc = in[0] * (1 << P1);
      for (int r = 0; r < 8; ++r) w[8 * r] = dc;
      continue;
    }
    long z2 = in[16], z3 = in[48];
    long z1 = (z2 + z3) * F_0_541;
    long tmp2 = z1 + z3 * (-F_1_847);
    long tmp3 = z1 + z2 * F_0_765;
    z2 = in[0];
    z3 = in[32];
    long tmp0 = (z2 + z3) * (1L << CB);
    long tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56];
    tmp1 = in[40];
    tmp2 = in[24];
    tmp3 = in[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298;
    tmp1 *= F_2_053;
    tmp2 *= F_3_072;
    tmp3 *= F_1_501;
    z1 *= -F_0_899;
    z2 *= -F_2_562;
    z3 *= -F_1_961;
    z4 *= -F_0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    w[0] = descale(tmp10 + tmp3, CB - P1);
    w[56] = descale(tmp10 - tmp3, CB - P1);
    w[8] = descale(tmp11 + tmp2, CB - P1);
    w[48] = descale(tmp11 - tmp2, CB - P1);
    w[16] = descale(tmp12 + tmp1, CB - P1);
    w[40] = descale(tmp12 - tmp1, CB - P1);
    w[24] = descale(tmp13 + tmp0, CB - P1);
    w[32] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; ++r) {
    const int* w = ws + 8 * r;
    uint8_t* o = out + (size_t)r * stride;
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F_0_541;
    long tmp2 = z1 + z3 * (-F_1_847);
    long tmp3 = z1 + z2 * F_0_765;
    long tmp0 = ((long)w[0] + w[4]) * (1L << CB);
    long tmp1 = ((long)w[0] - w[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7];
    tmp1 = w[5];
    tmp2 = w[3];
    tmp3 = w[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298;
    tmp1 *= F_2_053;
    tmp2 *= F_3_072;
    tmp3 *= F_1_501;
    z1 *= -F_0_899;
    z2 *= -F_2_562;
    z3 *= -F_1_961;
    z4 *= -F_0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    constexpr int S = CB + P1 + 3;
    o[0] = Clamp(descale(tmp10 + tmp3, S) + 128);
    o[7] = Clamp(descale(tmp10 - tmp3, S) + 128);
    o[1] = Clamp(descale(tmp11 + tmp2, S) + 128);
    o[6] = Clamp(descale(tmp11 - tmp2, S) + 128);
    o[2] = Clamp(descale(tmp12 + tmp1, S) + 128);
    o[5] = Clamp(descale(tmp12 - tmp1, S) + 128);
    o[3] = Clamp(descale(tmp13 + tmp0, S) + 128);
    o[4] = Clamp(descale(tmp13 - tmp0, S) + 128);
  }
}

uint16_t Be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

// Chroma plane -> full resolution with the triangle filter of the reference decoder ("fancy upsampling"):
// horizontally 3/4 nearer + 1/4 further with alternating rounding, vertically the same on row sums.
// in: cw x chh real samples (stride cstride); out: (cw*hs) x (chh*vs).
void Upsample(const Component& c, int hs, int vs, std::vector<uint8_t>& out, int& ow, int& oh) {
  const int cw = c.width, chh = c.height, cs = c.blocks_w * 8;
  ow = cw * hs;
  oh = chh * vs;
  out.assign((size_t)ow * oh, 0);
  auto in = [&](int y, int x) -> int { return c.plane[(size_t)(y < 0 ? 0 : (y >= chh ? chh - 1 : y)) * cs + x]; };
  if (hs == 1 && vs == 1) {
    for (int y = 0; y < chh; ++y) std::memcpy(&out[(size_t)y * ow], &c.plane[(size_t)y * cs], (size_t)cw);
  } else if (hs == 2 && vs == 1) {
    for (int y = 0; y < chh; ++y) {
      uint8_t* o = &out[(size_t)y * ow];
      if (cw == 1) {
        o[0] = o[1] = (uint8_t)in(y, 0);
        continue;
      }
      o[0] = (uint8_t)in(y, 0);
      o[1] = (uint8_t)((in(y, 0) * 3 + in(y, 1) + 2) >> 2);
      for (int x = 1; x < cw - 1; ++x) {
        const int v = in(y, x) * 3;
        o[2 * x] = (uint8_t)((v + in(y, x - 1) + 1) >> 2);
        o[2 * x + 1] = (uint8_t)((v + in(y, x + 1) + 2) >> 2);
      }
      const int v = in(y, cw - 1) * 3;
      o[2 * cw - 2] = (uint8_t)((v + in(y, cw - 2) + 1) >> 2);
      o[2 * cw - 1] = (uint8_t)in(y, cw - 1);
    }
  } else if (hs == 2 && vs == 2) {
    for (int y = 0; y < chh; ++y)
      for (int v = 0; v < 2; ++v) {
        const int yn = v == 0 ? y - 1 : y + 1;  // the further row; the image edge replicates
        uint8_t* o = &out[(size_t)(2 * y + v) * ow];
        auto colsum = [&](int x) -> int { return in(y, x) * 3 + in(yn, x); };
        if (cw == 1) {
          const int t = colsum(0);
          o[0] = (uint8_t)((t * 4 + 8) >> 4);
          o[1] = (uint8_t)((t * 4 + 7) >> 4);
          continue;
        }
        int thiscol = colsum(0), nextcol = colsum(1), lastcol = 0;
        o[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
        o[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol;
        thiscol = nextcol;
        for (int x = 1; x < cw - 1; ++x) {
          nextcol = colsum(x + 1);
          o[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          o[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol;
          thiscol = nextcol;
        }
        o[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        o[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
      }
  } else {  // other ratios: sample replication (the library's generic path)
    for (int y = 0; y < oh; ++y)
      for (int x = 0; x < ow; ++x) out[(size_t)y * ow + x] = (uint8_t)in(y / vs, x / hs);
  }
}

}  // namespace

int DecodeJpeg(const uint8_t* buf, size_t n, bool want_color, Image1b* gray, Image3b* color) {
  if (n < 4 || buf[0] != 0xFF || buf[1] != 0xD8) throw std::runtime_error("not a JPEG");
  uint16_t qt[4][64] = {};
  Huff hdc[4], hac[4];
  std::vector<Component> comps;
  int width = 0, height = 0, restart = 0, hmax = 1, vmax = 1;
  bool have_frame = false;
  size_t pos = 2;
  auto build = [](Huff& h, const uint8_t* counts, const uint8_t* vals, int nvals) {
    std::memcpy(h.vals, vals, (size_t)nvals);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      h.valptr[len] = k;
      h.mincode[len] = code;
      code += counts[len - 1];
      k += counts[len - 1];
      h.maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    h.present = true;
  };
  for (;;) {
    if (pos + 2 > n) break;  // no EOI: accept what was decoded (checked below)
    if (buf[pos] != 0xFF) throw std::runtime_error("JPEG: marker expected");
    while (pos < n && buf[pos] == 0xFF) ++pos;  // fill bytes
    if (pos >= n) break;
    const int m = buf[pos++];
    if (m == 0xD9) break;
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > n) throw std::runtime_error("JPEG: truncated segment");
    const int len = Be16(buf + pos);
    if (len < 2 || pos + (size_t)len > n) throw std::runtime_error("JPEG: bad segment length");
    const uint8_t* seg = buf + pos + 2;
    const int slen = len - 2;
    if (m == 0xDB) {
      int o = 0;
      while (o < slen) {
        const int pq = seg[o] >> 4, tq = seg[o] & 15;
        ++o;
        if (tq > 3 || pq > 1) throw std::runtime_error("JPEG: bad quantisation table id");
        if (o + (pq ? 128 : 64) > slen) throw std::runtime_error("JPEG: truncated quantisation table");
        for (int i = 0; i < 64; ++i) {
          qt[tq][kZigzag[i]] = pq ? Be16(seg + o) : seg[o];
          o += pq ? 2 : 1;
        }
      }
    } else if (m == 0xC4) {
      int o = 0;
      while (o < slen) {
        const int tc = seg[o] >> 4, th = seg[o] & 15;
        ++o;
        if (o + 16 > slen) throw std::runtime_error("JPEG: truncated Huffman table");
        int total = 0;
        for (int i = 0; i < 16; ++i) total += seg[o + i];
        if (tc > 1 || th > 3 || total > 256 || o + 16 + total > slen) throw std::runtime_error("JPEG: bad Huffman table");
        build(tc ? hac[th] : hdc[th], seg + o, seg + o + 16, total);
        o += 16 + total;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (have_frame) throw std::runtime_error("JPEG: second frame header");  // planes are sized by the first
      if (slen < 6) throw std::runtime_error("JPEG: truncated frame header");
      if (seg[0] != 8) throw std::runtime_error("JPEG: only 8-bit samples are supported");
      height = Be16(seg + 1);
      width = Be16(seg + 3);
      const int nc = seg[5];
      if ((nc != 1 && nc != 3) || width <= 0 || height <= 0) throw std::runtime_error("JPEG: unsupported frame");
      if (slen < 6 + 3 * nc) throw std::runtime_error("JPEG: truncated frame header");
      comps.resize((size_t)nc);
      for (int i = 0; i < nc; ++i) {
        comps[i].id = seg[6 + 3 * i];
        comps[i].h = seg[7 + 3 * i] >> 4;
        comps[i].v = seg[7 + 3 * i] & 15;
        comps[i].tq = seg[8 + 3 * i];
        if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3)
          throw std::runtime_error("JPEG: bad sampling factors");
        hmax = comps[i].h > hmax ? comps[i].h : hmax;
        vmax = comps[i].v > vmax ? comps[i].v : vmax;
      }
      have_frame = true;
    } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      throw std::runtime_error("JPEG: progressive / lossless / arithmetic coding is not supported");
    } else if (m == 0xDD) {
      if (slen < 2) throw std::runtime_error("JPEG: truncated restart interval");
      restart = Be16(seg);
    } else if (m == 0xDA) {
      if (!have_frame) throw std::runtime_error("JPEG: scan before frame header");
      if (slen < 1) throw std::runtime_error("JPEG: truncated scan header");
      const int ns = seg[0];
      if (ns < 1 || ns > (int)comps.size() || slen < 1 + 2 * ns + 3)
        throw std::runtime_error("JPEG: bad scan header");
      std::vector<int> order;
      for (int i = 0; i < ns; ++i) {
        const int cid = seg[1 + 2 * i];
        int idx = -1;
        for (size_t k = 0; k < comps.size(); ++k)
          if (comps[k].id == cid) idx = (int)k;
        if (idx < 0) throw std::runtime_error("JPEG: scan names an unknown component");
        for (int prev : order)
          if (prev == idx) throw std::runtime_error("JPEG: scan names a component twice");
        const int td = seg[2 + 2 * i] >> 4, ta = seg[2 + 2 * i] & 15;
        if (td > 3 || ta > 3) throw std::runtime_error("JPEG: bad Huffman table selector");
        comps[(size_t)idx].td = td;
        comps[(size_t)idx].ta = ta;
        order.push_back(idx);
      }
      // geometry
      const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
      for (Component& c : comps) {
        if (c.plane.empty()) {
          c.blocks_w = mcux * c.h;
          c.blocks_h = mcuy * c.v;
          c.width = (width * c.h + hmax - 1) / hmax;
          c.height = (height * c.v + vmax - 1) / vmax;
          c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
        }
      }
      BitReader br{buf + pos + (size_t)len, buf + n};
      int coef[64];
      auto decode_block = [&](Component& c, int bx, int by) {
        std::memset(coef, 0, sizeof(coef));
        const Huff& dc = hdc[c.td];
        const Huff& ac = hac[c.ta];
        if (!dc.present || !ac.present) throw std::runtime_error("JPEG: missing Huffman table");
        const int t = DecodeSymbol(br, dc);
        if (t > 15) throw std::runtime_error("JPEG: bad DC category");  // 8-bit samples: at most 11 (15 tolerated)
        const int diff = t ? Extend(br.bits(t), t) : 0;
        c.pred += diff;
        // a baseline DC value fits 12 bits; a stream that drifts beyond 16 is hostile (and c.pred * q would overflow)
        if (c.pred < -32768 || c.pred > 32767) throw std::runtime_error("JPEG: DC predictor out of range");
        coef[0] = Dequant(c.pred, qt[c.tq][0]);
        for (int k = 1; k < 64;) {
          const int rs = DecodeSymbol(br, ac);
          const int r = rs >> 4, s = rs & 15;
          if (s == 0) {
            if (r == 15) { k += 16; continue; }
            break;
          }
          k += r;
          if (k > 63) throw std::runtime_error("JPEG: coefficient index out of range");
          coef[kZigzag[k]] = Dequant(Extend(br.bits(s), s), qt[c.tq][kZigzag[k]]);
          ++k;
        }
        if (bx < c.blocks_w && by < c.blocks_h)
          IdctIslow(coef, &c.plane[((size_t)by * 8) * ((size_t)c.blocks_w * 8) + (size_t)bx * 8], c.blocks_w * 8);
      };
      for (int idx : order) comps[(size_t)idx].pred = 0;
      int since_restart = 0;
      auto maybe_restart = [&]() {
        if (!restart) return;
        if (++since_restart == restart) {
          since_restart = 0;
          // skip to the RSTn marker, then continue after it
          const uint8_t* q = br.p;
          while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
          if (q + 1 < br.end) br.p = q + 2;
          br.reset();
          for (int idx : order) comps[(size_t)idx].pred = 0;
        }
      };
      if (ns > 1) {  // interleaved: MCU = h x v blocks of every component
        for (int my = 0; my < mcuy; ++my)
          for (int mx = 0; mx < mcux; ++mx) {
            for (int idx : order) {
              Component& c = comps[(size_t)idx];
              for (int v = 0; v < c.v; ++v)
                for (int h = 0; h < c.h; ++h) decode_block(c, mx * c.h + h, my * c.v + v);
            }
            if (!(my == mcuy - 1 && mx == mcux - 1)) maybe_restart();
          }
      } else {  // a single component: its own blocks in raster order, only those covering real samples
        Component& c = comps[(size_t)order[0]];
        const int bw = (c.width + 7) / 8, bh = (c.height + 7) / 8;
        for (int by = 0; by < bh; ++by)
          for (int bx = 0; bx < bw; ++bx) {
            decode_block(c, bx, by);
            if (!(by == bh - 1 && bx == bw - 1)) maybe_restart();
          }
      }
      // continue after the entropy-coded data: find the next marker that is not RSTn / stuffed
      const uint8_t* q = br.p;
      while (q + 1 < buf + n && !(q[0] == 0xFF && q[1] != 0 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
      pos = (size_t)(q - buf);
      continue;
    }
    pos += (size_t)len;
  }
  if (!have_frame || comps.empty() || comps[0].plane.empty()) throw std::runtime_error("JPEG: no image data");

  const Component& Y = comps[0];
  const int ys = Y.blocks_w * 8;
  if (comps.size() == 1 || !want_color) {
    // luma must be at full resolution (it is in every YCbCr JPEG: h = hmax, v = vmax)
    if (Y.h != hmax || Y.v != vmax) throw std::runtime_error("JPEG: subsampled first component is not supported");
    if (!gray) throw std::runtime_error("DecodeJpeg: no gray output given");
    gray->create(height, width);
    for (int y = 0; y < height; ++y) std::memcpy(gray->ptr(y), &Y.plane[(size_t)y * ys], (size_t)width);
    return 1;
  }
  if (!color) throw std::runtime_error("DecodeJpeg: no colour output given");
  if (Y.h != hmax || Y.v != vmax) throw std::runtime_error("JPEG: subsampled first component is not supported");
  std::vector<uint8_t> cb, cr;
  int cbw, cbh, crw, crh;
  Upsample(comps[1], hmax / comps[1].h, vmax / comps[1].v, cb, cbw, cbh);
  Upsample(comps[2], hmax / comps[2].h, vmax / comps[2].v, cr, crw, crh);
  if (cbw < width || crw < width || cbh < height || crh < height) throw std::runtime_error("JPEG: chroma planes too small");
  // YCbCr -> RGB with the decoder's 16-bit fixed-point tables
  int cr_r[256], cb_b[256];
  long cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) {
    const long x = i - 128;
    cr_r[i] = (int)((91881L * x + 32768L) >> 16);            // 1.40200
    cb_b[i] = (int)((116130L * x + 32768L) >> 16);           // 1.77200
    cr_g[i] = -46802L * x;                                    // 0.71414
    cb_g[i] = -22554L * x + 32768L;                           // 0.34414 (+ rounding)
  }
  color->create(height, width);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const int yy = Y.plane[(size_t)y * ys + x];
      const int b = cb[(size_t)y * cbw + x], r = cr[(size_t)y * crw + x];
      Vec3b& o = color->at(y, x);
      o.v[2] = Clamp(yy + cr_r[r]);
      o.v[1] = Clamp(yy + (int)((cb_g[b] + cr_g[r]) >> 16));
      o.v[0] = Clamp(yy + cb_b[b]);
    }
  return 3;
}

}  // namespace core
}  // namespace bm
