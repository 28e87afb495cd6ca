// Height-map sources next to rsb_set_heightmap (host side only, no GPU): PNG files, Perlin-noise terrains from a
// raisim::TerrainProperties-shaped record, and a plain text format.  Upstream counterparts [RECALL, absent from
// /root/reference]: World::addHeightMap(pngFile, centerX, centerY, xSize, ySize, heightScale, heightOffset),
// World::addHeightMap(centerX, centerY, TerrainProperties&), World::addHeightMap(raisimHeightMapFile, cx, cy)
// (raisim/World.hpp, raisim/object/terrain/HeightMap.hpp).  The noise function and the PNG scaling convention are this
// repo's own (documented in include/rsb.h); RaiSim's exact ones cannot be read here.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "rsb.h"
#include "rsb_internal.h"

namespace {

struct Png { int w = 0, h = 0, depth = 0, channels = 0; std::vector<uint8_t> rows; };   // rows: unfiltered, h x stride

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Non-interlaced 8/16-bit grey, grey+alpha, RGB, RGBA.  Returns an error string or "".
std::string read_png(const char* path, bool header_only, Png& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return std::string("cannot open ") + path;
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 + 25 || std::memcmp(buf.data(), sig, 8) != 0) return "not a PNG file";
  std::vector<uint8_t> idat;
  size_t pos = 8;
  bool have_hdr = false;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(&buf[pos]);
    const char* type = reinterpret_cast<const char*>(&buf[pos + 4]);
    if (pos + 12 + (size_t)len > buf.size()) return "truncated PNG chunk";
    const uint8_t* data = &buf[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) return "bad IHDR";
      out.w = (int)be32(data); out.h = (int)be32(data + 4); out.depth = data[8];
      const int ct = data[9];
      if (data[12] != 0) return "interlaced PNGs are not supported";
      if (out.depth != 8 && out.depth != 16) return "only 8- and 16-bit PNGs are supported";
      out.channels = ct == 0 ? 1 : ct == 2 ? 3 : ct == 4 ? 2 : ct == 6 ? 4 : 0;
      if (!out.channels) return "palette PNGs are not supported";
      if (out.w < 2 || out.h < 2 || out.w > 16384 || out.h > 16384) return "PNG size out of range (2..16384)";
      have_hdr = true;
      if (header_only) return "";
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_hdr) return "PNG without IHDR";
  const int bpp = out.channels * out.depth / 8;
  const size_t stride = (size_t)out.w * bpp;
  std::vector<uint8_t> raw((stride + 1) * out.h);
  uLongf dl = (uLongf)raw.size();
  if (uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size()) != Z_OK || dl != raw.size()) return "PNG data does not inflate to the image size";
  out.rows.assign(stride * out.h, 0);
  for (int y = 0; y < out.h; ++y) {
    const uint8_t* src = &raw[(stride + 1) * y];
    uint8_t* cur = &out.rows[stride * y];
    const uint8_t* up = y ? cur - stride : nullptr;
    const int ft = src[0];
    if (ft > 4) return "bad PNG filter type";
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
      const int x = src[1 + i];
      const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : paeth(a, b, c);
      cur[i] = (uint8_t)(x + pred);
    }
  }
  return "";
}

// Improved Perlin noise (Perlin 2002) in 2-D with a seeded permutation; value range about [-1, 1].
struct Perlin {
  int p[512];
  explicit Perlin(uint32_t seed) {
    std::vector<int> v(256);
    std::iota(v.begin(), v.end(), 0);
    std::mt19937 g(seed);
    for (int i = 255; i > 0; --i) { const int j = (int)(g() % (uint32_t)(i + 1)); std::swap(v[i], v[j]); }   // own shuffle: std::shuffle is not portable
    for (int i = 0; i < 512; ++i) p[i] = v[i & 255];
  }
  static double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
  static double grad(int h, double x, double y) {
    switch (h & 7) {
      case 0: return x + y; case 1: return x - y; case 2: return -x + y; case 3: return -x - y;
      case 4: return x; case 5: return -x; case 6: return y; default: return -y;
    }
  }
  double operator()(double x, double y) const {
    const double fx = std::floor(x), fy = std::floor(y);
    // lattice cell modulo 256 in floating point first: (int) of a double beyond INT_MAX is undefined behaviour, and
    // frequency * lacunarity^octave grows without bound
    const int X = (int)(fx - 256.0 * std::floor(fx / 256.0)) & 255, Y = (int)(fy - 256.0 * std::floor(fy / 256.0)) & 255;
    x -= fx; y -= fy;
    const double u = fade(x), v = fade(y);
    const int A = p[X] + Y, B = p[X + 1] + Y;
    const double l0 = grad(p[A], x, y) + u * (grad(p[B], x - 1, y) - grad(p[A], x, y));
    const double l1 = grad(p[A + 1], x, y - 1) + u * (grad(p[B + 1], x - 1, y - 1) - grad(p[A + 1], x, y - 1));
    return l0 + v * (l1 - l0);
  }
};

constexpr int kMaxSamples = 16384;   // per axis, for every height-map source (an untrusted file must not size an allocation freely)

}  // namespace

extern "C" {

int rsb_heightmap_png_size(const char* path, int* x_samples, int* y_samples) {
  if (!path || !x_samples || !y_samples) return RSB_E_INVALID;
  Png png;
  const std::string err = read_png(path, true, png);
  if (!err.empty()) { rsb::set_error("rsb_heightmap_png_size: " + err); return RSB_E_PARSE; }
  *x_samples = png.w; *y_samples = png.h;
  return RSB_OK;
}

int rsb_heightmap_png_read(const char* path, double height_scale, double height_offset, float* heights, int n) {
  if (!path || !heights) return RSB_E_INVALID;
  Png png;
  const std::string err = read_png(path, false, png);
  if (!err.empty()) { rsb::set_error("rsb_heightmap_png_read: " + err); return RSB_E_PARSE; }
  if ((long long)n != (long long)png.w * png.h) { rsb::set_error("rsb_heightmap_png_read: buffer size != x_samples * y_samples"); return RSB_E_INVALID; }
  const int bps = png.depth / 8, bpp = png.channels * bps;
  const double inv = 1.0 / (png.depth == 8 ? 255.0 : 65535.0);
  for (int y = 0; y < png.h; ++y)
    for (int x = 0; x < png.w; ++x) {
      const uint8_t* px = &png.rows[((size_t)y * png.w + x) * bpp];     // first channel (grey or red)
      const double v = bps == 1 ? px[0] : (double)((px[0] << 8) | px[1]);
      heights[(size_t)y * png.w + x] = (float)(v * inv * height_scale + height_offset);
    }
  return RSB_OK;
}

int rsb_heightmap_perlin(const rsb_terrain_properties* tp, float* heights) {
  if (!tp || !heights || tp->x_samples < 2 || tp->y_samples < 2 || tp->x_samples > kMaxSamples || tp->y_samples > kMaxSamples ||
      tp->fractal_octaves < 1 || tp->fractal_octaves > 32 || !(tp->x_size > 0) || !(tp->y_size > 0) ||
      !std::isfinite(tp->frequency) || !std::isfinite(tp->fractal_lacunarity) || !std::isfinite(tp->fractal_gain)) {
    rsb::set_error("rsb_heightmap_perlin: bad terrain properties");
    return RSB_E_INVALID;
  }
  const Perlin noise(tp->seed);
  const double dx = tp->x_size / (tp->x_samples - 1), dy = tp->y_size / (tp->y_samples - 1);
  for (int iy = 0; iy < tp->y_samples; ++iy)
    for (int ix = 0; ix < tp->x_samples; ++ix) {
      double f = tp->frequency, amp = 1.0, h = 0.0;
      for (int o = 0; o < tp->fractal_octaves; ++o) {
        h += amp * noise(ix * dx * f + 0.5 * o, iy * dy * f + 0.25 * o);
        f *= tp->fractal_lacunarity; amp *= tp->fractal_gain;
      }
      h *= tp->z_scale;
      if (tp->step_size > 0) h = std::round(h / tp->step_size) * tp->step_size;
      heights[(size_t)iy * tp->x_samples + ix] = (float)(h + tp->height_offset);
    }
  return RSB_OK;
}

int rsb_heightmap_text_size(const char* path, int* x_samples, int* y_samples, double* x_size, double* y_size) {
  if (!path || !x_samples || !y_samples || !x_size || !y_size) return RSB_E_INVALID;
  FILE* f = std::fopen(path, "r");
  if (!f) { rsb::set_error(std::string("rsb_heightmap_text_size: cannot open ") + path); return RSB_E_PARSE; }
  const int got = std::fscanf(f, "%d %d %lf %lf", x_samples, y_samples, x_size, y_size);
  std::fclose(f);
  if (got != 4 || *x_samples < 2 || *y_samples < 2 || *x_samples > kMaxSamples || *y_samples > kMaxSamples) { rsb::set_error("rsb_heightmap_text_size: header must be 'xSamples ySamples xSize ySize' with 2..16384 samples per axis"); return RSB_E_PARSE; }
  return RSB_OK;
}

int rsb_heightmap_text_read(const char* path, float* heights, int n) {
  if (!path || !heights) return RSB_E_INVALID;
  FILE* f = std::fopen(path, "r");
  if (!f) { rsb::set_error(std::string("rsb_heightmap_text_read: cannot open ") + path); return RSB_E_PARSE; }
  int xs = 0, ys = 0; double sx = 0, sy = 0;
  int st = RSB_OK;
  if (std::fscanf(f, "%d %d %lf %lf", &xs, &ys, &sx, &sy) != 4 || xs < 2 || ys < 2 || xs > kMaxSamples || ys > kMaxSamples ||
      (long long)xs * ys != (long long)n) st = RSB_E_PARSE;
  for (int i = 0; st == RSB_OK && i < n; ++i) {
    double v;
    if (std::fscanf(f, "%lf", &v) != 1) st = RSB_E_PARSE; else heights[i] = (float)v;
  }
  std::fclose(f);
  if (st != RSB_OK) rsb::set_error("rsb_heightmap_text_read: header / sample count mismatch");
  return st;
}

}  // extern "C"
