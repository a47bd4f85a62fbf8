/*
 * synth.cpp — deterministic synthetic RGB-D + label stream: analytic ray cast of an
 * axis-aligned room (8 x 6 x 3 m, plane labels 1..6), 12 static primitives (spheres / boxes, labels
 * 7..18) placed by mt19937(seed) and one moving sphere (r = 0.3 m, label 19, 0.5 m/s).
 * Pinhole camera, depth = z-depth in metres (0 = invalid / beyond max_depth).
 * Used by tests/ and bench.py to feed both the HIP path and the oracle the same frames.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

struct Prim {
  int type;  // 0 sphere, 1 box
  float c[3];
  float h[3];  // sphere: h[0] = radius; box: half extents
  int label;
  float vel[3];
};

struct Scene {
  float room_min[3], room_max[3];
  std::vector<Prim> prims;
};

inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

const uint8_t kPalette[20][3] = {
    {0, 0, 0},       {200, 200, 200}, {180, 180, 200}, {200, 180, 180}, {180, 200, 180}, {120, 110, 100},
    {230, 230, 230}, {230, 25, 75},   {60, 180, 75},   {255, 225, 25},  {0, 130, 200},   {245, 130, 48},
    {145, 30, 180},  {70, 240, 240},  {240, 50, 230},  {210, 245, 60},  {250, 190, 212}, {0, 128, 128},
    {220, 190, 255}, {170, 110, 40}};

}  // namespace

extern "C" {

struct synth_scene;

void* synth_create(uint32_t seed, int num_static, int with_mover) {
  auto* s = new Scene();
  s->room_min[0] = -4.f; s->room_min[1] = -3.f; s->room_min[2] = 0.f;
  s->room_max[0] = 4.f;  s->room_max[1] = 3.f;  s->room_max[2] = 3.f;
  std::mt19937 rng(seed);
  auto uni = [&](float a, float b) {
    // mt19937 -> float in [a,b) without relying on std::uniform_real_distribution (impl-defined)
    return a + (b - a) * (static_cast<float>(rng() >> 8) * (1.0f / 16777216.0f));
  };
  for (int i = 0; i < num_static; ++i) {
    Prim p{};
    p.type = i % 2;
    p.label = 7 + (i % 12);
    if (p.type == 0) {
      p.h[0] = uni(0.15f, 0.5f);
      p.c[0] = uni(-3.4f, 3.4f);
      p.c[1] = uni(-2.4f, 2.4f);
      p.c[2] = uni(p.h[0], 2.2f);
    } else {
      p.h[0] = uni(0.15f, 0.5f); p.h[1] = uni(0.15f, 0.5f); p.h[2] = uni(0.15f, 0.6f);
      p.c[0] = uni(-3.4f, 3.4f);
      p.c[1] = uni(-2.4f, 2.4f);
      p.c[2] = p.h[2];  // boxes stand on the floor
    }
    // keep the camera circle (r = 1.5 m at z = 1.5) clear
    const float rr = std::sqrt(p.c[0] * p.c[0] + p.c[1] * p.c[1]);
    if (std::fabs(rr - 1.5f) < 0.8f) {
      const float sc = (rr < 1.5f ? 0.4f : 2.6f) / std::max(rr, 1e-3f);
      p.c[0] *= sc; p.c[1] *= sc;
      p.c[0] = std::min(3.4f, std::max(-3.4f, p.c[0]));
      p.c[1] = std::min(2.4f, std::max(-2.4f, p.c[1]));
    }
    s->prims.push_back(p);
  }
  if (with_mover) {
    Prim p{};
    p.type = 0; p.label = 19; p.h[0] = 0.3f;
    p.c[0] = -2.0f; p.c[1] = 2.2f; p.c[2] = 1.0f;
    p.vel[0] = 0.5f; p.vel[1] = 0.f; p.vel[2] = 0.f;
    s->prims.push_back(p);
  }
  return s;
}

void synth_destroy(void* s) { delete static_cast<Scene*>(s); }

/* override the mover (last primitive if created with_mover) start position / velocity */
void synth_set_mover(void* sp, const float* pos, const float* vel, float radius) {
  auto* s = static_cast<Scene*>(sp);
  if (s->prims.empty() || s->prims.back().label != 19) return;
  Prim& p = s->prims.back();
  for (int i = 0; i < 3; ++i) { p.c[i] = pos[i]; p.vel[i] = vel[i]; }
  p.h[0] = radius;
}

/* Render one frame. world_T_sensor: row-major 4x4 double. noise_sigma_rel: depth noise as a fraction
 * of z (0 = exact). */
void synth_render(void* sp, int W, int H, float fx, float fy, float cx, float cy, const double* T,
                  double t_sec, float max_depth, float noise_sigma_rel, uint32_t noise_seed, float* depth,
                  uint8_t* rgb, int32_t* label, int num_threads) {
  const Scene& s = *static_cast<Scene*>(sp);
  std::vector<Prim> prims = s.prims;
  for (auto& p : prims)
    for (int i = 0; i < 3; ++i) p.c[i] += p.vel[i] * static_cast<float>(t_sec);
  float R[9], o[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) R[3 * r + c] = static_cast<float>(T[4 * r + c]);
    o[r] = static_cast<float>(T[4 * r + 3]);
  }
  auto rows = [&](int v0, int v1) {
    for (int v = v0; v < v1; ++v) {
      for (int u = 0; u < W; ++u) {
        const float xc = (static_cast<float>(u) - cx) / fx, yc = (static_cast<float>(v) - cy) / fy;
        float d[3];
        for (int r = 0; r < 3; ++r) d[r] = R[3 * r] * xc + R[3 * r + 1] * yc + R[3 * r + 2];
        float best = 1e30f;
        int lab = 0;
        // room (inside-out box): exit distance per axis
        for (int ax = 0; ax < 3; ++ax) {
          if (d[ax] > 1e-9f) {
            const float sdist = (s.room_max[ax] - o[ax]) / d[ax];
            if (sdist > 1e-4f && sdist < best) { best = sdist; lab = 1 + 2 * ax; }
          } else if (d[ax] < -1e-9f) {
            const float sdist = (s.room_min[ax] - o[ax]) / d[ax];
            if (sdist > 1e-4f && sdist < best) { best = sdist; lab = 2 + 2 * ax; }
          }
        }
        for (const Prim& p : prims) {
          if (p.type == 0) {
            const float oc[3] = {o[0] - p.c[0], o[1] - p.c[1], o[2] - p.c[2]};
            const float a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const float b = oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2];
            const float c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - p.h[0] * p.h[0];
            const float disc = b * b - a * c;
            if (disc < 0.f) continue;
            const float sq = std::sqrt(disc);
            float sdist = (-b - sq) / a;
            if (sdist <= 1e-4f) sdist = (-b + sq) / a;
            if (sdist > 1e-4f && sdist < best) { best = sdist; lab = p.label; }
          } else {
            float t0 = -1e30f, t1 = 1e30f;
            bool miss = false;
            for (int ax = 0; ax < 3 && !miss; ++ax) {
              const float lo = p.c[ax] - p.h[ax], hi = p.c[ax] + p.h[ax];
              if (std::fabs(d[ax]) < 1e-9f) {
                if (o[ax] < lo || o[ax] > hi) miss = true;
              } else {
                float a = (lo - o[ax]) / d[ax], b = (hi - o[ax]) / d[ax];
                if (a > b) std::swap(a, b);
                t0 = std::max(t0, a);
                t1 = std::min(t1, b);
                if (t0 > t1) miss = true;
              }
            }
            if (miss) continue;
            const float sdist = t0 > 1e-4f ? t0 : t1;
            if (sdist > 1e-4f && sdist < best) { best = sdist; lab = p.label; }
          }
        }
        const int i = v * W + u;
        float z = best;  // direction has unit z in the camera frame => parameter == z-depth
        if (noise_sigma_rel > 0.f) {
          const uint32_t h1 = hash32(noise_seed * 0x9e3779b9u + static_cast<uint32_t>(i));
          const uint32_t h2 = hash32(h1 ^ 0x68bc21ebu);
          const float u1 = (static_cast<float>(h1 >> 8) + 1.f) * (1.0f / 16777217.0f);
          const float u2 = static_cast<float>(h2 >> 8) * (1.0f / 16777216.0f);
          const float g = std::sqrt(-2.f * std::log(u1)) * std::cos(6.2831853f * u2);
          z = z * (1.f + noise_sigma_rel * g);
        }
        if (!(z > 0.f) || z > max_depth) z = 0.f;
        depth[i] = z;
        if (label) label[i] = lab;
        if (rgb) {
          const uint8_t* c = kPalette[lab % 20];
          // light deterministic texture so that colour fusion is exercised
          const int tex = ((u / 8) ^ (v / 8)) & 1 ? 0 : 12;
          rgb[3 * i] = static_cast<uint8_t>(std::max(0, c[0] - tex));
          rgb[3 * i + 1] = static_cast<uint8_t>(std::max(0, c[1] - tex));
          rgb[3 * i + 2] = static_cast<uint8_t>(std::max(0, c[2] - tex));
        }
      }
    }
  };
  if (num_threads < 1) num_threads = std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::thread> th;
  const int step = (H + num_threads - 1) / num_threads;
  for (int t = 0; t < num_threads; ++t) {
    const int v0 = t * step, v1 = std::min(H, v0 + step);
    if (v0 < v1) th.emplace_back(rows, v0, v1);
  }
  for (auto& t : th) t.join();
}

}  // extern "C"
