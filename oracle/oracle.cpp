/*
 * oracle.cpp — CPU restatement of the Khronos active-window volumetric fusion path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY: the in-repo half is pinned against the reference's own code
 * (oracle/_ref, tests/test_cpu_ref_pin.py), the upstream half (integrator, mesh) is UNPINNED (see oracle.h).
 *
 * Structure deliberately mirrors the reference CPU path: an unordered_map from block index to
 * heap-allocated array-of-struct voxel blocks, std::thread workers pulling block indices from an
 * atomic cursor (hydra::IndexGetter usage at tracking_integrator.cpp:83-90), scalar per-voxel code.
 * Build with -ffp-contract=off so that float results are IEEE-reproducible.
 */
#include "oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

#include "mc_table.inc"  // kMcTriTable[256][16], Lorensen/Bourke numbering (ASSUMPTIONS.md A.5)

struct I3 {
  int32_t x, y, z;
  bool operator==(const I3& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator<(const I3& o) const {
    return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z);
  }
};
struct L3 {
  int64_t x, y, z;
  bool operator==(const L3& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator<(const L3& o) const {
    return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z);
  }
};
struct I3Hash {
  // voxblox-lineage block hash (ASSUMPTIONS.md A.1); container-internal only.
  size_t operator()(const I3& i) const {
    return static_cast<size_t>(static_cast<int64_t>(i.x) + 17191ll * i.y + 17191ll * 17191ll * i.z);
  }
};
struct L3Hash {
  size_t operator()(const L3& i) const {
    return static_cast<size_t>(i.x + 17191ll * i.y + 17191ll * 17191ll * i.z);
  }
};

// hydra voxel_types.h field sets (uses: tracking_integrator.cpp:146,164,231-251;
// mesh_object_extractor.cpp:250-261,344-355)
struct TsdfVoxel {
  float distance = 0.f;
  float weight = 0.f;
  uint8_t r = 0, g = 0, b = 0, a = 0;
};
struct TrackingVoxel {
  uint64_t last_observed = 0;
  uint64_t last_occupied = 0;
  bool ever_free = false;
  bool active = false;
  bool to_remove = false;
};
struct SemanticVoxel {
  uint32_t semantic_label = 0;
  bool empty = true;
};
struct MeshBlock {
  std::vector<float> points;  // 3 per vertex
  std::vector<uint8_t> colors;  // 4 per vertex
  std::vector<uint32_t> labels;
  std::vector<uint64_t> first_seen;
  std::vector<uint64_t> stamps;
};

struct Block {
  I3 index;
  std::vector<TsdfVoxel> tsdf;
  std::vector<TrackingVoxel> tracking;
  std::vector<SemanticVoxel> semantic;
  std::vector<float> likelihoods;  // voxel-major: [voxel][k]
  bool updated = false, mesh_updated = false, tracking_updated = false;
  bool has_active_data = false;
  MeshBlock mesh;
};

inline double toSeconds(uint64_t ns) { return static_cast<double>(ns) / 1e9; }

inline uint32_t mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
// Owner of a block when the map is sharded by contiguous hash range.
inline int ownerOf(const I3& b, int world) {
  if (world <= 1) return 0;
  uint32_t h = mix32(static_cast<uint32_t>(b.x) * 73856093u ^ mix32(static_cast<uint32_t>(b.y) * 19349663u ^
                                                                    mix32(static_cast<uint32_t>(b.z) * 83492791u)));
  return static_cast<int>((static_cast<uint64_t>(h) * static_cast<uint64_t>(world)) >> 32);
}

struct Pose {
  float R[9];  // sensor_T_world rotation, row-major
  float t[3];
  float Rw[9];  // world_T_sensor
  float tw[3];
};

Pose makePose(const double* T) {
  Pose p;
  // world_T_sensor as given (double) -> float
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) p.Rw[3 * r + c] = static_cast<float>(T[4 * r + c]);
    p.tw[r] = static_cast<float>(T[4 * r + 3]);
  }
  // inverse in double, then cast
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) p.R[3 * r + c] = static_cast<float>(T[4 * c + r]);
    const double v = -(T[4 * 0 + r] * T[3] + T[4 * 1 + r] * T[7] + T[4 * 2 + r] * T[11]);
    p.t[r] = static_cast<float>(v);
  }
  return p;
}

inline void xform(const float* R, const float* t, float x, float y, float z, float* o) {
  o[0] = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
  o[1] = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
  o[2] = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
}

struct Frustum {
  float n[4][3];  // inward normals: left, right, top, bottom
};

inline void crossn(const float* a, const float* b, float* o) {
  float x = a[1] * b[2] - a[2] * b[1];
  float y = a[2] * b[0] - a[0] * b[2];
  float z = a[0] * b[1] - a[1] * b[0];
  float n = std::sqrt((x * x + y * y) + z * z);
  o[0] = x / n;
  o[1] = y / n;
  o[2] = z / n;
}

Frustum makeFrustum(const orc_sensor& s) {
  const float xl = (0.f - s.cx) / s.fx, xr = (static_cast<float>(s.width) - s.cx) / s.fx;
  const float yt = (0.f - s.cy) / s.fy, yb = (static_cast<float>(s.height) - s.cy) / s.fy;
  const float tl[3] = {xl, yt, 1.f}, tr[3] = {xr, yt, 1.f}, bl[3] = {xl, yb, 1.f}, br[3] = {xr, yb, 1.f};
  Frustum f;
  crossn(bl, tl, f.n[0]);
  crossn(tr, br, f.n[1]);
  crossn(tl, tr, f.n[2]);
  crossn(br, bl, f.n[3]);
  return f;
}

inline bool pointInFrustum(const Frustum& f, const float* p, float max_range, float infl) {
  if (p[2] < -infl) return false;
  const float n2 = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
  const float lim = max_range + infl;
  if (n2 > lim * lim) return false;
  for (int i = 0; i < 4; ++i) {
    const float d = (p[0] * f.n[i][0] + p[1] * f.n[i][1]) + p[2] * f.n[i][2];
    if (d < -infl) return false;
  }
  return true;
}

}  // namespace

struct orc_map {
  orc_config cfg;
  int vps, nvox;
  float bs, bs_inv, vs_inv;
  std::unordered_map<I3, std::unique_ptr<Block>, I3Hash> blocks;
  // free-or-ever-free bit masks of blocks owned by other ranks (multi-GPU halo)
  std::unordered_map<I3, std::vector<uint64_t>, I3Hash> halo;
  std::vector<Block*> ef_work;  // tracking-updated blocks collected by phase 1
  // mesh halo: the three low voxel planes of blocks owned by other ranks (record layout: khronos_amd.h)
  std::unordered_map<I3, std::vector<uint32_t>, I3Hash> mesh_halo;
  // the pixel LISTS (with the reference's duplicates) of the clusters the latest motion detection kept, in id order
  std::vector<std::vector<int32_t>> last_md_pixels;
  // ProjectiveIntegrator::computeLabel as a callback (orc_set_label_hook)
  orc_label_hook_fn label_hook = nullptr;
  void* label_user = nullptr;

  Block* find(const I3& i) const {
    auto it = blocks.find(i);
    return it == blocks.end() ? nullptr : it->second.get();
  }
  Block* allocate(const I3& i, bool* created = nullptr) {
    auto it = blocks.find(i);
    if (it != blocks.end()) {
      if (created) *created = false;
      return it->second.get();
    }
    auto b = std::make_unique<Block>();
    b->index = i;
    b->tsdf.resize(nvox);
    if (cfg.with_tracking) b->tracking.resize(nvox);
    if (cfg.with_semantics) {
      b->semantic.resize(nvox);
      b->likelihoods.assign(static_cast<size_t>(nvox) * cfg.num_labels, 0.f);
    }
    Block* p = b.get();
    blocks.emplace(i, std::move(b));
    if (created) *created = true;
    return p;
  }
  std::vector<I3> sortedIndices() const {
    std::vector<I3> v;
    v.reserve(blocks.size());
    for (auto& kv : blocks) v.push_back(kv.first);
    std::sort(v.begin(), v.end());
    return v;
  }
};

namespace {

template <typename F>
void parallelFor(int num_threads, size_t n, F&& fn) {
  std::atomic<size_t> cursor{0};
  auto worker = [&]() {
    while (true) {
      size_t i = cursor.fetch_add(1);
      if (i >= n) break;
      fn(i);
    }
  };
  if (num_threads <= 1 || n <= 1) {
    worker();
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < num_threads; ++t) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

inline int32_t floorDivI64(int64_t a, int64_t b) {
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return static_cast<int32_t>(q);
}

// ---- interpolation (ASSUMPTIONS.md A.3 "interpolators") -------------------------------------
struct InterpW {
  int u[4], v[4];
  float w[4];
  int best;  // index of the max-weight pixel (first max)
};

inline bool computeInterp(const orc_config& c, const orc_sensor& s, const float* range, float u, float v,
                          InterpW* o) {
  const int u0 = static_cast<int>(std::floor(u)), v0 = static_cast<int>(std::floor(v));
  const int u1 = std::min(u0 + 1, s.width - 1), v1 = std::min(v0 + 1, s.height - 1);
  const float du = u - static_cast<float>(u0), dv = v - static_cast<float>(v0);
  o->u[0] = u0; o->v[0] = v0;
  o->u[1] = u0; o->v[1] = v1;
  o->u[2] = u1; o->v[2] = v0;
  o->u[3] = u1; o->v[3] = v1;
  const int nearest = (du >= 0.5f ? 2 : 0) + (dv >= 0.5f ? 1 : 0);
  bool use_nearest = c.interpolation_method == 0;
  if (c.interpolation_method == 2) {
    float mn = range[o->v[0] * s.width + o->u[0]], mx = mn;
    for (int i = 1; i < 4; ++i) {
      const float r = range[o->v[i] * s.width + o->u[i]];
      mn = std::min(mn, r);
      mx = std::max(mx, r);
    }
    if (mx - mn > c.adaptive_max_range_difference) use_nearest = true;
  }
  if (use_nearest) {
    for (int i = 0; i < 4; ++i) o->w[i] = 0.f;
    o->w[nearest] = 1.f;
    o->best = nearest;
  } else {
    o->w[0] = (1.f - du) * (1.f - dv);
    o->w[1] = (1.f - du) * dv;
    o->w[2] = du * (1.f - dv);
    o->w[3] = du * dv;
    int best = 0;
    for (int i = 1; i < 4; ++i)
      if (o->w[i] > o->w[best]) best = i;
    o->best = best;
  }
  return true;
}

inline uint8_t toU8(float f) {
  float r = std::floor(f + 0.5f);
  r = std::min(255.f, std::max(0.f, r));
  return static_cast<uint8_t>(r);
}

struct IntegrateCtx {
  orc_map* m;
  const orc_sensor* s;
  const orc_frame* f;
  Pose pose;
  std::vector<float> range;
  float log_match, log_nomatch;
  std::atomic<uint64_t> n_upd{0}, n_band{0};
};

// per-block update: restates the upstream per-voxel loop of hydra::ProjectiveIntegrator
// (call active_window.cpp:210; hook contract object_integrator.cpp:58-81; ASSUMPTIONS.md A.3)
void integrateBlock(IntegrateCtx& ctx, Block& blk) {
  orc_map& m = *ctx.m;
  const orc_config& c = m.cfg;
  const orc_sensor& s = *ctx.s;
  const orc_frame& f = *ctx.f;
  const int vps = m.vps;
  const float vs = c.voxel_size, trunc = c.truncation_distance;
  const float ox = static_cast<float>(blk.index.x) * m.bs, oy = static_cast<float>(blk.index.y) * m.bs,
              oz = static_cast<float>(blk.index.z) * m.bs;
  const float eps = c.weight_dropoff_epsilon > 0.f ? c.weight_dropoff_epsilon : c.weight_dropoff_epsilon * -vs;
  uint64_t n_upd = 0, n_band = 0;
  bool any = false;
  for (int iz = 0; iz < vps; ++iz) {
    for (int iy = 0; iy < vps; ++iy) {
      for (int ix = 0; ix < vps; ++ix) {
        const int lin = ix + vps * (iy + vps * iz);
        const float px = ox + (static_cast<float>(ix) + 0.5f) * vs;
        const float py = oy + (static_cast<float>(iy) + 0.5f) * vs;
        const float pz = oz + (static_cast<float>(iz) + 0.5f) * vs;
        float pc[3];
        xform(ctx.pose.R, ctx.pose.t, px, py, pz, pc);
        if (pc[2] <= 0.f) continue;
        const float voxel_range =
            c.range_mode == 0 ? pc[2] : std::sqrt((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
        if (voxel_range < s.min_range || voxel_range > s.max_range) continue;
        const float u = (pc[0] * s.fx) / pc[2] + s.cx;
        if (std::ceil(u) >= static_cast<float>(s.width) || std::floor(u) < 0.f) continue;
        const float v = (pc[1] * s.fy) / pc[2] + s.cy;
        if (std::ceil(v) >= static_cast<float>(s.height) || std::floor(v) < 0.f) continue;
        InterpW iw;
        computeInterp(c, s, ctx.range.data(), u, v, &iw);
        float r4[4];
        for (int i = 0; i < 4; ++i) r4[i] = ctx.range[iw.v[i] * s.width + iw.u[i]];
        const float dist_surface = ((iw.w[0] * r4[0] + iw.w[1] * r4[1]) + iw.w[2] * r4[2]) + iw.w[3] * r4[3];
        if (!(dist_surface >= s.min_range) || dist_surface > s.max_range) continue;
        const float sdf = dist_surface - voxel_range;
        if (sdf < -trunc) continue;
        const bool in_band = std::fabs(sdf) < trunc;
        const int best_px = iw.v[iw.best] * s.width + iw.u[iw.best];
        int label = -1;
        bool have_label = false;
        if (m.label_hook) {
          // the subclass hook decides mask and label (object_integrator.cpp:58-81, here the reference's own code: ref_harness.cpp)
          int32_t lab = -1;
          if (!m.label_hook(m.label_user, sdf, iw.u, iw.v, iw.w, &lab)) continue;
          if (in_band) {
            label = lab;
            have_label = lab >= 0;
          }
        } else if (in_band) {
          if (f.mask && f.mask[best_px] != 0) continue;  // object_integrator.cpp:70-73
          if (c.semantic_mode == 1) {
            if (f.object_image) {
              label = (f.object_image[best_px] == f.object_id) ? 1 : 0;  // object_integrator.cpp:77-79
              have_label = true;
            }
          } else if (f.label) {
            label = f.label[best_px];
            have_label = true;
          }
        }
        // weight (ASSUMPTIONS.md A.3 computeWeight)
        const float q = vs / pc[2];
        float w = (s.fx * s.fy) * (q * q);
        if (!c.use_constant_weight) w = w / (pc[2] * pc[2]);
        if (c.use_weight_dropoff && sdf < -eps) {
          w = w * ((trunc + sdf) / (trunc - eps));
          w = std::max(w, 0.f);
        }
        if (!(w > 0.f)) continue;

        // updateVoxel
        TsdfVoxel& tv = blk.tsdf[lin];
        const float sdf_c = std::max(std::min(trunc, sdf), -trunc);
        const float w_before = tv.weight;
        tv.distance = (tv.distance * tv.weight + sdf_c * w) / (tv.weight + w);
        tv.weight = std::min(tv.weight + w, c.max_weight);
        // [A] color_blend_weight: the blend below uses the voxel weight after (0, panoptic-lineage order) or before (1) this update
        const float w_blend = c.color_blend_weight ? w_before : tv.weight;
        if (c.with_tracking) blk.tracking[lin].last_observed = f.timestamp_ns;
        ++n_upd;
        any = true;
        if (in_band) {
          ++n_band;
          if (f.color) {
            float col[3];
            for (int ch = 0; ch < 3; ++ch) {
              float a = 0.f;
              for (int i = 0; i < 4; ++i)
                a = a + iw.w[i] * static_cast<float>(f.color[3 * (iw.v[i] * s.width + iw.u[i]) + ch]);
              col[ch] = static_cast<float>(toU8(a));
            }
            const float tot = w_blend + w;
            tv.r = toU8((static_cast<float>(tv.r) * w_blend + col[0] * w) / tot);
            tv.g = toU8((static_cast<float>(tv.g) * w_blend + col[1] * w) / tot);
            tv.b = toU8((static_cast<float>(tv.b) * w_blend + col[2] * w) / tot);
            tv.a = 255;
          }
          if (c.with_semantics && have_label && label >= 0 && label < c.num_labels) {
            SemanticVoxel& sv = blk.semantic[lin];
            float* l = &blk.likelihoods[static_cast<size_t>(lin) * c.num_labels];
            if (sv.empty) {
              for (int k = 0; k < c.num_labels; ++k) l[k] = 0.f;
              sv.empty = false;
            }
            if (c.semantic_mode == 1) {
              l[label] += 1.f;
            } else {
              for (int k = 0; k < c.num_labels; ++k) l[k] += (k == label ? ctx.log_match : ctx.log_nomatch);
            }
            int best = 0;
            for (int k = 1; k < c.num_labels; ++k)
              if (l[k] > l[best]) best = k;
            sv.semantic_label = static_cast<uint32_t>(best);
          }
        }
      }
    }
  }
  if (any) {
    blk.updated = true;
    blk.mesh_updated = true;
    blk.tracking_updated = true;
  }
  ctx.n_upd += n_upd;
  ctx.n_band += n_band;
}

std::vector<float> makeRange(const orc_config& c, const orc_sensor& s, const float* depth) {
  std::vector<float> range(static_cast<size_t>(s.width) * s.height);
  for (int v = 0; v < s.height; ++v) {
    for (int u = 0; u < s.width; ++u) {
      const float d = depth[v * s.width + u];
      float r = 0.f;
      if (d > 0.f && std::isfinite(d)) {
        if (c.range_mode == 0) {
          r = d;
        } else {
          const float x = (static_cast<float>(u) - s.cx) / s.fx, y = (static_cast<float>(v) - s.cy) / s.fy;
          r = d * std::sqrt((x * x + y * y) + 1.f);
        }
      }
      range[v * s.width + u] = r;
    }
  }
  return range;
}

const int kNeighborOffsets26[26][3] = {
    // 6 faces
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    // 12 edges
    {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 0, -1}, {-1, 0, 1}, {1, 0, -1}, {1, 0, 1},
    {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1},
    // 8 corners
    {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {1, 1, 1}};

inline bool voxelIsFree(const orc_config& c, const TrackingVoxel& v, uint64_t stamp) {
  // tracking_integrator.cpp:248-252
  return toSeconds(v.last_occupied) < toSeconds(stamp) - static_cast<double>(c.temporal_buffer) &&
         v.last_observed != 0u;
}

}  // namespace

extern "C" {

void orc_set_threads(orc_map* m, int32_t num_threads) { m->cfg.num_threads = num_threads; }

void orc_set_label_hook(orc_map* m, orc_label_hook_fn fn, void* user) {
  m->label_hook = fn;
  m->label_user = user;
}

orc_map* orc_create(const orc_config* cfg) {
  auto* m = new orc_map();
  m->cfg = *cfg;
  m->vps = cfg->voxels_per_side;
  m->nvox = m->vps * m->vps * m->vps;
  m->bs = cfg->voxel_size * static_cast<float>(cfg->voxels_per_side);
  m->bs_inv = 1.f / m->bs;
  m->vs_inv = 1.f / cfg->voxel_size;
  if (m->cfg.num_threads < 1) m->cfg.num_threads = std::max(1u, std::thread::hardware_concurrency());
  if (m->cfg.world_size < 1) m->cfg.world_size = 1;
  return m;
}

void orc_destroy(orc_map* m) { delete m; }

void orc_parse_input(const orc_config* cfg, const orc_sensor* s, const double* world_T_sensor,
                     const float* depth, float* range_out, float* vertex_out) {
  const Pose p = makePose(world_T_sensor);
  std::vector<float> range = makeRange(*cfg, *s, depth);
  for (int v = 0; v < s->height; ++v) {
    for (int u = 0; u < s->width; ++u) {
      const int i = v * s->width + u;
      const float d = depth[i];
      if (range_out) range_out[i] = range[i];
      if (!vertex_out) continue;
      if (range[i] <= 0.f) {
        vertex_out[3 * i] = vertex_out[3 * i + 1] = vertex_out[3 * i + 2] = 0.f;
        continue;
      }
      const float x = ((static_cast<float>(u) - s->cx) / s->fx) * d;
      const float y = ((static_cast<float>(v) - s->cy) / s->fy) * d;
      xform(p.Rw, p.tw, x, y, d, &vertex_out[3 * i]);
    }
  }
}

int orc_integrate(orc_map* m, const orc_sensor* s, const orc_frame* f, int allocate_blocks,
                  orc_stats* stats) {
  const orc_config& c = m->cfg;
  IntegrateCtx ctx;
  ctx.m = m;
  ctx.s = s;
  ctx.f = f;
  ctx.pose = makePose(f->world_T_sensor);
  ctx.range = makeRange(c, *s, f->depth);
  ctx.log_match = std::log(c.label_confidence);
  ctx.log_nomatch = c.num_labels > 1
                        ? std::log((1.f - c.label_confidence) / static_cast<float>(c.num_labels - 1))
                        : 0.f;

  std::vector<Block*> work;
  uint64_t n_new = 0;
  if (allocate_blocks) {
    // findBlocksInViewFrustum (ASSUMPTIONS.md A.3 allocation)
    const Frustum fr = makeFrustum(*s);
    const float infl = 0.8660254f * m->bs;
    const int n = static_cast<int>(std::ceil(s->max_range * m->bs_inv)) + 1;
    const I3 bc = {static_cast<int32_t>(std::floor(ctx.pose.tw[0] * m->bs_inv)),
                   static_cast<int32_t>(std::floor(ctx.pose.tw[1] * m->bs_inv)),
                   static_cast<int32_t>(std::floor(ctx.pose.tw[2] * m->bs_inv))};
    if (c.alloc_candidate == 1) {
      // [A] alloc_candidate = camera_offset (panoptic_mapping lineage, Camera::findVisibleBlocks): the candidate POINT
      // camera_W + offset * block_size, integer offsets up to floor((max_range + half diagonal) / block_size), is tested against
      // the inflated frustum, and the block that contains the point is allocated (an index set: two offsets may map to one block)
      const int ms = static_cast<int>(std::floor((s->max_range + infl) * m->bs_inv));
      std::set<std::array<int32_t, 3>> seen;
      for (int dz = -ms; dz <= ms; ++dz) {
        for (int dy = -ms; dy <= ms; ++dy) {
          for (int dx = -ms; dx <= ms; ++dx) {
            const float pxw = ctx.pose.tw[0] + static_cast<float>(dx) * m->bs;
            const float pyw = ctx.pose.tw[1] + static_cast<float>(dy) * m->bs;
            const float pzw = ctx.pose.tw[2] + static_cast<float>(dz) * m->bs;
            float pc[3];
            xform(ctx.pose.R, ctx.pose.t, pxw, pyw, pzw, pc);
            if (!pointInFrustum(fr, pc, s->max_range, infl)) continue;
            const I3 b = {static_cast<int32_t>(std::floor(pxw * m->bs_inv)), static_cast<int32_t>(std::floor(pyw * m->bs_inv)),
                          static_cast<int32_t>(std::floor(pzw * m->bs_inv))};
            if (ownerOf(b, c.world_size) != c.rank) continue;
            if (!seen.insert({b.x, b.y, b.z}).second) continue;
            bool created = false;
            Block* blk = m->allocate(b, &created);
            if (created) ++n_new;
            work.push_back(blk);
          }
        }
      }
    } else
    for (int dz = -n; dz <= n; ++dz) {
      for (int dy = -n; dy <= n; ++dy) {
        for (int dx = -n; dx <= n; ++dx) {
          const I3 b = {bc.x + dx, bc.y + dy, bc.z + dz};
          const float cxw = (static_cast<float>(b.x) + 0.5f) * m->bs;
          const float cyw = (static_cast<float>(b.y) + 0.5f) * m->bs;
          const float czw = (static_cast<float>(b.z) + 0.5f) * m->bs;
          float pc[3];
          xform(ctx.pose.R, ctx.pose.t, cxw, cyw, czw, pc);
          if (!pointInFrustum(fr, pc, s->max_range, infl)) continue;
          if (ownerOf(b, c.world_size) != c.rank) continue;
          bool created = false;
          Block* blk = m->allocate(b, &created);
          if (created) ++n_new;
          work.push_back(blk);
        }
      }
    }
  } else {
    for (auto& kv : m->blocks) work.push_back(kv.second.get());
  }

  parallelFor(c.num_threads, work.size(), [&](size_t i) { integrateBlock(ctx, *work[i]); });

  if (stats) {
    stats->n_visible_blocks = work.size();
    stats->n_new_blocks = n_new;
    stats->n_visited_voxels = static_cast<uint64_t>(work.size()) * m->nvox;
    stats->n_updated_voxels = ctx.n_upd.load();
    stats->n_band_voxels = ctx.n_band.load();
  }
  return 0;
}

int orc_update_tracking_phase(orc_map* m, uint64_t stamp, int phase) {
  const orc_config& c = m->cfg;
  if (!c.with_tracking) return 0;
  if (phase & 1) {
  // tracking_integrator.cpp:75-77
  std::vector<Block*> all, updated;
  for (auto& kv : m->blocks) {
    all.push_back(kv.second.get());
    if (kv.second->tracking_updated) updated.push_back(kv.second.get());
  }
  // tracking_integrator.cpp:136-138
  const float thr = c.tsdf_occupancy_threshold < 0 ? c.tsdf_occupancy_threshold * -c.voxel_size
                                                   : c.tsdf_occupancy_threshold;
  // updateBlockTracking: tracking_integrator.cpp:133-166, updateTrackingDuration :224-246
  parallelFor(c.num_threads, all.size(), [&](size_t bi) {
    Block& b = *all[bi];
    b.tracking_updated = false;
    bool active_any = false;
    for (int i = 0; i < m->nvox; ++i) {
      TsdfVoxel& tv = b.tsdf[i];
      TrackingVoxel& tr = b.tracking[i];
      if (tv.distance < thr) tr.last_occupied = stamp;
      const bool was_active = tr.active;
      tr.active = toSeconds(tr.last_observed) >= toSeconds(stamp) - static_cast<double>(c.temporal_window);
      if (was_active && !tr.active) tr.to_remove = true;
      if (tr.active) active_any = true;
    }
    b.has_active_data = active_any;
  });
  m->ef_work = updated;
  }
  if (!(phase & 2)) return 0;
  const std::vector<Block*>& updated = m->ef_work;
  // updateBlockEverFree: tracking_integrator.cpp:168-222
  const int nn = c.neighbor_connectivity;
  const int vps = m->vps;
  parallelFor(c.num_threads, updated.size(), [&](size_t bi) {
    Block& b = *updated[bi];
    for (int iz = 0; iz < vps; ++iz)
      for (int iy = 0; iy < vps; ++iy)
        for (int ix = 0; ix < vps; ++ix) {
          TrackingVoxel& v = b.tracking[ix + vps * (iy + vps * iz)];
          if (v.ever_free || !voxelIsFree(c, v, stamp)) continue;
          bool bad = false;
          for (int k = 0; k < nn && !bad; ++k) {
            int nx = ix + kNeighborOffsets26[k][0], ny = iy + kNeighborOffsets26[k][1],
                nz = iz + kNeighborOffsets26[k][2];
            I3 nb = b.index;
            if (nx < 0) { nx += vps; nb.x--; } else if (nx >= vps) { nx -= vps; nb.x++; }
            if (ny < 0) { ny += vps; nb.y--; } else if (ny >= vps) { ny -= vps; nb.y++; }
            if (nz < 0) { nz += vps; nb.z--; } else if (nz >= vps) { nz -= vps; nb.z++; }
            const Block* nblk = (nb == b.index) ? &b : m->find(nb);
            if (!nblk) {
              // not in this rank's shard: consult the halo records of the other ranks (absent => missing block)
              auto hit = m->halo.find(nb);
              const int nl = nx + vps * (ny + vps * nz);
              if (hit == m->halo.end() || !((hit->second[nl >> 6] >> (nl & 63)) & 1ull)) bad = true;
              continue;
            }
            const TrackingVoxel& nv = nblk->tracking[nx + vps * (ny + vps * nz)];
            if (nv.ever_free) continue;
            if (!voxelIsFree(c, nv, stamp)) bad = true;
          }
          if (!bad) v.ever_free = true;
        }
  });
  return 0;
}

int orc_update_tracking(orc_map* m, uint64_t stamp) { return orc_update_tracking_phase(m, stamp, 3); }

static inline uint64_t packBlockKey(const I3& b) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(b.x + (1 << 20)) & 0x1fffffu)) |
         (static_cast<uint64_t>(static_cast<uint32_t>(b.y + (1 << 20)) & 0x1fffffu) << 21) |
         (static_cast<uint64_t>(static_cast<uint32_t>(b.z + (1 << 20)) & 0x1fffffu) << 42);
}

// halo record: 66 x u64 = [packed block index, valid, 64 words of free-or-ever-free bits] (khronos_amd.h)
int64_t orc_export_halo(orc_map* m, uint64_t stamp, uint64_t* recs, int64_t cap) {
  std::memset(recs, 0, sizeof(uint64_t) * 66 * static_cast<size_t>(cap));
  int64_t n = 0;
  for (const I3& idx : m->sortedIndices()) {
    if (n >= cap) return -1;
    const Block* b = m->find(idx);
    uint64_t* r = recs + 66 * n;
    r[0] = packBlockKey(idx);
    r[1] = 1;
    for (int i = 0; i < m->nvox; ++i) {
      const TrackingVoxel& v = b->tracking[i];
      if (v.ever_free || voxelIsFree(m->cfg, v, stamp)) r[2 + (i >> 6)] |= 1ull << (i & 63);
    }
    ++n;
  }
  return n;
}

void orc_import_halo(orc_map* m, const uint64_t* recs, int64_t n) {
  m->halo.clear();
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t* r = recs + 66 * i;
    if (r[1] != 1) continue;
    const I3 b = {static_cast<int32_t>(r[0] & 0x1fffffu) - (1 << 20), static_cast<int32_t>((r[0] >> 21) & 0x1fffffu) - (1 << 20),
                  static_cast<int32_t>((r[0] >> 42) & 0x1fffffu) - (1 << 20)};
    if (ownerOf(b, m->cfg.world_size) == m->cfg.rank) continue;
    m->halo[b] = std::vector<uint64_t>(r + 2, r + 66);
  }
}

int64_t orc_reset_inactive(orc_map* m, int32_t* removed, int64_t cap) {
  // tracking_integrator.cpp:106-131
  int64_t n = 0;
  if (!m->cfg.with_tracking) return 0;
  for (const I3& idx : m->sortedIndices()) {
    Block* b = m->find(idx);
    bool remove_block = true;
    for (int i = 0; i < m->nvox; ++i) {
      if (!b->tracking[i].to_remove) { remove_block = false; break; }
    }
    if (!b->has_active_data || remove_block) {
      if (removed && n < cap) { removed[3 * n] = idx.x; removed[3 * n + 1] = idx.y; removed[3 * n + 2] = idx.z; }
      ++n;
      m->blocks.erase(idx);
    }
  }
  return n;
}

void orc_mark_all_inactive(orc_map* m) {
  for (auto& kv : m->blocks) kv.second->has_active_data = false;
}

void orc_clear_updated(orc_map* m) {
  // TsdfBlock::clearUpdated (active_window.cpp:169-171): clears the 'updated' flag only; mesh_updated is
  // cleared by generateMesh, tracking_updated by the tracking integrator (ASSUMPTIONS.md A.6)
  for (auto& kv : m->blocks) kv.second->updated = false;
}

// ---------------------------------------------------------------------------------------------
// ConnectedSemantics (connected_semantics.cpp) and the tracker's per-cluster voxel sets
// ---------------------------------------------------------------------------------------------
struct ObjCluster {
  int semantic_id = -1;
  std::vector<int> pixels;  // linear row-major pixel indices
  uint32_t first_cm = 0xffffffffu;
};

static void finishCluster(ObjCluster& c, int W, int H) {
  for (int i : c.pixels) {
    const uint32_t cm = static_cast<uint32_t>(i % W) * static_cast<uint32_t>(H) + static_cast<uint32_t>(i / W);
    c.first_cm = std::min(c.first_cm, cm);
  }
}

int orc_detect_objects(const orc_config* cfg, const orc_object_detector_config* oc, const orc_sensor* s, const orc_frame* f,
                       int32_t* object_image, orc_cluster* clusters_out, int cap) {
  const int W = s->width, H = s->height;
  const size_t n = static_cast<size_t>(W) * H;
  std::fill(object_image, object_image + n, 0);
  if (!f->label) return 0;
  std::vector<float> range(n), vertex(3 * n);
  orc_parse_input(cfg, s, f->world_T_sensor, f->depth, range.data(), vertex.data());
  std::set<int32_t> object_labels(oc->object_labels, oc->object_labels + oc->n_object_labels);
  auto isObject = [&](int32_t l) { return object_labels.count(l) != 0; };
  std::vector<ObjCluster> clusters;

  if (oc->use_3d) {
    // computeCandidateVoxels (:123-144)
    const float inv = 1.f / oc->grid_size;
    std::map<int, std::map<std::array<int64_t, 3>, std::vector<int>>> maps;
    for (int u = 0; u < W; ++u) {
      for (int v = 0; v < H; ++v) {
        const int i = v * W + u;
        if (oc->max_range > 0.f && range[i] > oc->max_range) continue;
        const int sem = f->label[i];
        if (!isObject(sem)) continue;
        const float* p = &vertex[3 * i];
        const std::array<int64_t, 3> vox = {static_cast<int64_t>(std::floor(p[0] * inv)), static_cast<int64_t>(std::floor(p[1] * inv)),
                                            static_cast<int64_t>(std::floor(p[2] * inv))};
        maps[sem][vox].push_back(i);
      }
    }
    // semanticClustering3D (:71-121): region growing per semantic id
    const int nn = oc->use_full_connectivity ? 26 : 6;
    for (auto& sm : maps) {
      auto& vox_to_pix = sm.second;
      std::vector<ObjCluster> of_this_id;
      while (!vox_to_pix.empty()) {
        ObjCluster c;
        c.semantic_id = sm.first;
        std::vector<std::array<int64_t, 3>> stack;
        auto it0 = vox_to_pix.begin();
        stack.push_back(it0->first);
        c.pixels.insert(c.pixels.end(), it0->second.begin(), it0->second.end());
        vox_to_pix.erase(it0);
        while (!stack.empty()) {
          const auto vi = stack.back();
          stack.pop_back();
          for (int k = 0; k < nn; ++k) {
            const std::array<int64_t, 3> nb = {vi[0] + kNeighborOffsets26[k][0], vi[1] + kNeighborOffsets26[k][1],
                                               vi[2] + kNeighborOffsets26[k][2]};
            auto it = vox_to_pix.find(nb);
            if (it == vox_to_pix.end()) continue;
            stack.push_back(it->first);
            c.pixels.insert(c.pixels.end(), it->second.begin(), it->second.end());
            vox_to_pix.erase(it);
          }
        }
        finishCluster(c, W, H);
        of_this_id.push_back(std::move(c));
      }
      // ASSUMPTIONS.md C.4: the start voxel comes from an unordered map in the reference; canonical order here
      std::sort(of_this_id.begin(), of_this_id.end(), [](const ObjCluster& a, const ObjCluster& b) { return a.first_cm < b.first_cm; });
      for (auto& c : of_this_id) {
        const int size = static_cast<int>(c.pixels.size());
        if (size < oc->min_cluster_size || (oc->max_cluster_size > 0 && size > oc->max_cluster_size)) continue;  // :104-110
        clusters.push_back(std::move(c));
      }
    }
    for (size_t k = 0; k < clusters.size(); ++k)
      for (int i : clusters[k].pixels) object_image[i] = static_cast<int32_t>(k + 1);  // :111-116
  } else {
    // semanticClustering2D + growCluster2D (:146-198)
    std::vector<int32_t> ids_all;  // id of every grown cluster, before filtering
    for (int u = 0; u < W; ++u) {
      for (int v = 0; v < H; ++v) {
        if (object_image[v * W + u] != 0) continue;
        const int sem = f->label[v * W + u];
        if (!isObject(sem)) continue;
        ObjCluster c;
        c.semantic_id = sem;
        const int32_t id = static_cast<int32_t>(clusters.size() + 1);
        std::vector<std::pair<int, int>> stack{{u, v}};
        c.pixels.push_back(v * W + u);
        object_image[v * W + u] = id;
        while (!stack.empty()) {
          const auto px = stack.back();
          stack.pop_back();
          for (int dv = -1; dv <= 1; ++dv)
            for (int du = -1; du <= 1; ++du) {
              if (du == 0 && dv == 0) continue;
              if (!oc->use_full_connectivity && du != 0 && dv != 0) continue;
              const int nu = px.first + du, nv = px.second + dv;
              if (nu < 0 || nv < 0 || nu >= W || nv >= H) continue;
              if (object_image[nv * W + nu] != 0) continue;
              if (f->label[nv * W + nu] != sem) continue;
              c.pixels.push_back(nv * W + nu);
              object_image[nv * W + nu] = id;
              stack.push_back({nu, nv});
            }
        }
        finishCluster(c, W, H);
        clusters.push_back(std::move(c));
      }
    }
    // filterClusters (:200-216): ids of the survivors are kept
    std::vector<ObjCluster> kept;
    std::vector<int32_t> kept_ids;
    for (size_t k = 0; k < clusters.size(); ++k) {
      if (static_cast<int>(clusters[k].pixels.size()) < oc->min_cluster_size) {
        for (int i : clusters[k].pixels) object_image[i] = 0;
      } else {
        kept_ids.push_back(static_cast<int32_t>(k + 1));
        kept.push_back(std::move(clusters[k]));
      }
    }
    clusters = std::move(kept);
    for (size_t k = 0; k < clusters.size() && static_cast<int>(k) < cap; ++k) clusters_out[k].id = kept_ids[k];
  }
  for (size_t k = 0; k < clusters.size() && static_cast<int>(k) < cap; ++k) {
    orc_cluster& o = clusters_out[k];
    if (oc->use_3d) o.id = static_cast<int32_t>(k + 1);
    o.semantic_id = clusters[k].semantic_id;
    o.num_pixels = clusters[k].pixels.size();
    double sum[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d) { o.bbox_min[d] = 3.0e38f; o.bbox_max[d] = -3.0e38f; }
    for (int i : clusters[k].pixels)
      for (int d = 0; d < 3; ++d) {
        const float x = vertex[3 * i + d];
        o.bbox_min[d] = std::min(o.bbox_min[d], x);
        o.bbox_max[d] = std::max(o.bbox_max[d], x);
        sum[d] += x;
      }
    for (int d = 0; d < 3; ++d) o.centroid[d] = static_cast<float>(sum[d] / static_cast<double>(o.num_pixels));
  }
  return static_cast<int>(clusters.size());
}

int64_t orc_cluster_voxels(const orc_config* cfg, const orc_sensor* s, const orc_frame* f, const int32_t* id_image, float voxel_size,
                           int32_t* ids_out, int64_t* voxels_out, int64_t cap) {
  const int W = s->width, H = s->height;
  const size_t n = static_cast<size_t>(W) * H;
  std::vector<float> vertex(3 * n);
  orc_parse_input(cfg, s, f->world_T_sensor, f->depth, nullptr, vertex.data());
  const float inv = 1.f / voxel_size;  // spatial_hash::Grid
  std::set<std::array<int64_t, 4>> pairs;
  for (size_t i = 0; i < n; ++i) {
    if (id_image[i] <= 0) continue;
    const float* p = &vertex[3 * i];
    pairs.insert({static_cast<int64_t>(id_image[i]), static_cast<int64_t>(std::floor(p[0] * inv)),
                  static_cast<int64_t>(std::floor(p[1] * inv)), static_cast<int64_t>(std::floor(p[2] * inv))});
  }
  int64_t k = 0;
  for (const auto& e : pairs) {
    if (k < cap) {
      ids_out[k] = static_cast<int32_t>(e[0]);
      for (int d = 0; d < 3; ++d) voxels_out[3 * k + d] = e[d + 1];
    }
    ++k;
  }
  return k;
}

// ---------------------------------------------------------------------------------------------
// RayVerificator (khronos/src/backend/change_detection/ray_verificator.cpp)
// ---------------------------------------------------------------------------------------------
struct orc_rayver {
  float block_size, inv, radial_tol, depth_tol;
  std::vector<uint64_t> stamp;
  std::vector<std::array<float, 3>> src, tgt;
  std::map<std::array<int64_t, 3>, std::set<size_t>> block_seen_by_rays;  // ordered containers: ascending ray index
};

static inline std::array<float, 3> rvSub(const std::array<float, 3>& a, const std::array<float, 3>& b) {
  return {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
}
static inline float rvDot(const std::array<float, 3>& a, const std::array<float, 3>& b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline float rvNorm(const std::array<float, 3>& a) { return std::sqrt(rvDot(a, a)); }

orc_rayver* orc_rv_create(float block_size, float radial_tolerance, float depth_tolerance) {
  auto* rv = new orc_rayver();
  rv->block_size = block_size;
  rv->inv = 1.f / block_size;
  rv->radial_tol = radial_tolerance;
  rv->depth_tol = depth_tolerance;
  return rv;
}
void orc_rv_destroy(orc_rayver* rv) { delete rv; }

void orc_rv_add_rays(orc_rayver* rv, int64_t n, const uint64_t* stamps, const float* sources, const float* targets) {
  for (int64_t i = 0; i < n; ++i) {
    rv->stamp.push_back(stamps[i]);
    rv->src.push_back({sources[3 * i], sources[3 * i + 1], sources[3 * i + 2]});
    rv->tgt.push_back({targets[3 * i], targets[3 * i + 1], targets[3 * i + 2]});
    const size_t ray_index = rv->stamp.size() - 1;
    // addRayToHash (:327-349)
    const auto& source = rv->src.back();
    const auto d = rvSub(rv->tgt.back(), source);
    const float max_depth = rvNorm(d);
    if (!std::isfinite(max_depth) || !(max_depth > 0.f)) continue;  // reference: endless march / NaN block index (undefined)
    const std::array<float, 3> direction = {d[0] / max_depth, d[1] / max_depth, d[2] / max_depth};
    const float ray_step = rv->block_size / 4;
    float ray_distance = 0.f;
    while (ray_distance <= max_depth) {
      ray_distance += ray_step;
      const std::array<float, 3> p = {source[0] + ray_distance * direction[0], source[1] + ray_distance * direction[1],
                                      source[2] + ray_distance * direction[2]};
      const std::array<int64_t, 3> index = {static_cast<int64_t>(std::floor(p[0] * rv->inv)), static_cast<int64_t>(std::floor(p[1] * rv->inv)),
                                            static_cast<int64_t>(std::floor(p[2] * rv->inv))};
      rv->block_seen_by_rays[index].insert(ray_index);
    }
  }
}

int64_t orc_rv_num_pairs(const orc_rayver* rv) {
  int64_t n = 0;
  for (const auto& kv : rv->block_seen_by_rays) n += static_cast<int64_t>(kv.second.size());
  return n;
}

void orc_rv_check(const orc_rayver* rv, const float* pt, uint64_t earliest, uint64_t latest, uint64_t* present, int64_t cap_present,
                  int64_t* n_present, uint64_t* absent, int64_t cap_absent, int64_t* n_absent) {
  // check (:66-145)
  *n_present = 0;
  *n_absent = 0;
  const std::array<float, 3> point = {pt[0], pt[1], pt[2]};
  const std::array<int64_t, 3> index = {static_cast<int64_t>(std::floor(point[0] * rv->inv)), static_cast<int64_t>(std::floor(point[1] * rv->inv)),
                                        static_cast<int64_t>(std::floor(point[2] * rv->inv))};
  const auto it = rv->block_seen_by_rays.find(index);
  if (it == rv->block_seen_by_rays.end()) return;
  for (size_t ray_index : it->second) {
    const uint64_t ts = rv->stamp[ray_index];
    if (ts < earliest || ts > latest) continue;
    const auto& source = rv->src[ray_index];
    const auto& vertex = rv->tgt[ray_index];
    const auto ps = rvSub(point, source);
    const float depth = rvNorm(ps);
    const std::array<float, 3> direction = {ps[0] / depth, ps[1] / depth, ps[2] / depth};
    const auto sv = rvSub(source, vertex);
    const std::array<float, 3> cr = {ps[1] * sv[2] - ps[2] * sv[1], ps[2] * sv[0] - ps[0] * sv[2], ps[0] * sv[1] - ps[1] * sv[0]};
    const float radial_distance = rvNorm(cr) / depth;
    if (radial_distance > rv->radial_tol) continue;
    const float depth_distance = rvDot(rvSub(vertex, source), direction);
    if (depth - depth_distance > rv->depth_tol) continue;
    if (depth_distance - depth > rv->depth_tol) {
      if (*n_absent < cap_absent) absent[*n_absent] = ts;
      ++*n_absent;
    } else {
      if (*n_present < cap_present) present[*n_present] = ts;
      ++*n_present;
    }
  }
}

void orc_allocate_block(orc_map* m, int32_t bx, int32_t by, int32_t bz) { m->allocate({bx, by, bz}); }

int64_t orc_num_blocks(const orc_map* m) { return static_cast<int64_t>(m->blocks.size()); }

int64_t orc_block_indices(const orc_map* m, int32_t* out, int64_t cap) {
  auto v = m->sortedIndices();
  int64_t n = 0;
  for (auto& i : v) {
    if (n >= cap) break;
    out[3 * n] = i.x; out[3 * n + 1] = i.y; out[3 * n + 2] = i.z;
    ++n;
  }
  return static_cast<int64_t>(v.size());
}

int orc_set_block_flags(orc_map* m, int32_t bx, int32_t by, int32_t bz, uint8_t flags) {
  Block* b = m->find({bx, by, bz});
  if (!b) return -1;
  b->updated = (flags & 1) != 0;
  b->mesh_updated = (flags & 2) != 0;
  b->tracking_updated = (flags & 4) != 0;
  return 0;
}

int orc_remove_block(orc_map* m, int32_t bx, int32_t by, int32_t bz) { return m->blocks.erase({bx, by, bz}) ? 0 : -1; }

int orc_set_distance(orc_map* m, int32_t bx, int32_t by, int32_t bz, const float* distance) {
  Block* b = m->find({bx, by, bz});
  if (!b) return -1;
  for (int i = 0; i < m->nvox; ++i) b->tsdf[i].distance = distance[i];
  return 0;
}

int orc_get_block(const orc_map* m, int32_t bx, int32_t by, int32_t bz, float* distance, float* weight,
                  uint8_t* color, uint64_t* last_observed, uint64_t* last_occupied, uint8_t* flags,
                  uint32_t* sem_label, float* likelihoods, uint8_t* block_flags) {
  const Block* b = m->find({bx, by, bz});
  if (!b) return -1;
  const int K = m->cfg.num_labels;
  for (int i = 0; i < m->nvox; ++i) {
    if (distance) distance[i] = b->tsdf[i].distance;
    if (weight) weight[i] = b->tsdf[i].weight;
    if (color) {
      color[4 * i] = b->tsdf[i].r; color[4 * i + 1] = b->tsdf[i].g;
      color[4 * i + 2] = b->tsdf[i].b; color[4 * i + 3] = b->tsdf[i].a;
    }
    uint8_t fl = 0;
    if (m->cfg.with_tracking) {
      if (last_observed) last_observed[i] = b->tracking[i].last_observed;
      if (last_occupied) last_occupied[i] = b->tracking[i].last_occupied;
      fl |= b->tracking[i].active ? 1 : 0;
      fl |= b->tracking[i].ever_free ? 2 : 0;
      fl |= b->tracking[i].to_remove ? 4 : 0;
    } else {
      if (last_observed) last_observed[i] = 0;
      if (last_occupied) last_occupied[i] = 0;
    }
    if (m->cfg.with_semantics) {
      fl |= b->semantic[i].empty ? 0 : 8;
      if (sem_label) sem_label[i] = b->semantic[i].semantic_label;
      if (likelihoods)
        for (int k = 0; k < K; ++k)
          likelihoods[static_cast<size_t>(k) * m->nvox + i] =
              b->semantic[i].empty ? 0.f : b->likelihoods[static_cast<size_t>(i) * K + k];
    } else if (sem_label) {
      sem_label[i] = 0;
    }
    if (flags) flags[i] = fl;
  }
  if (block_flags)
    *block_flags = (b->updated ? 1 : 0) | (b->mesh_updated ? 2 : 0) | (b->tracking_updated ? 4 : 0) |
                   (b->has_active_data ? 8 : 0);
  return 0;
}

// Whole-map digests (test tooling; the definition is include/khronos_amd.h: khr_map_digest): per layer the sum over all
// blocks and voxels of mix(mix(key * G + layer * L + i) ^ value bits), on the values orc_get_block hands out.
static inline uint64_t digestMix(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
static inline uint64_t digestTerm(uint64_t key, uint32_t layer, uint64_t i, uint64_t value) {
  return digestMix(digestMix(key * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(layer) * 0x632be59bd9b4e019ull + i) ^ value);
}
static inline uint32_t floatBits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
void orc_map_digest(const orc_map* m, uint64_t* out /* 12 */) {
  const std::vector<I3> idx = m->sortedIndices();
  const int K = m->cfg.num_labels, nv = m->nvox;
  std::vector<std::array<uint64_t, 12>> part(idx.size());
  parallelFor(m->cfg.num_threads, idx.size(), [&](size_t bi) {
    const Block* b = m->find(idx[bi]);
    const uint64_t key = packBlockKey(idx[bi]);
    std::array<uint64_t, 12> a{};
    for (int i = 0; i < nv; ++i) {
      const TsdfVoxel& t = b->tsdf[i];
      a[0] += digestTerm(key, 0, i, floatBits(t.distance));
      a[1] += digestTerm(key, 1, i, floatBits(t.weight));
      a[2] += digestTerm(key, 2, i, static_cast<uint32_t>(t.r) | (static_cast<uint32_t>(t.g) << 8) | (static_cast<uint32_t>(t.b) << 16) |
                                        (static_cast<uint32_t>(t.a) << 24));
      uint8_t fl = 0;
      uint64_t lobs = 0, locc = 0;
      if (m->cfg.with_tracking) {
        const TrackingVoxel& v = b->tracking[i];
        lobs = v.last_observed;
        locc = v.last_occupied;
        fl |= (v.active ? 1 : 0) | (v.ever_free ? 2 : 0) | (v.to_remove ? 4 : 0);
      }
      uint32_t lab = 0;
      if (m->cfg.with_semantics) {
        const bool valid = !b->semantic[i].empty;
        fl |= valid ? 8 : 0;
        lab = b->semantic[i].semantic_label;
        for (int k = 0; k < K; ++k)
          a[7] += digestTerm(key, 7, static_cast<uint64_t>(k) * nv + i, valid ? floatBits(b->likelihoods[static_cast<size_t>(i) * K + k]) : 0u);
      }
      a[3] += digestTerm(key, 3, i, lobs);
      a[4] += digestTerm(key, 4, i, locc);
      a[5] += digestTerm(key, 5, i, fl);
      a[6] += digestTerm(key, 6, i, lab);
    }
    a[8] = digestTerm(key, 8, 0, (b->updated ? 1 : 0) | (b->mesh_updated ? 2 : 0) | (b->tracking_updated ? 4 : 0) | (b->has_active_data ? 8 : 0));
    a[9] = digestMix(key);
    a[10] = 1;
    part[bi] = a;
  });
  for (int l = 0; l < 12; ++l) out[l] = 0;
  for (const auto& a : part)
    for (int l = 0; l < 12; ++l) out[l] += a[l];
}

// ---------------------------------------------------------------------------------------------
// FreeSpaceMotionDetector (free_space_motion_detector.cpp:73-399)
// ---------------------------------------------------------------------------------------------
static inline uint64_t packVoxelKey(int64_t x, int64_t y, int64_t z) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(x + (1 << 20)) & 0x1fffffu)) |
         (static_cast<uint64_t>(static_cast<uint32_t>(y + (1 << 20)) & 0x1fffffu) << 21) |
         (static_cast<uint64_t>(static_cast<uint32_t>(z + (1 << 20)) & 0x1fffffu) << 42);
}

// stage 1: setUpPointMap(Part) (free_space_motion_detector.cpp:105-203) against THIS shard of the map: one
// key per pixel = packed global voxel index, bit 63 = the voxel is ever-free (seed); 0 = skipped / not in
// this shard (multi-GPU key exchange format of include/khronos_amd.h)
void orc_motion_keys(orc_map* m, const orc_sensor* s, const orc_frame* f, uint64_t* keys_out) {
  const orc_config& c = m->cfg;
  const int W = s->width, H = s->height;
  std::vector<float> range(static_cast<size_t>(W) * H), vertex(static_cast<size_t>(W) * H * 3);
  orc_parse_input(&c, s, f->world_T_sensor, f->depth, range.data(), vertex.data());
  // :80  min_z_world_ = sensor z + min_z_coordinate
  const float min_z_world =
      static_cast<float>(f->world_T_sensor[11] + static_cast<double>(c.md_min_z_coordinate));
  const int vps = m->vps;
  for (int v = 0; v < H; ++v) {
    for (int u = 0; u < W; ++u) {
      const int i = v * W + u;
      keys_out[i] = 0;
      const float r = range[i];
      if (r <= 0.f || r > c.md_max_range) continue;  // :169-172
      const float* p = &vertex[3 * i];
      if (p[2] < min_z_world) continue;  // :176
      const I3 bi = {static_cast<int32_t>(std::floor(p[0] * m->bs_inv)),
                     static_cast<int32_t>(std::floor(p[1] * m->bs_inv)),
                     static_cast<int32_t>(std::floor(p[2] * m->bs_inv))};
      const Block* b = m->find(bi);  // :180
      if (!b) continue;
      const float ox = static_cast<float>(bi.x) * m->bs, oy = static_cast<float>(bi.y) * m->bs,
                  oz = static_cast<float>(bi.z) * m->bs;
      const int vx = static_cast<int>(std::floor((p[0] - ox) * m->vs_inv));
      const int vy = static_cast<int>(std::floor((p[1] - oy) * m->vs_inv));
      const int vz = static_cast<int>(std::floor((p[2] - oz) * m->vs_inv));
      // :192-197 invalid voxel index (float rounding at block border): the pixel lands in a map entry
      // that can never be addressed by keyFromGlobalIndex => equivalent to dropping it.
      if (vx < 0 || vy < 0 || vz < 0 || vx >= vps || vy >= vps || vz >= vps) continue;
      uint64_t key = packVoxelKey(static_cast<int64_t>(bi.x) * vps + vx, static_cast<int64_t>(bi.y) * vps + vy,
                                  static_cast<int64_t>(bi.z) * vps + vz);
      if (b->tracking[vx + vps * (vy + vps * vz)].ever_free) key |= 1ull << 63;  // :198-201
      keys_out[i] = key;
    }
  }
}

// stage 2: clustering, merging, filtering and painting from the (complete) key image
int orc_detect_motion_from_keys(orc_map* m, int W, int H, const uint64_t* keys, int32_t* dynamic_image_out,
                                int64_t* n_seeds_out) {
  const orc_config& c = m->cfg;
  std::memset(dynamic_image_out, 0, sizeof(int32_t) * W * H);
  std::unordered_map<L3, std::vector<int32_t>, L3Hash> point_map;  // pixel = v*W+u, in (v, u) order
  std::unordered_set<L3, L3Hash> seeds;
  for (int i = 0; i < W * H; ++i) {
    const uint64_t k = keys[i];
    if (k == 0) continue;
    const L3 g = {static_cast<int64_t>(k & 0x1fffffu) - (1 << 20), static_cast<int64_t>((k >> 21) & 0x1fffffu) - (1 << 20),
                  static_cast<int64_t>((k >> 42) & 0x1fffffu) - (1 << 20)};
    point_map[g].push_back(i);
    if (k >> 63) seeds.insert(g);
  }
  if (n_seeds_out) *n_seeds_out = static_cast<int64_t>(seeds.size());

  // clusterDynamicVoxels :205-272.  Seed iteration order in the reference is unordered_set order
  // (implementation-defined); the canonical order here is ascending (x,y,z) (ASSUMPTIONS.md C.1).
  struct Cluster {
    std::vector<int32_t> pixels;
    std::vector<L3> voxels;
  };
  std::vector<Cluster> clusters;
  std::vector<L3> seed_list(seeds.begin(), seeds.end());
  std::sort(seed_list.begin(), seed_list.end());
  std::unordered_set<L3, L3Hash> closed;
  const int nn = c.md_neighbor_connectivity;
  for (const L3& seed : seed_list) {
    if (closed.count(seed)) continue;
    std::vector<L3> stack = {seed};
    Cluster cl;
    while (!stack.empty()) {
      const L3 g = stack.back();
      stack.pop_back();
      if (closed.count(g)) continue;
      closed.insert(g);
      auto it = point_map.find(g);
      if (it == point_map.end()) continue;
      cl.pixels.insert(cl.pixels.end(), it->second.begin(), it->second.end());
      cl.voxels.push_back(g);
      for (int k = 0; k < nn; ++k) {
        const L3 ng = {g.x + kNeighborOffsets26[k][0], g.y + kNeighborOffsets26[k][1],
                       g.z + kNeighborOffsets26[k][2]};
        if (seeds.count(ng)) {
          stack.push_back(ng);
        } else {
          auto it2 = point_map.find(ng);
          if (it2 != point_map.end()) {
            // NOTE: restated literally (:255-265): no closed-set test before the insert, so a non-seed
            // occupied voxel adjacent to several seeds of the same or different clusters is appended
            // once per adjacency.
            cl.pixels.insert(cl.pixels.end(), it2->second.begin(), it2->second.end());
            cl.voxels.push_back(ng);
            closed.insert(ng);
          }
        }
      }
    }
    clusters.push_back(std::move(cl));
  }
  // cluster.voxels is a set in the reference
  for (auto& cl : clusters) {
    std::sort(cl.voxels.begin(), cl.voxels.end());
    cl.voxels.erase(std::unique(cl.voxels.begin(), cl.voxels.end()), cl.voxels.end());
  }

  // mergeClusters :274-355
  const size_t nc = clusters.size();
  std::vector<uint8_t> overlap(nc * nc, 0);
  for (size_t i = 0; i < nc; ++i)
    for (size_t j = i + 1; j < nc; ++j) {
      bool ov = false;
      for (const L3& a : clusters[i].voxels) {
        for (const L3& b : clusters[j].voxels) {
          const double dx = static_cast<double>(a.x - b.x), dy = static_cast<double>(a.y - b.y),
                       dz = static_cast<double>(a.z - b.z);
          // (p1 - p2).norm() on an int64 vector (:349) -> integer sqrt semantics of Eigen: norm() of an
          // integer matrix is sqrt of the integer squared norm, truncated to the integer scalar type.
          const int64_t n2 = static_cast<int64_t>(dx * dx + dy * dy + dz * dz);
          const int64_t nrm = static_cast<int64_t>(std::sqrt(static_cast<double>(n2)));
          if (static_cast<float>(nrm) < c.md_min_separation_distance) { ov = true; break; }
        }
        if (ov) break;
      }
      overlap[i * nc + j] = overlap[j * nc + i] = ov;
    }
  std::vector<bool> merged(nc, false), keep(nc, false);
  std::function<void(size_t, std::vector<int>&)> connected = [&](size_t ci, std::vector<int>& out) {
    for (size_t i = 0; i < nc; ++i) {
      if (merged[i]) continue;
      if (overlap[ci * nc + i]) {
        merged[i] = true;
        out.push_back(static_cast<int>(i));
        connected(i, out);
      }
    }
  };
  for (size_t cur = 0; cur < nc; ++cur) {
    if (merged[cur]) continue;
    std::vector<int> idx;
    connected(cur, idx);
    for (int i : idx) {
      if (static_cast<size_t>(i) == cur) continue;
      auto& src = clusters[i];
      auto& dst = clusters[cur];
      dst.pixels.insert(dst.pixels.end(), src.pixels.begin(), src.pixels.end());
      dst.voxels.insert(dst.voxels.end(), src.voxels.begin(), src.voxels.end());
    }
    keep[cur] = true;
  }
  // applyClusterLevelFilters :365-379 + writeClustersToData :381-399
  int id = 1, n_out = 0;
  m->last_md_pixels.clear();
  for (size_t ci = 0; ci < nc; ++ci) {
    if (!keep[ci]) continue;
    const int size = static_cast<int>(clusters[ci].pixels.size());
    if (size < c.md_min_cluster_size || size > c.md_max_cluster_size) continue;
    m->last_md_pixels.push_back(clusters[ci].pixels);
    for (int32_t px : clusters[ci].pixels) dynamic_image_out[px] = id;
    if (id < 255) ++id;
    ++n_out;
  }
  return n_out;
}

// the kept clusters of the latest orc_detect_motion* call as the reference's consumers see them: length of cluster.pixels (a
// boundary voxel's pixels once per adjacent seed, :255-265) and the mean of the listed pixels' vertices -- what
// MeshObjectExtractor::extractDynamicObject (mesh_object_extractor.cpp:136-147) and MaxIoUTracker::computeCentroid for
// track_by pixels (max_iou_tracker.cpp:541-548) compute.  Sum in double, in list order.
int64_t orc_last_motion_clusters(const orc_map* m, const orc_sensor* s, const orc_frame* f, int64_t* n_listed_out, float* centroid_out,
                                 int64_t cap) {
  const int W = s->width, H = s->height;
  std::vector<float> range(static_cast<size_t>(W) * H), vertex(static_cast<size_t>(W) * H * 3);
  orc_parse_input(&m->cfg, s, f->world_T_sensor, f->depth, range.data(), vertex.data());
  int64_t k = 0;
  for (const auto& px : m->last_md_pixels) {
    if (k < cap) {
      double sum[3] = {0, 0, 0};
      for (int32_t p : px)
        for (int a = 0; a < 3; ++a) sum[a] += static_cast<double>(vertex[3 * static_cast<size_t>(p) + a]);
      n_listed_out[k] = static_cast<int64_t>(px.size());
      for (int a = 0; a < 3; ++a) centroid_out[3 * k + a] = static_cast<float>(sum[a] / static_cast<double>(px.size()));
    }
    ++k;
  }
  return k;
}

int orc_detect_motion(orc_map* m, const orc_sensor* s, const orc_frame* f, int32_t* dynamic_image_out,
                      int64_t* n_seeds_out) {
  std::vector<uint64_t> keys(static_cast<size_t>(s->width) * s->height);
  orc_motion_keys(m, s, f, keys.data());
  return orc_detect_motion_from_keys(m, s->width, s->height, keys.data(), dynamic_image_out, n_seeds_out);
}

// ---------------------------------------------------------------------------------------------
// MeshIntegrator (ASSUMPTIONS.md A.5)
// ---------------------------------------------------------------------------------------------
static const int kCubeOffsets[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0},
                                       {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
static const int kEdgePairs[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                      {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

int64_t orc_generate_mesh(orc_map* m, int only_mesh_updated, int clear_flag) {
  const orc_config& c = m->cfg;
  const int vps = m->vps;
  const float mesh_eps = c.mesh_degenerate_eps > 0.f ? c.mesh_degenerate_eps : 1e-6f;
  std::vector<Block*> work;
  for (auto& kv : m->blocks) {
    if (!only_mesh_updated || kv.second->mesh_updated) work.push_back(kv.second.get());
  }
  parallelFor(c.num_threads, work.size(), [&](size_t bi) {
    Block& b = *work[bi];
    MeshBlock& mesh = b.mesh;
    mesh = MeshBlock();
    const float ox = static_cast<float>(b.index.x) * m->bs, oy = static_cast<float>(b.index.y) * m->bs,
                oz = static_cast<float>(b.index.z) * m->bs;
    // neighbour blocks (+x,+y,+z combos)
    const Block* nb[8];
    const uint32_t* nrec[8];  // halo record of a neighbour that lives on another rank
    for (int k = 0; k < 8; ++k) {
      const I3 ni = {b.index.x + (k & 1), b.index.y + ((k >> 1) & 1), b.index.z + ((k >> 2) & 1)};
      nb[k] = k == 0 ? &b : m->find(ni);
      nrec[k] = nullptr;
      if (!nb[k]) {
        auto it = m->mesh_halo.find(ni);
        if (it != m->mesh_halo.end()) nrec[k] = it->second.data();
      }
    }
    const int PL = vps * vps, PW = PL * 6;
    auto planeIdx = [&](int sel, int x, int y, int z, int* pl) {
      *pl = (sel & 1) ? 0 : ((sel & 2) ? 1 : 2);
      return (sel & 1) ? (y + vps * z) : ((sel & 2) ? (x + vps * z) : (x + vps * y));
    };
    for (int iz = 0; iz < vps; ++iz)
      for (int iy = 0; iy < vps; ++iy)
        for (int ix = 0; ix < vps; ++ix) {
          float sdf[8], pos[8][3];
          const Block* cb[8];
          int clin[8];
          const uint32_t* crec[8];
          int cpl[8];
          bool ok = true;
          for (int k = 0; k < 8 && ok; ++k) {
            int x = ix + kCubeOffsets[k][0], y = iy + kCubeOffsets[k][1], z = iz + kCubeOffsets[k][2];
            int sel = 0;
            if (x >= vps) { x -= vps; sel |= 1; }
            if (y >= vps) { y -= vps; sel |= 2; }
            if (z >= vps) { z -= vps; sel |= 4; }
            const Block* blk = nb[sel];
            crec[k] = nullptr;
            if (!blk) {
              if (!nrec[sel]) { ok = false; break; }
              // remote neighbour: distance / weight from its halo record
              int pl;
              const int pi = planeIdx(sel, x, y, z, &pl);
              const uint32_t* pw = nrec[sel] + 4 + pl * PW;
              float dd, ww;
              std::memcpy(&dd, &pw[pi], 4);
              std::memcpy(&ww, &pw[PL + pi], 4);
              if (!(ww >= c.mesh_min_weight)) { ok = false; break; }
              sdf[k] = dd;
              cb[k] = nullptr;
              crec[k] = nrec[sel];
              cpl[k] = pl;
              clin[k] = pi;
            } else {
            const int lin = x + vps * (y + vps * z);
            const TsdfVoxel& tv = blk->tsdf[lin];
            if (!(tv.weight >= c.mesh_min_weight)) { ok = false; break; }
            sdf[k] = tv.distance;
            cb[k] = blk;
            clin[k] = lin;
            }
            pos[k][0] = ox + (static_cast<float>(ix + kCubeOffsets[k][0]) + 0.5f) * c.voxel_size;
            pos[k][1] = oy + (static_cast<float>(iy + kCubeOffsets[k][1]) + 0.5f) * c.voxel_size;
            pos[k][2] = oz + (static_cast<float>(iz + kCubeOffsets[k][2]) + 0.5f) * c.voxel_size;
          }
          if (!ok) continue;
          int index = 0;
          for (int k = 0; k < 8; ++k)
            if (sdf[k] < 0.f) index |= (1 << k);
          if (index == 0 || index == 255) continue;
          float ev[12][3];
          int esrc[12];
          for (int e = 0; e < 12; ++e) {
            const int a = kEdgePairs[e][0], bq = kEdgePairs[e][1];
            const float s0 = sdf[a], s1 = sdf[bq];
            esrc[e] = a;
            if ((s0 < 0.f && s1 >= 0.f) || (s0 >= 0.f && s1 < 0.f)) {
              const float diff = s0 - s1;
              float t = 0.5f;
              if (std::fabs(diff) >= mesh_eps) t = s0 / diff;
              for (int d = 0; d < 3; ++d) ev[e][d] = pos[a][d] + t * (pos[bq][d] - pos[a][d]);
              if (c.mesh_attr_source == 1) {
                // [A] mesh_attr_source = containing voxel: the voxel whose cell holds the vertex; exactly half way the one with the
                // larger coordinate along the edge (floor of the vertex position)
                const bool b_is_upper = (kCubeOffsets[bq][0] + kCubeOffsets[bq][1] + kCubeOffsets[bq][2]) >
                                        (kCubeOffsets[a][0] + kCubeOffsets[a][1] + kCubeOffsets[a][2]);
                esrc[e] = (t < 0.5f) ? a : ((t > 0.5f) ? bq : (b_is_upper ? bq : a));
              } else {
                esrc[e] = (t <= 0.5f) ? a : bq;
              }
            }
          }
          const int8_t* row = kMcTriTable[index];
          for (int col = 0; row[col] != -1; col += 3) {
            for (int k = 2; k >= 0; --k) {
              const int e = row[col + k];
              mesh.points.push_back(ev[e][0]);
              mesh.points.push_back(ev[e][1]);
              mesh.points.push_back(ev[e][2]);
              const Block* sb = cb[esrc[e]];
              const int sl = clin[esrc[e]];
              if (!sb) {  // vertex attributes from the halo record of the remote source voxel
                const uint32_t* pw = crec[esrc[e]] + 4 + cpl[esrc[e]] * PW;
                const uint32_t col = pw[2 * PL + sl];
                mesh.colors.push_back(col & 0xff);
                mesh.colors.push_back((col >> 8) & 0xff);
                mesh.colors.push_back((col >> 16) & 0xff);
                mesh.colors.push_back((col >> 24) & 0xff);
                mesh.labels.push_back(c.with_semantics ? pw[3 * PL + sl] : 0u);
                const uint64_t st = c.with_tracking
                                        ? (static_cast<uint64_t>(pw[4 * PL + 2 * sl]) | (static_cast<uint64_t>(pw[4 * PL + 2 * sl + 1]) << 32))
                                        : 0u;
                mesh.first_seen.push_back(st);
                mesh.stamps.push_back(st);
                continue;
              }
              mesh.colors.push_back(sb->tsdf[sl].r);
              mesh.colors.push_back(sb->tsdf[sl].g);
              mesh.colors.push_back(sb->tsdf[sl].b);
              mesh.colors.push_back(sb->tsdf[sl].a);
              mesh.labels.push_back(c.with_semantics ? sb->semantic[sl].semantic_label : 0u);
              const uint64_t st = c.with_tracking ? sb->tracking[sl].last_observed : 0u;
              mesh.first_seen.push_back(st);
              mesh.stamps.push_back(st);
            }
          }
        }
    if (clear_flag) b.mesh_updated = false;
  });
  return static_cast<int64_t>(work.size());
}

// ---- mesh halo (multi-GPU emulation; protocol and record layout: include/khronos_amd.h) ----
int64_t orc_mesh_halo_requests(orc_map* m, int only_mesh_updated, uint64_t* keys_out, int64_t cap) {
  std::memset(keys_out, 0, sizeof(uint64_t) * cap);
  int64_t n = 0;
  for (const I3& idx : m->sortedIndices()) {
    const Block* b = m->find(idx);
    if (only_mesh_updated && !b->mesh_updated) continue;
    for (int k = 1; k < 8; ++k) {
      const I3 ni = {idx.x + (k & 1), idx.y + ((k >> 1) & 1), idx.z + ((k >> 2) & 1)};
      if (ownerOf(ni, m->cfg.world_size) == m->cfg.rank || m->find(ni)) continue;
      if (n < cap) keys_out[n] = packBlockKey(ni);
      ++n;
    }
  }
  return n;
}

int64_t orc_mesh_halo_export(orc_map* m, const uint64_t* reqs, int64_t n_req, uint32_t* recs, int64_t cap) {
  const int vps = m->vps, PL = vps * vps, PW = PL * 6, words = 4 + 3 * PW;
  std::memset(recs, 0, sizeof(uint32_t) * static_cast<size_t>(words) * cap);
  std::vector<I3> want;
  for (int64_t i = 0; i < n_req; ++i) {
    if (reqs[i] == 0) continue;
    const I3 b = {static_cast<int32_t>(reqs[i] & 0x1fffffu) - (1 << 20), static_cast<int32_t>((reqs[i] >> 21) & 0x1fffffu) - (1 << 20),
                  static_cast<int32_t>((reqs[i] >> 42) & 0x1fffffu) - (1 << 20)};
    if (ownerOf(b, m->cfg.world_size) == m->cfg.rank && m->find(b)) want.push_back(b);
  }
  std::sort(want.begin(), want.end());
  want.erase(std::unique(want.begin(), want.end()), want.end());
  if (static_cast<int64_t>(want.size()) > cap) return -1;
  for (size_t r = 0; r < want.size(); ++r) {
    const Block* b = m->find(want[r]);
    uint32_t* rec = recs + r * words;
    const uint64_t key = packBlockKey(want[r]);
    rec[0] = static_cast<uint32_t>(key);
    rec[1] = static_cast<uint32_t>(key >> 32);
    rec[2] = 1;
    for (int pl = 0; pl < 3; ++pl)
      for (int i = 0; i < PL; ++i) {
        const int a = i % vps, bq = i / vps;
        const int lin = pl == 0 ? (vps * (a + vps * bq)) : (pl == 1 ? (a + vps * vps * bq) : (a + vps * bq));
        uint32_t* pw = rec + 4 + pl * PW;
        std::memcpy(&pw[i], &b->tsdf[lin].distance, 4);
        std::memcpy(&pw[PL + i], &b->tsdf[lin].weight, 4);
        pw[2 * PL + i] = static_cast<uint32_t>(b->tsdf[lin].r) | (static_cast<uint32_t>(b->tsdf[lin].g) << 8) |
                         (static_cast<uint32_t>(b->tsdf[lin].b) << 16) | (static_cast<uint32_t>(b->tsdf[lin].a) << 24);
        pw[3 * PL + i] = m->cfg.with_semantics ? b->semantic[lin].semantic_label : 0u;
        const uint64_t st = m->cfg.with_tracking ? b->tracking[lin].last_observed : 0u;
        pw[4 * PL + 2 * i] = static_cast<uint32_t>(st);
        pw[4 * PL + 2 * i + 1] = static_cast<uint32_t>(st >> 32);
      }
  }
  return static_cast<int64_t>(want.size());
}

void orc_mesh_halo_import(orc_map* m, const uint32_t* recs, int64_t n) {
  const int words = 4 + 3 * 6 * m->vps * m->vps;
  m->mesh_halo.clear();
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* r = recs + static_cast<size_t>(i) * words;
    if (r[2] != 1) continue;
    const uint64_t key = static_cast<uint64_t>(r[0]) | (static_cast<uint64_t>(r[1]) << 32);
    const I3 b = {static_cast<int32_t>(key & 0x1fffffu) - (1 << 20), static_cast<int32_t>((key >> 21) & 0x1fffffu) - (1 << 20),
                  static_cast<int32_t>((key >> 42) & 0x1fffffu) - (1 << 20)};
    if (ownerOf(b, m->cfg.world_size) == m->cfg.rank) continue;
    m->mesh_halo[b] = std::vector<uint32_t>(r, r + words);
  }
}

int64_t orc_mesh_num_vertices(orc_map* m) {
  int64_t n = 0;
  for (auto& kv : m->blocks) n += static_cast<int64_t>(kv.second->mesh.labels.size());
  return n;
}

int64_t orc_mesh_copy(orc_map* m, float* points, uint8_t* colors, uint32_t* labels, uint64_t* first_seen,
                      uint64_t* stamps, int64_t cap) {
  int64_t n = 0;
  for (const I3& idx : m->sortedIndices()) {
    const MeshBlock& mb = m->find(idx)->mesh;
    const int64_t k = static_cast<int64_t>(mb.labels.size());
    if (n + k > cap) return -1;
    if (points) std::memcpy(points + 3 * n, mb.points.data(), sizeof(float) * 3 * k);
    if (colors) std::memcpy(colors + 4 * n, mb.colors.data(), 4 * k);
    if (labels) std::memcpy(labels + n, mb.labels.data(), sizeof(uint32_t) * k);
    if (first_seen) std::memcpy(first_seen + n, mb.first_seen.data(), sizeof(uint64_t) * k);
    if (stamps) std::memcpy(stamps + n, mb.stamps.data(), sizeof(uint64_t) * k);
    n += k;
  }
  return n;
}

int64_t orc_object_prune(orc_map* m, float min_confidence, float min_observations) {
  // mesh_object_extractor.cpp:246-264 + computeConfidence :342-356
  int64_t pruned = 0;
  const int K = m->cfg.num_labels;
  for (auto& kv : m->blocks) {
    Block& b = *kv.second;
    for (int i = 0; i < m->nvox; ++i) {
      TsdfVoxel& tv = b.tsdf[i];
      if (tv.distance > 0.f) continue;
      float conf;
      if (b.semantic[i].empty) {
        conf = 0.f;
      } else {
        const float* l = &b.likelihoods[static_cast<size_t>(i) * K];
        const float total = l[0] + l[1];
        conf = total < min_observations ? -1.f : l[1] / total;
      }
      if (conf < min_confidence) {
        tv.distance = m->cfg.truncation_distance;
        ++pruned;
      }
    }
  }
  return pruned;
}

}  // extern "C"
