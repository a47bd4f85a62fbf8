// ref_harness.cpp -- C ABI around the reference's OWN TrackingIntegrator / FreeSpaceMotionDetector / geometry utilities, compiled from
// /root/reference against the functional stand-ins of oracle/ref_recipe/standin (see ref_standin.h for what that pins and what
// it does not).  TEST INFRASTRUCTURE: built by oracle/ref_recipe/build_ref.sh into oracle/_ref/libref_khronos.so, loaded by
// tests/test_cpu_ref_pin.py only.
//
// The map on this side lives its own life: the tracking state (last_occupied, active, ever_free, to_remove, has_active_data)
// is written by the reference's code alone, frame after frame.  The one thing handed in from outside is what the projective
// integrator does (it is not in /root/reference): per block the TSDF distances, the last_observed stamps and the
// tracking_updated flag after a frame's update (ref_put_block).
#include <algorithm>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "khronos/active_window/active_window.h"
#include "khronos/active_window/integration/tracking_integrator.h"
#include "khronos/active_window/motion_detection/free_space_motion_detector.h"
#include "khronos/active_window/object_detection/connected_semantics.h"
#include "khronos/active_window/object_detection/instance_forwarding.h"
#include "khronos/active_window/object_extraction/mesh_object_extractor.h"
#include "khronos/active_window/data/frame_data_buffer.h"
#include "khronos/active_window/tracking/external_tracker.h"
#include "khronos/active_window/tracking/max_iou_tracker.h"
#include "khronos/backend/change_detection/background/ray_background_change_detector.h"
#include "khronos/backend/change_detection/objects/ray_object_change_detector.h"
#include "khronos/backend/change_detection/ray_change_detector.h"
#include "khronos/backend/change_detection/ray_verificator.h"
#include "khronos/utils/geometry_utils.h"

#include "../oracle.h"  // (the bridge behind the two integrators that are not in /root/reference)

// how often the reference's ObjectIntegrator::computeLabel ran / returned false (ref_label_hook_stats)
static std::atomic<uint64_t> g_label_hook_calls{0}, g_label_hook_skips{0};

namespace {
struct RefMap {
  std::unique_ptr<hydra::VolumetricMap> map;
  std::unique_ptr<khronos::TrackingIntegrator> tracking;
  std::unique_ptr<khronos::FreeSpaceMotionDetector> motion;
};

std::vector<hydra::BlockIndex> sortedIndices(const RefMap& r) {
  auto v = r.map->getTsdfLayer().allocatedBlockIndices();
  std::sort(v.begin(), v.end(), [](const hydra::BlockIndex& a, const hydra::BlockIndex& b) {
    return a[0] != b[0] ? a[0] < b[0] : (a[1] != b[1] ? a[1] < b[1] : a[2] < b[2]);
  });
  return v;
}
}  // namespace

extern "C" {

struct ref_config {
  float voxel_size;
  int32_t voxels_per_side;
  /* khronos::TrackingIntegrator::Config (tracking_integrator.h:59-83) */
  float temporal_buffer;
  float tsdf_occupancy_threshold;
  int32_t neighbor_connectivity;
  float temporal_window;
  /* khronos::FreeSpaceMotionDetector::Config (free_space_motion_detector.h:72-97) */
  int32_t md_neighbor_connectivity;
  int32_t md_min_cluster_size;
  int32_t md_max_cluster_size;
  float md_min_separation_distance;
  float md_max_range;
  float md_min_z_coordinate;
  int32_t num_threads;
};

RefMap* ref_create(const ref_config* c) {
  auto* r = new RefMap();
  hydra::VolumetricMap::Config mc;
  mc.voxel_size = c->voxel_size;
  mc.voxels_per_side = static_cast<size_t>(c->voxels_per_side);
  r->map = std::make_unique<hydra::VolumetricMap>(mc);
  khronos::TrackingIntegrator::Config tc;
  tc.temporal_buffer = c->temporal_buffer;
  tc.tsdf_occupancy_threshold = c->tsdf_occupancy_threshold;
  tc.neighbor_connectivity = c->neighbor_connectivity;
  tc.temporal_window = c->temporal_window;
  tc.num_threads = c->num_threads;
  r->tracking = std::make_unique<khronos::TrackingIntegrator>(tc);
  khronos::FreeSpaceMotionDetector::Config dc;
  dc.neighbor_connectivity = c->md_neighbor_connectivity;
  dc.min_cluster_size = c->md_min_cluster_size;
  dc.max_cluster_size = c->md_max_cluster_size;
  dc.min_separation_distance = c->md_min_separation_distance;
  dc.max_range = c->md_max_range;
  dc.min_z_coordinate = c->md_min_z_coordinate;
  dc.num_threads = c->num_threads;
  r->motion = std::make_unique<khronos::FreeSpaceMotionDetector>(dc);
  return r;
}

void ref_destroy(RefMap* r) { delete r; }

/* the projective integrator's footprint on one block (external to /root/reference): allocate when missing, TSDF distance and
 * last_observed of every voxel, the tracking_updated flag */
void ref_put_block(RefMap* r, const int32_t* idx, const float* distance, const uint64_t* last_observed, int tracking_updated) {
  const hydra::BlockIndex bi(idx[0], idx[1], idx[2]);
  if (!r->map->getTsdfLayer().hasBlock(bi)) r->map->allocateBlock(bi);
  auto tsdf = r->map->getTsdfLayer().getBlockPtr(bi);
  auto trk = r->map->getTrackingLayer()->getBlockPtr(bi);
  for (size_t i = 0; i < tsdf->numVoxels(); ++i) {
    tsdf->getVoxel(i).distance = distance[i];
    trk->getVoxel(i).last_observed = last_observed[i];
  }
  tsdf->tracking_updated = tracking_updated != 0;
}

/* TrackingIntegrator::updateBlocks (tracking_integrator.cpp:71-104) */
void ref_update_tracking(RefMap* r, uint64_t stamp) {
  hydra::InputData in;
  in.timestamp_ns = stamp;
  const khronos::FrameData data(in);
  r->tracking->updateBlocks(data, *r->map);
}

/* TrackingIntegrator::resetInactive (tracking_integrator.cpp:106-131); removed indices sorted */
int64_t ref_reset_inactive(RefMap* r, int32_t* removed, int64_t cap) {
  spatial_hash::BlockIndices rem;
  r->tracking->resetInactive(*r->map, &rem);
  std::sort(rem.begin(), rem.end(), [](const hydra::BlockIndex& a, const hydra::BlockIndex& b) {
    return a[0] != b[0] ? a[0] < b[0] : (a[1] != b[1] ? a[1] < b[1] : a[2] < b[2]);
  });
  for (int64_t i = 0; i < static_cast<int64_t>(rem.size()) && i < cap; ++i)
    for (int a = 0; a < 3; ++a) removed[3 * i + a] = rem[i][a];
  return static_cast<int64_t>(rem.size());
}

int64_t ref_num_blocks(const RefMap* r) { return static_cast<int64_t>(r->map->getTsdfLayer().numBlocks()); }

int64_t ref_block_indices(const RefMap* r, int32_t* out, int64_t cap) {
  const auto v = sortedIndices(*r);
  for (int64_t i = 0; i < static_cast<int64_t>(v.size()) && i < cap; ++i)
    for (int a = 0; a < 3; ++a) out[3 * i + a] = v[i][a];
  return static_cast<int64_t>(v.size());
}

/* flags: bit0 active, bit1 ever_free, bit2 to_remove (as orc_get_block); block_flags: bit2 tracking_updated, bit3 has_active_data */
int ref_get_block(const RefMap* r, const int32_t* idx, uint64_t* last_observed, uint64_t* last_occupied, uint8_t* flags, uint8_t* block_flags) {
  const hydra::BlockIndex bi(idx[0], idx[1], idx[2]);
  const auto tsdf = r->map->getTsdfLayer().getBlockPtr(bi);
  const auto trk = r->map->getTrackingLayer()->getBlockPtr(bi);
  if (!tsdf || !trk) return -1;
  for (size_t i = 0; i < trk->numVoxels(); ++i) {
    const hydra::TrackingVoxel& v = trk->getVoxel(i);
    if (last_observed) last_observed[i] = v.last_observed;
    if (last_occupied) last_occupied[i] = v.last_occupied;
    if (flags) flags[i] = static_cast<uint8_t>((v.active ? 1 : 0) | (v.ever_free ? 2 : 0) | (v.to_remove ? 4 : 0));
  }
  if (block_flags) *block_flags = static_cast<uint8_t>((tsdf->tracking_updated ? 4 : 0) | (trk->has_active_data ? 8 : 0));
  return 0;
}

/* FreeSpaceMotionDetector::processInput (free_space_motion_detector.cpp:73-103) on this side's map.  range: H*W, vertex: H*W*3
 * world-frame vertex map (the input conversion is external).  dynamic_out: H*W cluster ids (0 = static);
 * bbox_out: up to cap_clusters x 6 floats (min, max) of the clusters' bounding boxes.  returns the number of clusters. */
int ref_detect_motion(RefMap* r, int W, int H, uint64_t stamp, double sensor_z, const float* range, const float* vertex,
                      int32_t* dynamic_out, int64_t* n_seeds_out, int64_t* n_cluster_pixels_out, float* bbox_out, int cap_clusters,
                      float* centroid_out) {
  hydra::InputData in;
  in.timestamp_ns = stamp;
  in.range_image = cv::Mat(H, W, sizeof(float));
  in.vertex_map = cv::Mat(H, W, sizeof(cv::Vec3f));
  std::memcpy(in.range_image.data(), range, sizeof(float) * static_cast<size_t>(W) * H);
  std::memcpy(in.vertex_map.data(), vertex, sizeof(float) * 3 * static_cast<size_t>(W) * H);
  in.world_T_sensor.translation() = Eigen::Isometry3d::Vec(0.0, 0.0, sensor_z);
  khronos::FrameData data(in);
  data.dynamic_image = cv::Mat(H, W, sizeof(int));   // zero CV_32SC1 images (active_window.cpp:283-284)
  data.object_image = cv::Mat(H, W, sizeof(int));
  const hydra::VolumetricMap& cmap = *r->map;
  r->motion->processInput(cmap, data);
  if (n_seeds_out) {  // the seed set of the same frame (processInput keeps it to itself)
    khronos::GlobalIndexSet seeds;
    khronos::FreeSpaceMotionDetector::BlockToPointsMap pm;
    r->motion->setUpPointMap(data, *cmap.getTrackingLayer(), pm, seeds);
    *n_seeds_out = static_cast<int64_t>(seeds.size());
  }
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) dynamic_out[v * W + u] = data.dynamic_image.at<int>(v, u);
  int k = 0;
  for (const auto& cl : data.dynamic_clusters) {
    if (k < cap_clusters) {
      if (n_cluster_pixels_out) n_cluster_pixels_out[k] = static_cast<int64_t>(cl.pixels.size());
      if (bbox_out)
        for (int a = 0; a < 3; ++a) {
          bbox_out[6 * k + a] = cl.bounding_box.min[a];
          bbox_out[6 * k + 3 + a] = cl.bounding_box.max[a];
        }
      if (centroid_out) {  // the centroid as extractDynamicObject forms it (mesh_object_extractor.cpp:136-147): over cluster.pixels
        khronos::Points points;
        for (const khronos::Pixel& px : cl.pixels) {
          const auto& v = data.input.vertex_map.at<hydra::InputData::VertexType>(px.v, px.u);
          points.emplace_back(v[0], v[1], v[2]);
        }
        const khronos::Point c = khronos::utils::computeCentroid(points);
        for (int a = 0; a < 3; ++a) centroid_out[3 * k + a] = c[a];
      }
    }
    ++k;
  }
  return k;
}

/* ConnectedSemantics::processInput (connected_semantics.cpp:59-216).  label: H*W int32; object_labels: the ids for which
 * LabelSpaceConfig::isObject holds.  object_out: H*W cluster ids; per cluster (up to cap): semantic id, pixel count, bounding box
 * of its vertices (what MaxIoUTracker builds from it, max_iou_tracker.cpp:466-476).  returns the number of clusters. */
int ref_detect_objects(int W, int H, const float* range, const float* vertex, const int32_t* label, const int32_t* object_labels,
                       int n_object_labels, int use_full_connectivity, int min_cluster_size, int max_cluster_size, int use_3d,
                       float grid_size, float max_range, int32_t* object_out, int32_t* semantic_ids_out, int64_t* n_pixels_out,
                       float* bbox_out, int cap) {
  auto& labels = hydra::GlobalInfo::instance().mutableLabelSpaceConfig().object_labels;
  labels.clear();
  labels.insert(object_labels, object_labels + n_object_labels);
  khronos::ConnectedSemantics::Config oc;
  oc.use_full_connectivity = use_full_connectivity != 0;
  oc.min_cluster_size = min_cluster_size;
  oc.max_cluster_size = max_cluster_size;
  oc.use_3d = use_3d != 0;
  oc.grid_size = grid_size;
  oc.max_range = max_range;
  khronos::ConnectedSemantics detector(oc);
  hydra::InputData in;
  in.range_image = cv::Mat(H, W, sizeof(float));
  in.vertex_map = cv::Mat(H, W, sizeof(cv::Vec3f));
  in.label_image = cv::Mat(H, W, sizeof(int));
  std::memcpy(in.range_image.data(), range, sizeof(float) * static_cast<size_t>(W) * H);
  std::memcpy(in.vertex_map.data(), vertex, sizeof(float) * 3 * static_cast<size_t>(W) * H);
  std::memcpy(in.label_image.data(), label, sizeof(int32_t) * static_cast<size_t>(W) * H);
  khronos::FrameData data(in);
  data.dynamic_image = cv::Mat(H, W, sizeof(int));
  data.object_image = cv::Mat(H, W, sizeof(int));
  hydra::VolumetricMap::Config mc;
  const hydra::VolumetricMap map(mc);  // (unused by the detector, connected_semantics.cpp:59)
  detector.processInput(map, data);
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) object_out[v * W + u] = data.object_image.at<int>(v, u);
  int k = 0;
  for (const auto& cl : data.semantic_clusters) {
    if (k < cap) {
      semantic_ids_out[2 * k] = cl.id;
      semantic_ids_out[2 * k + 1] = cl.semantics ? cl.semantics->category_id : -1;
      n_pixels_out[k] = static_cast<int64_t>(cl.pixels.size());
      const khronos::BoundingBox box(khronos::utils::VertexMapAdaptor(cl.pixels, data.input.vertex_map));
      for (int a = 0; a < 3; ++a) {
        bbox_out[6 * k + a] = box.min[a];
        bbox_out[6 * k + 3 + a] = box.max[a];
      }
    }
    ++k;
  }
  return k;
}

/* MaxIoUTracker::processInput (max_iou_tracker.cpp:198-593) over a scenario in the format of khronos_amd/host/host_selftest.cpp's
 * --tracker replay (frames of semantic / dynamic clusters given as voxel sets and boxes); the track list after every frame as
 * the same JSON lines.  The reference derives a cluster's voxels and box from its pixels' vertices (:444-487): a cluster
 * becomes one pixel per voxel at the voxel's centre (track_by voxels) or two pixels at the box corners (track_by bounding_box).
 * returns the length of the output (truncated to cap). */
int64_t ref_tracker_replay(const char* scenario, char* out, int64_t cap) {
  std::istringstream in(scenario);
  std::string result, tok;
  std::unique_ptr<khronos::Tracker> tracker;
  bool by_voxels = true;
  float voxel_size = 0.2f;
  struct Cl {
    bool semantic;
    int id, cat;
    float lo[3], hi[3];
    std::vector<std::array<int64_t, 3>> voxels;
    std::vector<std::array<int, 2>> pixels;       // track_by pixels: the cluster's pixels and their world-frame vertices
    std::vector<std::array<float, 3>> points;
  };
  std::vector<Cl> clusters;
  uint64_t stamp = 0;
  bool by_pixels = false;
  int img_w = 1, img_h = 1;
  hydra::Sensor sensor;
  double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  while (in >> tok) {
    if (tok == "C") {
      std::string kind, by, assoc;
      khronos::MaxIoUTracker::Config c;
      in >> kind >> by >> assoc >> c.min_semantic_iou >> c.min_cosine_sim >> c.min_cross_iou >> c.max_dynamic_distance >> c.temporal_window >>
          c.min_num_observations >> c.voxel_size;
      by_voxels = by == "voxels";
      by_pixels = by == "pixels";
      voxel_size = c.voxel_size;
      c.track_by = by_pixels ? khronos::MaxIoUTracker::Config::TrackBy::kPixels
                             : (by_voxels ? khronos::MaxIoUTracker::Config::TrackBy::kVoxels : khronos::MaxIoUTracker::Config::TrackBy::kBouningBox);
      c.semantic_association = assoc == "assign_track" ? khronos::MaxIoUTracker::Config::SemanticAssociation::kAssignTrack
                                                       : khronos::MaxIoUTracker::Config::SemanticAssociation::kAssignCluster;
      if (kind == "external") {  // external_tracker.cpp:59-143
        khronos::ExternalTracker::Config e;
        e.temporal_window = c.temporal_window;
        e.min_num_observations = c.min_num_observations;
        tracker = std::make_unique<khronos::ExternalTracker>(e);
      } else {
        tracker = std::make_unique<khronos::MaxIoUTracker>(c);
      }
    } else if (tok == "F") {
      in >> stamp;
      clusters.clear();
    } else if (tok == "I") {  // image size and intrinsics (track_by pixels)
      in >> img_w >> img_h >> sensor.fx >> sensor.fy >> sensor.cx >> sensor.cy;
      sensor.width = img_w;
      sensor.height = img_h;
    } else if (tok == "T") {  // the frame's world_T_sensor, row-major
      for (double& v : pose) in >> v;
    } else if (tok == "SP" || tok == "DP") {  // a cluster as pixels + vertices: <id> [<category>] <n> (<u> <v> <x> <y> <z>)*
      Cl c;
      c.semantic = tok == "SP";
      c.cat = -1;
      in >> c.id;
      if (c.semantic) in >> c.cat;
      size_t n;
      in >> n;
      c.pixels.resize(n);
      c.points.resize(n);
      for (size_t i = 0; i < n; ++i) in >> c.pixels[i][0] >> c.pixels[i][1] >> c.points[i][0] >> c.points[i][1] >> c.points[i][2];
      clusters.push_back(std::move(c));
    } else if (tok == "S" || tok == "D") {
      Cl c;
      c.semantic = tok == "S";
      c.cat = -1;
      in >> c.id;
      if (c.semantic) in >> c.cat;
      for (float& v : c.lo) in >> v;
      for (float& v : c.hi) in >> v;
      size_t n;
      in >> n;
      c.voxels.resize(n);
      for (auto& v : c.voxels) in >> v[0] >> v[1] >> v[2];
      clusters.push_back(std::move(c));
    } else if (tok == "E") {
      // one image row holding every cluster's pixels
      size_t n_px = 0;
      for (const Cl& c : clusters) n_px += by_voxels ? c.voxels.size() : 2;
      hydra::InputData input;
      input.timestamp_ns = stamp;
      if (by_pixels) {
        input.vertex_map = cv::Mat(img_h, img_w, sizeof(cv::Vec3f));
        input.sensor = sensor;
        for (int r = 0; r < 3; ++r) {
          for (int cc = 0; cc < 3; ++cc) input.world_T_sensor.linear(r, cc) = pose[4 * r + cc];
          input.world_T_sensor.translation()[r] = pose[4 * r + 3];
        }
      } else {
        input.vertex_map = cv::Mat(1, static_cast<int>(std::max<size_t>(n_px, 1)), sizeof(cv::Vec3f));
      }
      khronos::FrameData data(input);
      const spatial_hash::Grid<khronos::GlobalIndex> grid(voxel_size);
      cv::Mat vm = data.input.vertex_map;  // (shares the pixels)
      int u = 0;
      for (const Cl& c : clusters) {
        khronos::MeasurementCluster mc;
        mc.id = c.id;
        if (c.semantic) mc.semantics = khronos::SemanticClusterInfo(c.cat);
        auto put = [&](const khronos::Point& p) {
          cv::Vec3f& v = vm.at<cv::Vec3f>(0, u);
          v[0] = p[0], v[1] = p[1], v[2] = p[2];
          mc.pixels.emplace_back(u++, 0);
        };
        if (by_pixels) {
          for (size_t i = 0; i < c.pixels.size(); ++i) {
            cv::Vec3f& v = vm.at<cv::Vec3f>(c.pixels[i][1], c.pixels[i][0]);
            v[0] = c.points[i][0], v[1] = c.points[i][1], v[2] = c.points[i][2];
            mc.pixels.emplace_back(c.pixels[i][0], c.pixels[i][1]);
          }
        } else if (by_voxels) {
          for (const auto& v : c.voxels) put(grid.toPoint(khronos::GlobalIndex(v[0], v[1], v[2])));
        } else {
          put(khronos::Point(c.lo[0], c.lo[1], c.lo[2]));
          put(khronos::Point(c.hi[0], c.hi[1], c.hi[2]));
        }
        (c.semantic ? data.semantic_clusters : data.dynamic_clusters).push_back(std::move(mc));
      }
      tracker->processInput(data);
      result += "[";
      bool first = true;
      for (const khronos::Track& t : tracker->getTracks()) {
        const khronos::Observation& o = t.observations.back();
        char buf[512];
        std::snprintf(buf, sizeof(buf),
                      "%s{\"id\": %d, \"dyn\": %d, \"active\": %d, \"conf\": %.9g, \"first\": %llu, \"last\": %llu, \"cat\": %d, "
                      "\"n_obs\": %zu, \"obs\": [%llu, %d, %d], \"n_vox\": %zu, \"n_pts\": %zu, \"centroid\": [%.9g, %.9g, %.9g]}",
                      first ? "" : ", ", t.id, int(t.is_dynamic), int(t.is_active), t.confidence, static_cast<unsigned long long>(t.first_seen),
                      static_cast<unsigned long long>(t.last_seen), t.semantics ? t.semantics->category_id : -1, t.observations.size(),
                      static_cast<unsigned long long>(o.stamp), o.semantic_cluster_id, o.dynamic_cluster_id, t.last_voxels.size(), t.last_points.size(),
                      t.is_dynamic ? t.last_centroid[0] : 0.f, t.is_dynamic ? t.last_centroid[1] : 0.f, t.is_dynamic ? t.last_centroid[2] : 0.f);
        result += buf;
        first = false;
      }
      result += "]\n";
    }
  }
  const int64_t n = std::min<int64_t>(static_cast<int64_t>(result.size()), cap > 0 ? cap - 1 : 0);
  if (cap > 0) {
    std::memcpy(out, result.data(), static_cast<size_t>(n));
    out[n] = 0;
  }
  return static_cast<int64_t>(result.size());
}

/* FrameDataBuffer (frame_data_buffer.cpp:57-123) over a script in the format of host_selftest --buffer; the same lines. */
int64_t ref_buffer_replay(const char* script, char* out, int64_t cap) {
  std::istringstream in(script);
  std::string result, tok;
  std::unique_ptr<khronos::FrameDataBuffer> buffer;
  std::vector<uint64_t> stamps;
  while (in >> tok) {
    if (tok == "B") {
      khronos::FrameDataBuffer::Config c;
      in >> c.max_buffer_size >> c.store_every_n_frames;
      buffer = std::make_unique<khronos::FrameDataBuffer>(c);
      continue;
    }
    if (tok == "S") {
      hydra::InputData input;
      in >> input.timestamp_ns;
      stamps.push_back(input.timestamp_ns);
      buffer->storeData(std::make_shared<khronos::FrameData>(input));
    } else if (tok == "T") {
      size_t n_tracks;
      in >> n_tracks;
      khronos::Tracks tracks(n_tracks);
      for (khronos::Track& t : tracks) {
        size_t n_obs;
        in >> n_obs;
        for (size_t i = 0; i < n_obs; ++i) {
          uint64_t st;
          in >> st;
          t.observations.emplace_back(st);
        }
      }
      buffer->trimBuffer(tracks);
    }
    result += std::to_string(buffer->size()) + " " + std::to_string(buffer->size() ? buffer->getLatestData().input.timestamp_ns : 0);
    for (uint64_t st : stamps) result += buffer->getData(st) ? " 1" : " 0";
    result += "\n";
  }
  const int64_t n = std::min<int64_t>(static_cast<int64_t>(result.size()), cap > 0 ? cap - 1 : 0);
  if (cap > 0) {
    std::memcpy(out, result.data(), static_cast<size_t>(n));
    out[n] = 0;
  }
  return static_cast<int64_t>(result.size());
}

/* RayVerificator (ray_verificator.cpp:66-145,212-349) over rays given as arrays.  The reference finds a ray's source in the scene
 * graph's agent layer and its target in the graph's mesh: ray i becomes agent node i (position = source, timestamp = stamp) and
 * mesh vertex i (= target, first seen one nanosecond before the stamp), and with ray_policy First each vertex draws exactly the
 * ray from its own agent node (computeVertexSources, :275-283).  Stamps must be ascending and distinct. */
struct RefRayVerificator {
  std::shared_ptr<spark_dsg::DynamicSceneGraph> dsg;
  std::shared_ptr<khronos::RayVerificator> rv;
};

RefRayVerificator* ref_rv_create(float block_size, float radial_tolerance, float depth_tolerance, int64_t n, const uint64_t* stamps,
                                 const float* sources, const float* targets) {
  auto* r = new RefRayVerificator();
  khronos::RayVerificator::Config c;
  c.block_size = block_size;
  c.radial_tolerance = radial_tolerance;
  c.depth_tolerance = depth_tolerance;
  c.ray_policy = khronos::RayVerificator::Config::RayPolicy::kFirst;
  r->rv = std::make_shared<khronos::RayVerificator>(c);
  r->dsg = std::make_shared<spark_dsg::DynamicSceneGraph>();
  auto& agents = r->dsg->layers[{r->dsg->layer_ids.at(spark_dsg::DsgLayers::AGENTS), c.prefix.key}];
  r->dsg->mesh_ = std::make_shared<spark_dsg::Mesh>();
  for (int64_t i = 0; i < n; ++i) {
    auto attrs = std::make_unique<spark_dsg::AgentNodeAttributes>();
    attrs->position = Eigen::Vector3d(sources[3 * i], sources[3 * i + 1], sources[3 * i + 2]);
    attrs->timestamp = std::chrono::nanoseconds(static_cast<int64_t>(stamps[i]));
    auto node = std::make_unique<spark_dsg::SceneGraphNode>();
    node->attrs = std::move(attrs);
    agents.nodes_[static_cast<spark_dsg::NodeId>(i)] = std::move(node);
    r->dsg->mesh_->points.emplace_back(targets[3 * i], targets[3 * i + 1], targets[3 * i + 2]);
    r->dsg->mesh_->first_seen_stamps.push_back(stamps[i] - 1);
    r->dsg->mesh_->stamps.push_back(stamps[i]);
  }
  r->rv->setDsg(r->dsg);
  return r;
}

void ref_rv_destroy(RefRayVerificator* r) { delete r; }

/* RayVerificator::check; the stamps of the two lists in ascending order (the reference walks an unordered_set of rays) */
void ref_rv_check(const RefRayVerificator* r, const float* point, uint64_t earliest, uint64_t latest, uint64_t* present, int64_t cap_present,
                  int64_t* n_present, uint64_t* absent, int64_t cap_absent, int64_t* n_absent) {
  auto res = r->rv->check(khronos::Point(point[0], point[1], point[2]), earliest, latest);
  std::sort(res.present.begin(), res.present.end());
  std::sort(res.absent.begin(), res.absent.end());
  *n_present = static_cast<int64_t>(res.present.size());
  *n_absent = static_cast<int64_t>(res.absent.size());
  for (int64_t i = 0; i < *n_present && i < cap_present; ++i) present[i] = res.present[i];
  for (int64_t i = 0; i < *n_absent && i < cap_absent; ++i) absent[i] = res.absent[i];
}

/* the two callers of the ray verificator (SURVEY.md section 8 f4 tail).  vote: {temporal_resolution, window_size,
 * use_relative_confidence, absence_confidence, presence_confidence}.
 * RayBackgroundChangeDetector::detectChanges (ray_background_change_detector.cpp:59-88): states of the first n_prev vertices given,
 * the others are new; re-observed vertices are recomputed.  states_out: n_vertices entries (0 unobserved, 1 persistent, 2 absent) */
static std::shared_ptr<khronos::RayChangeDetector> makeVote(const float* vote) {
  khronos::RayChangeDetector::Config c;
  c.temporal_resolution = vote[0];
  c.window_size = static_cast<size_t>(vote[1]);
  c.use_relative_confidence = vote[2] != 0.f;
  c.absence_confidence = vote[3];
  c.presence_confidence = vote[4];
  return std::make_shared<khronos::RayChangeDetector>(c);
}

static uint8_t stateCode(khronos::ChangeState s) {
  return s == khronos::ChangeState::kAbsent ? 2 : (s == khronos::ChangeState::kPersistent ? 1 : 0);
}

void ref_cd_background(RefRayVerificator* r, const float* vote, float time_filtering_threshold, int64_t n_vertices, const float* points,
                       const uint64_t* stamps, const uint8_t* prev_states, int64_t n_prev, const int64_t* reobserved, int64_t n_reobserved,
                       uint8_t* states_out) {
  khronos::RayBackgroundChangeDetector::Config c;
  c.time_filtering_threshold = time_filtering_threshold;
  khronos::RayBackgroundChangeDetector detector(c, r->rv, makeVote(vote));
  spark_dsg::DynamicSceneGraph dsg;
  dsg.mesh_ = std::make_shared<spark_dsg::Mesh>();
  for (int64_t i = 0; i < n_vertices; ++i) {
    dsg.mesh_->points.emplace_back(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    dsg.mesh_->stamps.push_back(stamps[i]);
    dsg.mesh_->first_seen_stamps.push_back(stamps[i]);
  }
  khronos::BackgroundChanges changes;
  for (int64_t i = 0; i < n_prev; ++i)
    changes.push_back(prev_states[i] == 2 ? khronos::ChangeState::kAbsent : (prev_states[i] == 1 ? khronos::ChangeState::kPersistent : khronos::ChangeState::kUnobserved));
  r->rv->reobserved_vertices_.clear();  // (RayVerificator::updateDsg fills this; given here)
  for (int64_t i = 0; i < n_reobserved; ++i) r->rv->reobserved_vertices_.insert(static_cast<size_t>(reobserved[i]));
  detector.detectChanges(dsg, changes);
  r->rv->reobserved_vertices_.clear();
  for (int64_t i = 0; i < n_vertices; ++i) states_out[i] = stateCode(changes.at(static_cast<size_t>(i)));
}

/* RayObjectChangeDetector::detectChanges for ONE object node (ray_object_change_detector.cpp:62-160): mesh in its box frame, box,
 * first / last observed; merges as (from, to, valid) triples.  out: {merged_id, first_absent, last_absent, first_persistent,
 * last_persistent}; returns 0 when the object got no change entry (dynamic objects, :86-89) */
int ref_cd_object(RefRayVerificator* r, const float* vote, float time_filtering_threshold, int query_subsampling, uint64_t node_id, int64_t n_points,
                  const float* local_points, const float* bbox_min, const float* bbox_max, uint64_t first_observed, uint64_t last_observed,
                  int is_dynamic, const uint64_t* merges, int64_t n_merges, uint64_t* out) {
  khronos::RayObjectChangeDetector::Config c;
  c.time_filtering_threshold = time_filtering_threshold;
  c.query_subsampling = query_subsampling;
  khronos::RayObjectChangeDetector detector(c, r->rv, makeVote(vote));
  spark_dsg::DynamicSceneGraph dsg;
  auto attrs = std::make_unique<spark_dsg::KhronosObjectAttributes>();
  for (int64_t i = 0; i < n_points; ++i) attrs->mesh.points.emplace_back(local_points[3 * i], local_points[3 * i + 1], local_points[3 * i + 2]);
  attrs->bounding_box.include(Eigen::Vector3f(bbox_min[0], bbox_min[1], bbox_min[2]));
  attrs->bounding_box.include(Eigen::Vector3f(bbox_max[0], bbox_max[1], bbox_max[2]));
  attrs->bounding_box.finish();
  attrs->first_observed_ns = {first_observed};
  attrs->last_observed_ns = {last_observed};
  if (is_dynamic) attrs->trajectory_positions.emplace_back(0.f, 0.f, 0.f);
  auto node = std::make_unique<spark_dsg::SceneGraphNode>();
  node->attrs = std::move(attrs);
  dsg.layers[{dsg.layer_ids.at(spark_dsg::DsgLayers::OBJECTS), 0}].nodes_[node_id] = std::move(node);
  khronos::RPGOMerges rpgo;
  for (int64_t i = 0; i < n_merges; ++i) rpgo.emplace_back(merges[3 * i], merges[3 * i + 1], merges[3 * i + 2] != 0);
  khronos::ObjectChanges changes;
  detector.detectChanges(dsg, rpgo, changes);
  if (changes.empty()) return 0;
  const khronos::ObjectChange& ch = changes.front();
  out[0] = ch.merged_id;
  out[1] = ch.first_absent;
  out[2] = ch.last_absent;
  out[3] = ch.first_persistent;
  out[4] = ch.last_persistent;
  return 1;
}

/* RayVerificator::computeVertexSources (ray_verificator.cpp:266-325; private, reached with -fno-access-control): which of the
 * pose stamps a vertex seen from first_seen to last_seen draws rays from, for the deterministic policies (0 First, 1 Last,
 * 2 FirstAndLast, 3 Middle, 4 All).  returns the count, indices ascending */
int64_t ref_rv_vertex_sources(int policy, const uint64_t* pose_stamps, int64_t n_poses, uint64_t first_seen, uint64_t last_seen, int64_t* out, int64_t cap) {
  khronos::RayVerificator::Config c;
  c.ray_policy = static_cast<khronos::RayVerificator::Config::RayPolicy>(policy);
  khronos::RayVerificator rv(c);
  rv.timestamps_.assign(pose_stamps, pose_stamps + n_poses);
  const auto res = rv.computeVertexSources(first_seen, last_seen);
  std::vector<size_t> v(res.begin(), res.end());
  std::sort(v.begin(), v.end());
  for (int64_t i = 0; i < static_cast<int64_t>(v.size()) && i < cap; ++i) out[i] = static_cast<int64_t>(v[i]);
  return static_cast<int64_t>(v.size());
}

/* RayChangeDetector::detectChanges (ray_change_detector.cpp:66-133).  out: {has closest_absent, closest_absent, has
 * furthest_persistent, furthest_persistent} */
void ref_detect_changes(float temporal_resolution, int64_t window_size, int use_relative_confidence, float absence_confidence,
                        float presence_confidence, const uint64_t* present, int64_t n_present, const uint64_t* absent, int64_t n_absent,
                        int forward, uint64_t* out) {
  khronos::RayChangeDetector::Config c;
  c.temporal_resolution = temporal_resolution;
  c.window_size = static_cast<size_t>(window_size);
  c.use_relative_confidence = use_relative_confidence != 0;
  c.absence_confidence = absence_confidence;
  c.presence_confidence = presence_confidence;
  const khronos::RayChangeDetector detector(c);
  khronos::RayVerificator::CheckResult check;
  check.present.assign(present, present + n_present);
  check.absent.assign(absent, absent + n_absent);
  const auto res = detector.detectChanges(check, forward != 0);
  out[0] = res.closest_absent.has_value();
  out[1] = res.closest_absent.value_or(0);
  out[2] = res.furthest_persistent.has_value();
  out[3] = res.furthest_persistent.value_or(0);
}

/* MeshObjectExtractor::extractObject (mesh_object_extractor.cpp:81-356): the reference's own extraction glue -- track validity,
 * frame collection, extent merge, volume gates, object-map sizing and block allocation, the confidence pruning loop, bounding box,
 * shift to the box frame -- around the two integrators it drives.  Those two are not in /root/reference: ref_standin::bridge()
 * hands ProjectiveIntegrator::updateMap / MeshIntegrator::generateMesh to the CPU oracle (ASSUMPTIONS.md A.3 - A.5), keeping the
 * stand-in map's voxels in step with it, so that the reference's loops read and write real data.  (This file is compiled with
 * -fno-access-control: the bridge reads ObjectIntegrator's private frame pointer and target id.) */
// what the bridge needs to stand behind a stand-in map: the oracle configurations of the window's map and of the object maps,
// and the sensor.  One environment is current at a time (the extraction workers are detached threads, so it is process-wide).
struct BridgeEnv {
  orc_config main_cfg;
  orc_config object_cfg;
  orc_sensor sensor;
};

struct RefExtractor {
  std::unique_ptr<khronos::MeshObjectExtractor> extractor;
  std::unique_ptr<khronos::FrameDataBuffer> buffer;
  BridgeEnv env;
};

namespace {
BridgeEnv* g_env = nullptr;

orc_map* backendOf(hydra::VolumetricMap& map) {
  if (!map.backend) {
    const bool window_map = map.config.with_tracking;  // (object maps: tracking off, mesh_object_extractor.cpp:211)
    orc_config c = window_map ? g_env->main_cfg : g_env->object_cfg;
    c.voxel_size = map.config.voxel_size;
    c.voxels_per_side = static_cast<int32_t>(map.config.voxels_per_side);
    c.truncation_distance = map.config.truncation_distance;
    c.with_semantics = map.config.with_semantics ? 1 : 0;
    c.with_tracking = map.config.with_tracking ? 1 : 0;
    if (!window_map) {
      c.num_labels = 2;  // BinarySemanticIntegrator (object_integrator.cpp:44-48)
      c.semantic_mode = 1;
    }
    c.rank = 0;
    c.world_size = 1;
    orc_map* m = orc_create(&c);
    map.backend = std::shared_ptr<void>(m, [](void* p) { orc_destroy(static_cast<orc_map*>(p)); });
    for (const auto& idx : map.getTsdfLayer().allocatedBlockIndices()) orc_allocate_block(m, idx[0], idx[1], idx[2]);
    map.on_remove = [m](const hydra::BlockIndex& i) { orc_remove_block(m, i[0], i[1], i[2]); };
  }
  return static_cast<orc_map*>(map.backend.get());
}

// the three block flags the reference's own code sets / clears outside the integrators go to the backend before a bridged call ...
void pushFlags(hydra::VolumetricMap& map, orc_map* m) {
  for (const hydra::TsdfBlock& tb : map.getTsdfLayer())
    orc_set_block_flags(m, tb.index[0], tb.index[1], tb.index[2], static_cast<uint8_t>((tb.updated ? 1 : 0) | (tb.mesh_updated ? 2 : 0) | (tb.tracking_updated ? 4 : 0)));
}

// ... and after it the stand-in map follows the backend: blocks, distance, weight, last_observed (window map) or the two counters
// (object maps), the flags
void pull(hydra::VolumetricMap& map, orc_map* m, bool voxels) {
  const int64_t nb = orc_num_blocks(m);
  std::vector<int32_t> idx(3 * std::max<int64_t>(nb, 1));
  orc_block_indices(m, idx.data(), nb);
  const size_t nv = map.config.voxels_per_side * map.config.voxels_per_side * map.config.voxels_per_side;
  const bool window_map = map.config.with_tracking;
  std::vector<float> dist(nv), weight(nv), lik(window_map ? 1 : 2 * nv);
  std::vector<uint64_t> last_obs(nv);
  std::vector<uint8_t> flags(nv);
  for (int64_t b = 0; b < nb; ++b) {
    const hydra::BlockIndex bi(idx[3 * b], idx[3 * b + 1], idx[3 * b + 2]);
    if (!map.getTsdfLayer().hasBlock(bi)) map.allocateBlock(bi);
    auto tb = map.getTsdfLayer().getBlockPtr(bi);
    uint8_t bf = 0;
    if (!voxels) {
      orc_get_block(m, bi[0], bi[1], bi[2], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &bf);
    } else if (window_map) {
      orc_get_block(m, bi[0], bi[1], bi[2], dist.data(), weight.data(), nullptr, last_obs.data(), nullptr, nullptr, nullptr, nullptr, &bf);
      auto kb = map.getTrackingLayer()->getBlockPtr(bi);
      for (size_t i = 0; i < nv; ++i) {
        tb->getVoxel(i).distance = dist[i];
        tb->getVoxel(i).weight = weight[i];
        kb->getVoxel(i).last_observed = last_obs[i];
      }
    } else {
      orc_get_block(m, bi[0], bi[1], bi[2], dist.data(), weight.data(), nullptr, nullptr, nullptr, flags.data(), nullptr, lik.data(), &bf);
      auto sb = map.getSemanticLayer()->getBlockPtr(bi);
      for (size_t i = 0; i < nv; ++i) {
        tb->getVoxel(i).distance = dist[i];
        tb->getVoxel(i).weight = weight[i];
        hydra::SemanticVoxel& sv = sb->getVoxel(i);
        sv.empty = (flags[i] & 8) == 0;
        sv.semantic_likelihoods(0) = lik[i];
        sv.semantic_likelihoods(1) = lik[nv + i];
      }
    }
    tb->updated = (bf & 1) != 0;
    tb->mesh_updated = (bf & 2) != 0;
    tb->tracking_updated = (bf & 4) != 0;
  }
}

void installBridge() {
  static bool done = false;
  if (done) return;
  done = true;
  ref_standin::bridge().integrate = [](const hydra::ProjectiveIntegrator& integrator, const hydra::InputData& data, hydra::VolumetricMap& map, bool allocate,
                                       const cv::Mat& mask) {
    orc_map* m = backendOf(map);
    pushFlags(map, m);
    orc_frame f{};
    f.timestamp_ns = data.timestamp_ns;
    std::memcpy(f.world_T_sensor, data.world_T_sensor16, sizeof(f.world_T_sensor));
    f.depth = data.depth->data();
    f.color = data.rgb ? data.rgb->data() : nullptr;
    f.object_id = -1;
    struct HookCtx {
      const khronos::ObjectIntegrator* integrator;
      const hydra::VolumetricMap::Config* map_config;
      const hydra::InputData* data;
      const cv::Mat* mask;
    } hook_ctx{nullptr, &map.config, &data, &mask};
    if (const auto* object_integrator = dynamic_cast<const khronos::ObjectIntegrator*>(&integrator)) {
      // round 5: the reference's OWN ObjectIntegrator::computeLabel (object_integrator.cpp:58-81) decides mask and label of every
      // measurement -- the oracle supplies sdf and interpolation weights and applies none of its own rules (orc_set_label_hook)
      hook_ctx.integrator = object_integrator;
      orc_set_label_hook(m, [](void* user, float sdf, const int32_t* u4, const int32_t* v4, const float* w4, int32_t* label_out) -> int {
        const HookCtx& h = *static_cast<const HookCtx*>(user);
        hydra::VoxelMeasurement meas;
        meas.sdf = sdf;
        for (int k = 0; k < 4; ++k) {
          meas.interpolation_weights.u[k] = u4[k];
          meas.interpolation_weights.v[k] = v4[k];
          meas.interpolation_weights.w[k] = w4[k];
        }
        meas.label = -1;
        g_label_hook_calls.fetch_add(1, std::memory_order_relaxed);
        if (!h.integrator->computeLabel(*h.map_config, *h.data, *h.mask, meas)) {
          g_label_hook_skips.fetch_add(1, std::memory_order_relaxed);
          return 0;
        }
        *label_out = meas.label;
        return 1;
      }, &hook_ctx);
      // (the oracle still needs the frame for colour; its own object-image rule is bypassed by the hook)
      f.object_image = reinterpret_cast<const int32_t*>(object_integrator->current_data_->object_image.data());
      f.object_id = object_integrator->current_object_id_;
    } else {  // the window's integrator: labels fused, dynamic pixels masked (active_window.cpp:207-210)
      f.label = data.label_image.empty() ? nullptr : reinterpret_cast<const int32_t*>(data.label_image.data());
      f.mask = mask.empty() ? nullptr : reinterpret_cast<const int32_t*>(mask.data());
    }
    orc_stats st{};
    orc_integrate(m, &g_env->sensor, &f, allocate ? 1 : 0, &st);
    orc_set_label_hook(m, nullptr, nullptr);
    pull(map, m, true);
  };
  ref_standin::bridge().mesh = [](const hydra::MeshIntegrator&, hydra::VolumetricMap& map, bool only_updated, bool clear) {
    orc_map* m = backendOf(map);
    pushFlags(map, m);
    if (!map.config.with_tracking) {  // object maps: what the reference's pruning loop did to the distances
      const size_t nv = map.config.voxels_per_side * map.config.voxels_per_side * map.config.voxels_per_side;
      std::vector<float> dist(nv);
      for (hydra::TsdfBlock& tb : map.getTsdfLayer()) {
        for (size_t i = 0; i < nv; ++i) dist[i] = tb.getVoxel(i).distance;
        orc_set_distance(m, tb.index[0], tb.index[1], tb.index[2], dist.data());
      }
    }
    orc_generate_mesh(m, only_updated ? 1 : 0, clear ? 1 : 0);
    pull(map, m, false);
    const int64_t n = orc_mesh_num_vertices(m);
    std::vector<float> pts(3 * std::max<int64_t>(n, 1));
    std::vector<uint8_t> col(4 * std::max<int64_t>(n, 1));
    std::vector<uint32_t> lab(std::max<int64_t>(n, 1));
    std::vector<uint64_t> fs(std::max<int64_t>(n, 1)), stp(std::max<int64_t>(n, 1));
    orc_mesh_copy(m, pts.data(), col.data(), lab.data(), fs.data(), stp.data(), n);
    hydra::MeshBlock mb;
    for (int64_t i = 0; i < n; ++i) {
      mb.points.emplace_back(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
      mb.colors.push_back(spark_dsg::Color{col[4 * i], col[4 * i + 1], col[4 * i + 2], col[4 * i + 3]});
      mb.labels.push_back(lab[i]);
      mb.first_seen_stamps.push_back(fs[i]);
      mb.stamps.push_back(stp[i]);
      if (i % 3 == 2) mb.faces.push_back({static_cast<size_t>(i - 2), static_cast<size_t>(i - 1), static_cast<size_t>(i)});
    }
    map.getMeshLayer().clear();
    map.getMeshLayer().push_back(std::move(mb));
  };
  ref_standin::bridge().parse_input = [](hydra::InputData& d) {
    orc_parse_input(&g_env->main_cfg, &g_env->sensor, d.world_T_sensor16, d.depth->data(), reinterpret_cast<float*>(d.range_image.data()),
                    reinterpret_cast<float*>(d.vertex_map.data()));
  };
}
}  // namespace

RefExtractor* ref_ex_create(const orc_config* object_map_cfg, const orc_sensor* sensor, float min_object_allocation_confidence, float min_object_volume,
                            float max_object_volume, int only_extract_reconstructed_objects, float min_dynamic_displacement,
                            float min_object_reconstruction_confidence, int min_object_reconstruction_observations,
                            float object_reconstruction_resolution, float min_reconstruction_resolution, int64_t max_buffer_size) {
  installBridge();
  auto* r = new RefExtractor();
  khronos::MeshObjectExtractor::Config c;
  c.min_object_allocation_confidence = min_object_allocation_confidence;
  c.min_object_volume = min_object_volume;
  c.max_object_volume = max_object_volume;
  c.only_extract_reconstructed_objects = only_extract_reconstructed_objects != 0;
  c.min_dynamic_displacement = min_dynamic_displacement;
  c.min_object_reconstruction_confidence = min_object_reconstruction_confidence;
  c.min_object_reconstruction_observations = min_object_reconstruction_observations;
  c.object_reconstruction_resolution = object_reconstruction_resolution;
  c.min_reconstruction_resolution = min_reconstruction_resolution;
  r->extractor = std::make_unique<khronos::MeshObjectExtractor>(c);
  khronos::FrameDataBuffer::Config bc;
  bc.max_buffer_size = static_cast<size_t>(max_buffer_size);
  r->buffer = std::make_unique<khronos::FrameDataBuffer>(bc);
  r->env.main_cfg = *object_map_cfg;
  r->env.object_cfg = *object_map_cfg;
  r->env.sensor = *sensor;
  return r;
}

void ref_ex_destroy(RefExtractor* r) { delete r; }

/* one frame into the reference's FrameDataBuffer: raw images for the bridge, the object image, the semantic clusters' ids and boxes */
void ref_ex_add_frame(RefExtractor* r, uint64_t stamp, const double* world_T_sensor, const float* depth, const uint8_t* rgb,
                      const int32_t* object_image, int n_clusters, const int32_t* cluster_ids, const float* cluster_boxes) {
  const int W = r->env.sensor.width, H = r->env.sensor.height;
  hydra::InputData in;
  in.timestamp_ns = stamp;
  in.depth = std::make_shared<std::vector<float>>(depth, depth + static_cast<size_t>(W) * H);
  if (rgb) in.rgb = std::make_shared<std::vector<uint8_t>>(rgb, rgb + 3 * static_cast<size_t>(W) * H);
  std::memcpy(in.world_T_sensor16, world_T_sensor, sizeof(in.world_T_sensor16));
  auto fd = std::make_shared<khronos::FrameData>(in);
  fd->dynamic_image = cv::Mat(H, W, sizeof(int));
  fd->object_image = cv::Mat(H, W, sizeof(int));
  std::memcpy(fd->object_image.data(), object_image, sizeof(int32_t) * static_cast<size_t>(W) * H);
  for (int k = 0; k < n_clusters; ++k) {
    khronos::MeasurementCluster mc;
    mc.id = cluster_ids[k];
    mc.bounding_box.include(khronos::Point(cluster_boxes[6 * k], cluster_boxes[6 * k + 1], cluster_boxes[6 * k + 2]));
    mc.bounding_box.include(khronos::Point(cluster_boxes[6 * k + 3], cluster_boxes[6 * k + 4], cluster_boxes[6 * k + 5]));
    mc.bounding_box.finish();
    fd->semantic_clusters.push_back(std::move(mc));
  }
  r->buffer->storeData(fd);
}

/* extractObject for one track.  returns 0 (no object) or 1; points (shifted to the box frame) up to cap, their number, the box
 * (min, max), label, first / last observed */
int ref_ex_extract(RefExtractor* r, int track_id, int is_dynamic, float confidence, uint64_t first_seen, uint64_t last_seen, int category, int n_obs,
                   const uint64_t* obs_stamps, const int32_t* obs_semantic_ids, const int32_t* obs_dynamic_ids, float* points_out, int64_t cap_points,
                   int64_t* n_points_out, float* bbox_out, int64_t* info_out) {
  g_env = &r->env;
  khronos::Track t;
  t.id = track_id;
  t.is_dynamic = is_dynamic != 0;
  t.confidence = confidence;
  t.first_seen = first_seen;
  t.last_seen = last_seen;
  if (category >= 0) t.semantics = khronos::SemanticClusterInfo(category);
  for (int i = 0; i < n_obs; ++i) t.observations.emplace_back(obs_stamps[i], obs_semantic_ids[i], obs_dynamic_ids[i]);
  const auto obj = r->extractor->extractObject(t, *r->buffer);
  if (!obj) return 0;
  *n_points_out = static_cast<int64_t>(obj->mesh.points.size());
  for (int64_t i = 0; i < *n_points_out && i < cap_points; ++i)
    for (int a = 0; a < 3; ++a) points_out[3 * i + a] = obj->mesh.points[i][a];
  for (int a = 0; a < 3; ++a) {
    bbox_out[a] = obj->bounding_box.min[a];
    bbox_out[3 + a] = obj->bounding_box.max[a];
  }
  info_out[0] = obj->semantic_label;
  info_out[1] = obj->first_observed_ns.empty() ? 0 : static_cast<int64_t>(obj->first_observed_ns[0]);
  info_out[2] = obj->last_observed_ns.empty() ? 0 : static_cast<int64_t>(obj->last_observed_ns[0]);
  return 1;
}

/* khronos::ActiveWindow (active_window.cpp:76-286 + object_worker_pool.cpp): the reference's own module -- constructor, spinOnce,
 * createData, updateMap, extractOutputData, extractInactiveObjects, the worker pool -- with the reference's own sub-modules
 * (FreeSpaceMotionDetector, ConnectedSemantics, MaxIoUTracker, TrackingIntegrator, MeshObjectExtractor, FrameDataBuffer) plugged
 * in through the factories, and the three things that are not in /root/reference (input conversion, projective integrator, mesh
 * integrator) bridged to the CPU oracle. */
struct ref_aw_config {
  float voxel_size;
  int32_t voxels_per_side;
  float truncation_distance;
  float min_output_separation;
  int32_t detach_object_extraction;
  float temporal_buffer, tsdf_occupancy_threshold;
  int32_t neighbor_connectivity;
  float temporal_window;
  int32_t md_neighbor_connectivity, md_min_cluster_size, md_max_cluster_size;
  float md_min_separation_distance, md_max_range, md_min_z_coordinate;
  int32_t od_use_full_connectivity, od_min_cluster_size, od_max_cluster_size, od_use_3d;
  float od_grid_size, od_max_range;
  int32_t tr_assign_track;
  float tr_min_semantic_iou, tr_min_cross_iou, tr_max_dynamic_distance, tr_temporal_window;
  int32_t tr_min_num_observations;
  float tr_voxel_size;
  float ex_min_allocation_confidence, ex_min_volume, ex_max_volume;
  int32_t ex_only_reconstructed;
  float ex_min_dynamic_displacement, ex_min_reconstruction_confidence;
  int32_t ex_min_reconstruction_observations;
  float ex_resolution, ex_min_resolution;
  int32_t buffer_size, num_threads, num_workers;
};

struct RefObject {
  int label;
  uint64_t first_seen, last_seen;
  float bbox[6];
  std::vector<float> points;
};

struct RefActiveWindow {
  BridgeEnv env;
  std::unique_ptr<khronos::ActiveWindow> aw;
  hydra::ActiveWindowOutput::Ptr last_output;
  std::vector<RefObject> objects;
  int width = 0, height = 0;
};

namespace {
void takeObjects(RefActiveWindow* r, hydra::LayerUpdate& update) {
  for (auto& attrs : update.attributes) {
    const auto* o = dynamic_cast<const spark_dsg::KhronosObjectAttributes*>(attrs.get());
    if (!o) continue;
    RefObject ro;
    ro.label = o->semantic_label;
    ro.first_seen = o->first_observed_ns.empty() ? 0 : o->first_observed_ns[0];
    ro.last_seen = o->last_observed_ns.empty() ? 0 : o->last_observed_ns[0];
    for (int a = 0; a < 3; ++a) {
      ro.bbox[a] = o->bounding_box.min[a];
      ro.bbox[3 + a] = o->bounding_box.max[a];
    }
    for (const auto& p : o->mesh.points) {
      ro.points.push_back(p[0]);
      ro.points.push_back(p[1]);
      ro.points.push_back(p[2]);
    }
    r->objects.push_back(std::move(ro));
  }
  update.attributes.clear();
}
}  // namespace

RefActiveWindow* ref_aw_create(const ref_aw_config* c, const orc_config* main_cfg, const orc_config* object_cfg, const orc_sensor* sensor,
                               const int32_t* object_labels, int n_object_labels) {
  installBridge();
  auto* r = new RefActiveWindow();
  r->env.main_cfg = *main_cfg;
  r->env.object_cfg = *object_cfg;
  r->env.sensor = *sensor;
  r->width = sensor->width;
  r->height = sensor->height;
  g_env = &r->env;
  auto& labels = hydra::GlobalInfo::instance().mutableLabelSpaceConfig().object_labels;
  labels.clear();
  labels.insert(object_labels, object_labels + n_object_labels);

  khronos::ActiveWindow::Config cfg;
  cfg.volumetric_map.voxel_size = c->voxel_size;
  cfg.volumetric_map.voxels_per_side = static_cast<size_t>(c->voxels_per_side);
  cfg.volumetric_map.truncation_distance = c->truncation_distance;
  cfg.volumetric_map.with_semantics = true;  // (uHumans2.yaml:49)
  cfg.min_output_separation = c->min_output_separation;
  cfg.detach_object_extraction = c->detach_object_extraction != 0;
  cfg.tracking_integrator.temporal_buffer = c->temporal_buffer;
  cfg.tracking_integrator.tsdf_occupancy_threshold = c->tsdf_occupancy_threshold;
  cfg.tracking_integrator.neighbor_connectivity = c->neighbor_connectivity;
  cfg.tracking_integrator.temporal_window = c->temporal_window;
  cfg.tracking_integrator.num_threads = c->num_threads;
  const ref_aw_config k = *c;
  cfg.motion_detector.factory = [k]() -> std::unique_ptr<khronos::MotionDetector> {
    khronos::FreeSpaceMotionDetector::Config d;
    d.neighbor_connectivity = k.md_neighbor_connectivity;
    d.min_cluster_size = k.md_min_cluster_size;
    d.max_cluster_size = k.md_max_cluster_size;
    d.min_separation_distance = k.md_min_separation_distance;
    d.max_range = k.md_max_range;
    d.min_z_coordinate = k.md_min_z_coordinate;
    d.num_threads = k.num_threads;
    return std::make_unique<khronos::FreeSpaceMotionDetector>(d);
  };
  cfg.object_detector.factory = [k]() -> std::unique_ptr<khronos::ObjectDetector> {
    khronos::ConnectedSemantics::Config d;
    d.use_full_connectivity = k.od_use_full_connectivity != 0;
    d.min_cluster_size = k.od_min_cluster_size;
    d.max_cluster_size = k.od_max_cluster_size;
    d.use_3d = k.od_use_3d != 0;
    d.grid_size = k.od_grid_size;
    d.max_range = k.od_max_range;
    return std::make_unique<khronos::ConnectedSemantics>(d);
  };
  cfg.tracker.factory = [k]() -> std::unique_ptr<khronos::Tracker> {
    khronos::MaxIoUTracker::Config d;
    d.track_by = khronos::MaxIoUTracker::Config::TrackBy::kVoxels;
    d.semantic_association = k.tr_assign_track ? khronos::MaxIoUTracker::Config::SemanticAssociation::kAssignTrack
                                               : khronos::MaxIoUTracker::Config::SemanticAssociation::kAssignCluster;
    d.min_semantic_iou = k.tr_min_semantic_iou;
    d.min_cross_iou = k.tr_min_cross_iou;
    d.max_dynamic_distance = k.tr_max_dynamic_distance;
    d.temporal_window = k.tr_temporal_window;
    d.min_num_observations = k.tr_min_num_observations;
    d.voxel_size = k.tr_voxel_size;
    return std::make_unique<khronos::MaxIoUTracker>(d);
  };
  cfg.object_extractor.factory = [k]() -> std::unique_ptr<khronos::ObjectExtractor> {
    khronos::MeshObjectExtractor::Config d;
    d.min_object_allocation_confidence = k.ex_min_allocation_confidence;
    d.min_object_volume = k.ex_min_volume;
    d.max_object_volume = k.ex_max_volume;
    d.only_extract_reconstructed_objects = k.ex_only_reconstructed != 0;
    d.min_dynamic_displacement = k.ex_min_dynamic_displacement;
    d.min_object_reconstruction_confidence = k.ex_min_reconstruction_confidence;
    d.min_object_reconstruction_observations = k.ex_min_reconstruction_observations;
    d.object_reconstruction_resolution = k.ex_resolution;
    d.min_reconstruction_resolution = k.ex_min_resolution;
    return std::make_unique<khronos::MeshObjectExtractor>(d);
  };
  cfg.extraction_worker.num_workers = static_cast<size_t>(c->num_workers);
  cfg.frame_data_buffer.max_buffer_size = static_cast<size_t>(c->buffer_size);
  r->aw = std::make_unique<khronos::ActiveWindow>(cfg, nullptr);
  return r;
}

namespace {
// the reference's workers are DETACHED threads (object_worker_pool.cpp:131) that use the pool and the bridge environment: nothing
// may go away under them
void drainWorkers(RefActiveWindow* r) {
  auto& pool = r->aw->extraction_worker_;
  while (pool.work_queue_.size() > 0 || pool.curr_workers_ > 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));
  // (a worker decrements the counter BEFORE it appends its object, object_worker_pool.cpp:142-146: give that tail time to finish)
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
}
}  // namespace

void ref_aw_destroy(RefActiveWindow* r) {
  if (!r) return;
  if (r->aw) {
    g_env = &r->env;
    drainWorkers(r);
    r->aw.reset();  // (joins the worker pool's thread while the environment is still there)
  }
  if (g_env == &r->env) g_env = nullptr;
  delete r;
}

/* ActiveWindow::spinOnce on one raw frame; returns 1 when the frame produced an output (active_window.cpp:158-160) */
int ref_aw_spin(RefActiveWindow* r, uint64_t stamp, const double* world_T_sensor, const float* depth, const uint8_t* rgb, const int32_t* label) {
  g_env = &r->env;
  const size_t n = static_cast<size_t>(r->width) * r->height;
  hydra::InputPacket in;
  in.timestamp_ns = stamp;
  in.width = r->width;
  in.height = r->height;
  in.sensor.width = r->width;
  in.sensor.height = r->height;
  in.sensor.fx = r->env.sensor.fx, in.sensor.fy = r->env.sensor.fy, in.sensor.cx = r->env.sensor.cx, in.sensor.cy = r->env.sensor.cy;
  in.depth = std::make_shared<std::vector<float>>(depth, depth + n);
  if (rgb) in.rgb = std::make_shared<std::vector<uint8_t>>(rgb, rgb + 3 * n);
  if (label) in.label = std::make_shared<std::vector<int32_t>>(label, label + n);
  std::memcpy(in.world_T_sensor16, world_T_sensor, sizeof(in.world_T_sensor16));
  r->last_output = r->aw->spinOnce(in);
  if (!r->last_output) return 0;
  for (auto& kv : r->last_output->graph_update) takeObjects(r, *kv.second);
  return 1;
}

/* the latest frame's dynamic and object image */
void ref_aw_frame_images(RefActiveWindow* r, int32_t* dynamic_out, int32_t* object_out) {
  const khronos::FrameData& d = r->aw->getLatestFrameData();
  for (int v = 0; v < r->height; ++v)
    for (int u = 0; u < r->width; ++u) {
      dynamic_out[v * r->width + u] = d.dynamic_image.at<int>(v, u);
      object_out[v * r->width + u] = d.object_image.at<int>(v, u);
    }
}

/* the tracker's tracks as one JSON line (the format of host_selftest --tracker) */
int64_t ref_aw_tracks(RefActiveWindow* r, char* out, int64_t cap) {
  std::string result = "[";
  bool first = true;
  for (const khronos::Track& t : r->aw->getTracks()) {
    const khronos::Observation& o = t.observations.back();
    char buf[512];
    std::snprintf(buf, sizeof(buf),
                  "%s{\"id\": %d, \"dyn\": %d, \"active\": %d, \"conf\": %.9g, \"first\": %llu, \"last\": %llu, \"cat\": %d, "
                  "\"n_obs\": %zu, \"obs\": [%llu, %d, %d], \"n_vox\": %zu}",
                  first ? "" : ", ", t.id, int(t.is_dynamic), int(t.is_active), t.confidence, static_cast<unsigned long long>(t.first_seen),
                  static_cast<unsigned long long>(t.last_seen), t.semantics ? t.semantics->category_id : -1, t.observations.size(),
                  static_cast<unsigned long long>(o.stamp), o.semantic_cluster_id, o.dynamic_cluster_id, t.last_voxels.size());
    result += buf;
    first = false;
  }
  result += "]";
  const int64_t n = std::min<int64_t>(static_cast<int64_t>(result.size()), cap > 0 ? cap - 1 : 0);
  if (cap > 0) {
    std::memcpy(out, result.data(), static_cast<size_t>(n));
    out[n] = 0;
  }
  return static_cast<int64_t>(result.size());
}

/* the window's map: sorted block indices; one block's tracking state (as ref_get_block, block_flags additionally bit0 updated,
 * bit1 mesh_updated) */
int64_t ref_aw_block_indices(RefActiveWindow* r, int32_t* out, int64_t cap) {
  auto v = r->aw->getMap().getTsdfLayer().allocatedBlockIndices();
  std::sort(v.begin(), v.end(), [](const hydra::BlockIndex& a, const hydra::BlockIndex& b) {
    return a[0] != b[0] ? a[0] < b[0] : (a[1] != b[1] ? a[1] < b[1] : a[2] < b[2]);
  });
  for (int64_t i = 0; i < static_cast<int64_t>(v.size()) && i < cap; ++i)
    for (int a = 0; a < 3; ++a) out[3 * i + a] = v[i][a];
  return static_cast<int64_t>(v.size());
}

int ref_aw_get_block(RefActiveWindow* r, const int32_t* idx, float* distance, float* weight, uint64_t* last_observed, uint64_t* last_occupied,
                     uint8_t* flags, uint8_t* block_flags) {
  const hydra::BlockIndex bi(idx[0], idx[1], idx[2]);
  const auto tsdf = r->aw->getMap().getTsdfLayer().getBlockPtr(bi);
  const auto trk = r->aw->getMap().getTrackingLayer()->getBlockPtr(bi);
  if (!tsdf || !trk) return -1;
  for (size_t i = 0; i < trk->numVoxels(); ++i) {
    const hydra::TrackingVoxel& v = trk->getVoxel(i);
    if (distance) distance[i] = tsdf->getVoxel(i).distance;
    if (weight) weight[i] = tsdf->getVoxel(i).weight;
    if (last_observed) last_observed[i] = v.last_observed;
    if (last_occupied) last_occupied[i] = v.last_occupied;
    if (flags) flags[i] = static_cast<uint8_t>((v.active ? 1 : 0) | (v.ever_free ? 2 : 0) | (v.to_remove ? 4 : 0));
  }
  if (block_flags)
    *block_flags = static_cast<uint8_t>((tsdf->updated ? 1 : 0) | (tsdf->mesh_updated ? 2 : 0) | (tsdf->tracking_updated ? 4 : 0) | (trk->has_active_data ? 8 : 0));
  return 0;
}

/* the latest output (extractOutputData, active_window.cpp:217-249): info = {stamp, archived blocks, cloned (updated) blocks, mesh
 * vertices of the window}; the two index lists sorted */
void ref_aw_output(RefActiveWindow* r, int64_t* info, int32_t* archived, int64_t cap_archived, int32_t* cloned, int64_t cap_cloned) {
  const auto& o = r->last_output;
  auto sorted = [](hydra::BlockIndices v) {
    std::sort(v.begin(), v.end(), [](const hydra::BlockIndex& a, const hydra::BlockIndex& b) {
      return a[0] != b[0] ? a[0] < b[0] : (a[1] != b[1] ? a[1] < b[1] : a[2] < b[2]);
    });
    return v;
  };
  const auto arch = sorted(o->archived_mesh_indices);
  const auto cl = sorted(o->map->getTsdfLayer().allocatedBlockIndices());
  info[0] = static_cast<int64_t>(o->timestamp_ns);
  info[1] = static_cast<int64_t>(arch.size());
  info[2] = static_cast<int64_t>(cl.size());
  size_t nvert = 0;
  for (const auto& mb : r->aw->getMap().getMeshLayer()) nvert += mb.points.size();
  info[3] = static_cast<int64_t>(nvert);
  for (int64_t i = 0; i < static_cast<int64_t>(arch.size()) && i < cap_archived; ++i)
    for (int a = 0; a < 3; ++a) archived[3 * i + a] = arch[i][a];
  for (int64_t i = 0; i < static_cast<int64_t>(cl.size()) && i < cap_cloned; ++i)
    for (int a = 0; a < 3; ++a) cloned[3 * i + a] = cl[i][a];
}

/* wait for the detached extractions (object_worker_pool.cpp:115-146) and take what they produced; returns the number of objects so far */
int64_t ref_aw_collect(RefActiveWindow* r) {
  auto& pool = r->aw->extraction_worker_;
  drainWorkers(r);
  hydra::LayerUpdate update(2);
  pool.fill(update);
  takeObjects(r, update);
  return static_cast<int64_t>(r->objects.size());
}

/* ActiveWindow::extractObjects (active_window.cpp:190-201): every remaining track extracted on the calling thread; the objects
 * are appended to the list (returns their number) */
int64_t ref_aw_extract_objects(RefActiveWindow* r) {
  g_env = &r->env;
  const size_t before = r->objects.size();
  hydra::LayerUpdate update(2);
  for (auto& o : r->aw->extractObjects()) update.attributes.emplace_back(std::make_unique<spark_dsg::KhronosObjectAttributes>(*o));
  takeObjects(r, update);
  return static_cast<int64_t>(r->objects.size() - before);
}

/* ActiveWindow::finishMapping (active_window.cpp:176-188): every block and every track is marked inactive and a last output is
 * extracted on the calling thread (its objects stay in the worker pool: ref_aw_collect) */
void ref_aw_finish(RefActiveWindow* r) {
  g_env = &r->env;
  r->aw->finishMapping();
}

/* object i: info = {label, first seen, last seen, vertices}; bbox (min, max); the vertices (box frame) up to cap */
void ref_aw_object(RefActiveWindow* r, int64_t i, int64_t* info, float* bbox, float* points, int64_t cap_points) {
  const RefObject& o = r->objects[static_cast<size_t>(i)];
  info[0] = o.label;
  info[1] = static_cast<int64_t>(o.first_seen);
  info[2] = static_cast<int64_t>(o.last_seen);
  info[3] = static_cast<int64_t>(o.points.size() / 3);
  std::memcpy(bbox, o.bbox, sizeof(o.bbox));
  std::memcpy(points, o.points.data(), sizeof(float) * std::min<size_t>(o.points.size(), 3 * static_cast<size_t>(cap_points)));
}

/* MeshObjectExtractor::extractDynamicObject (mesh_object_extractor.cpp:120-172, through extractObject) over a script in the
 * format of host_selftest --dynobj: a cluster given as a box becomes two pixels whose vertices are the box corners; the same lines. */
int64_t ref_dynobj_replay(const char* script, char* out, int64_t cap) {
  std::istringstream in(script);
  std::string result, tok;
  khronos::MeshObjectExtractor::Config cfg;
  khronos::FrameDataBuffer::Config bc;
  bc.max_buffer_size = 4096;
  khronos::FrameDataBuffer buffer(bc);
  std::unique_ptr<khronos::MeshObjectExtractor> extractor;
  while (in >> tok) {
    if (tok == "X") {
      in >> cfg.min_dynamic_displacement >> cfg.min_object_allocation_confidence;
      extractor = std::make_unique<khronos::MeshObjectExtractor>(cfg);
    } else if (tok == "F") {
      hydra::InputData input;
      size_t n;
      in >> input.timestamp_ns >> n;
      input.vertex_map = cv::Mat(1, static_cast<int>(std::max<size_t>(2 * n, 1)), sizeof(cv::Vec3f));
      auto fd = std::make_shared<khronos::FrameData>(input);
      cv::Mat vm = fd->input.vertex_map;  // (shares the pixels)
      for (size_t k = 0; k < n; ++k) {
        khronos::MeasurementCluster c;
        in >> c.id;
        for (int corner = 0; corner < 2; ++corner) {
          cv::Vec3f& v = vm.at<cv::Vec3f>(0, static_cast<int>(2 * k + corner));
          in >> v[0] >> v[1] >> v[2];
          c.pixels.emplace_back(static_cast<int>(2 * k + corner), 0);
        }
        fd->dynamic_clusters.push_back(std::move(c));
      }
      buffer.storeData(fd);
    } else if (tok == "K") {
      khronos::Track t;
      t.id = 0;
      t.is_dynamic = true;
      size_t n;
      in >> t.confidence >> t.first_seen >> t.last_seen >> n;
      for (size_t i = 0; i < n; ++i) {
        uint64_t st;
        int id;
        in >> st >> id;
        t.observations.emplace_back(st, -1, id);
      }
      const auto obj = extractor->extractObject(t, buffer);
      if (!obj) {
        result += "null\n";
      } else {
        char buf[512];
        std::snprintf(buf, sizeof(buf), "%zu %llu %llu %.9g %.9g %.9g %.9g %.9g %.9g\n", obj->trajectory_positions.size(),
                      static_cast<unsigned long long>(obj->first_observed_ns[0]), static_cast<unsigned long long>(obj->last_observed_ns[0]),
                      obj->bounding_box.min[0], obj->bounding_box.min[1], obj->bounding_box.min[2], obj->bounding_box.max[0], obj->bounding_box.max[1],
                      obj->bounding_box.max[2]);
        result += buf;
      }
    }
  }
  const int64_t n = std::min<int64_t>(static_cast<int64_t>(result.size()), cap > 0 ? cap - 1 : 0);
  if (cap > 0) {
    std::memcpy(out, result.data(), static_cast<size_t>(n));
    out[n] = 0;
  }
  return static_cast<int64_t>(result.size());
}

/* InstanceForwarding::processInput (instance_forwarding.cpp:73-149): instance ids of the label image forwarded as clusters.
 * features: n_features rows of (id, dim floats) -- empty = closed set; background: n_background prompt embeddings of dim floats
 * (empty = no background filter).  object_out: H*W; per cluster (up to cap): {id, category or -1, has feature}, pixel count.
 * returns the number of clusters */
int ref_forward_instances(int W, int H, const float* range, const float* vertex, const int32_t* label, float max_range, int min_cluster_size,
                          int max_cluster_size, double min_object_volume, double max_object_volume, double max_background_score, int dim,
                          const int32_t* feature_ids, const float* features, int n_features, const float* background, int n_background,
                          int32_t* object_out, int32_t* cluster_info_out, int64_t* n_pixels_out, int cap) {
  khronos::InstanceForwarding::Config c;
  c.max_range = max_range;
  c.min_cluster_size = min_cluster_size;
  c.max_cluster_size = max_cluster_size;
  c.min_object_volume = min_object_volume;
  c.max_object_volume = max_object_volume;
  c.max_background_score = max_background_score;
  auto toFeature = [dim](const float* p) {
    hydra::FeatureVector f(static_cast<size_t>(dim), 1);
    for (int i = 0; i < dim; ++i) f(static_cast<size_t>(i)) = p[i];
    return f;
  };
  if (n_background > 0) {
    std::vector<hydra::FeatureVector> prompts;
    for (int k = 0; k < n_background; ++k) prompts.push_back(toFeature(background + static_cast<size_t>(k) * dim));
    c.background.factory = [prompts]() {
      auto g = std::make_unique<hydra::EmbeddingGroup>();
      g->embeddings = prompts;
      return g;
    };
  }
  khronos::InstanceForwarding detector(c);
  hydra::InputData in;
  in.range_image = cv::Mat(H, W, sizeof(float));
  in.vertex_map = cv::Mat(H, W, sizeof(cv::Vec3f));
  in.label_image = cv::Mat(H, W, sizeof(int));
  std::memcpy(in.range_image.data(), range, sizeof(float) * static_cast<size_t>(W) * H);
  std::memcpy(in.vertex_map.data(), vertex, sizeof(float) * 3 * static_cast<size_t>(W) * H);
  std::memcpy(in.label_image.data(), label, sizeof(int32_t) * static_cast<size_t>(W) * H);
  for (int k = 0; k < n_features; ++k) in.label_features[feature_ids[k]] = toFeature(features + static_cast<size_t>(k) * dim);
  khronos::FrameData data(in);
  data.dynamic_image = cv::Mat(H, W, sizeof(int));
  data.object_image = cv::Mat(H, W, sizeof(int));
  hydra::VolumetricMap::Config mc;
  const hydra::VolumetricMap map(mc);
  detector.processInput(map, data);
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) object_out[v * W + u] = data.object_image.at<int>(v, u);
  int k = 0;
  for (const auto& cl : data.semantic_clusters) {
    if (k < cap) {
      cluster_info_out[3 * k] = cl.id;
      cluster_info_out[3 * k + 1] = cl.semantics ? cl.semantics->category_id : -1;
      cluster_info_out[3 * k + 2] = cl.semantics && cl.semantics->feature.size() > 1 ? 1 : 0;
      n_pixels_out[k] = static_cast<int64_t>(cl.pixels.size());
    }
    ++k;
  }
  return k;
}

/* the YAML keys the reference's modules declare: every declare_config() of the compiled files is RUN with a recording
 * config::field; one line per module, "Module: key key ..." */
int64_t ref_config_keys(char* out, int64_t cap) {
  std::string result;
  auto dump = [&](const char* module, auto cfg) {  // (cfg: default-constructed, so the values are the reference's defaults)
    config::recordedKeys().clear();
    config::recordedDefaults().clear();
    config::recordedChecks().clear();
    declare_config(cfg);
    result += module;
    result += ":";
    for (const auto& k : config::recordedKeys()) result += " " + k;
    result += " |";
    for (const auto& k : config::recordedDefaults()) result += " " + k;
    result += " |";
    for (const auto& k : config::recordedChecks()) result += " " + k + ";";
    result += "\n";
  };
  dump("ActiveWindow", khronos::ActiveWindow::Config());
  dump("TrackingIntegrator", khronos::TrackingIntegrator::Config());
  dump("FreeSpaceMotionDetector", khronos::FreeSpaceMotionDetector::Config());
  dump("ConnectedSemantics", khronos::ConnectedSemantics::Config());
  dump("InstanceForwarding", khronos::InstanceForwarding::Config());
  dump("MaxIoUTracker", khronos::MaxIoUTracker::Config());
  dump("ExternalTracker", khronos::ExternalTracker::Config());
  dump("MeshObjectExtractor", khronos::MeshObjectExtractor::Config());
  dump("ObjectWorkerPool", khronos::ObjectWorkerPool::Config());
  dump("FrameDataBuffer", khronos::FrameDataBuffer::Config());
  dump("RayVerificator", khronos::RayVerificator::Config());
  dump("RayChangeDetector", khronos::RayChangeDetector::Config());
  dump("RayBackgroundChangeDetector", khronos::RayBackgroundChangeDetector::Config());
  dump("RayObjectChangeDetector", khronos::RayObjectChangeDetector::Config());
  const int64_t n = std::min<int64_t>(static_cast<int64_t>(result.size()), cap > 0 ? cap - 1 : 0);
  if (cap > 0) {
    std::memcpy(out, result.data(), static_cast<size_t>(n));
    out[n] = 0;
  }
  return static_cast<int64_t>(result.size());
}

/* utils::combineMeshLayer (geometry_utils.cpp:61-86): blocks given as vertex counts + faces per block (local indices);
 * returns the combined faces (global indices) and the combined order of a per-vertex tag */
int64_t ref_combine_mesh(int n_blocks, const int64_t* n_vertices, const int64_t* n_faces, const float* points, const uint32_t* labels,
                         const int64_t* faces, float* points_out, uint32_t* labels_out, int64_t* faces_out) {
  hydra::MeshLayer layer;
  size_t vo = 0, fo = 0;
  for (int b = 0; b < n_blocks; ++b) {
    hydra::MeshBlock mb;
    for (int64_t i = 0; i < n_vertices[b]; ++i, ++vo) {
      mb.points.emplace_back(points[3 * vo], points[3 * vo + 1], points[3 * vo + 2]);
      mb.labels.push_back(labels[vo]);
      mb.colors.emplace_back();
      mb.stamps.push_back(vo);
      mb.first_seen_stamps.push_back(vo);
    }
    for (int64_t i = 0; i < n_faces[b]; ++i, ++fo)
      mb.faces.push_back({static_cast<size_t>(faces[3 * fo]), static_cast<size_t>(faces[3 * fo + 1]), static_cast<size_t>(faces[3 * fo + 2])});
    layer.push_back(std::move(mb));
  }
  const hydra::Mesh out = khronos::utils::combineMeshLayer(layer);
  for (size_t i = 0; i < out.points.size(); ++i) {
    for (int a = 0; a < 3; ++a) points_out[3 * i + a] = out.points[i][a];
    labels_out[i] = out.labels[i];
  }
  for (size_t i = 0; i < out.faces.size(); ++i)
    for (int a = 0; a < 3; ++a) faces_out[3 * i + a] = static_cast<int64_t>(out.faces[i][a]);
  return static_cast<int64_t>(out.faces.size());
}

}  // extern "C"

// how often the reference's own ObjectIntegrator::computeLabel (object_integrator.cpp:58-81) has run under the bridge / returned false
extern "C" void ref_label_hook_stats(uint64_t* calls, uint64_t* skips) {
  *calls = g_label_hook_calls.load();
  *skips = g_label_hook_skips.load();
}
