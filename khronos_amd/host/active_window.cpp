// active_window.cpp — host orchestration of the fusion path, mirroring khronos::ActiveWindow
// (khronos/src/active_window/active_window.cpp:73-286) on top of the C ABI.  See active_window.h.
#include "active_window.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>

namespace khronos {

namespace {

void chk(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + " failed: " + khr_last_error());
}

int interpolationFromName(const std::string& n) {
  if (n == "nearest") return 0;
  if (n == "bilinear") return 1;
  if (n == "adaptive") return 2;
  throw std::invalid_argument("interpolation_method must be one of {nearest, bilinear, adaptive}");
}

// a two-valued [A] switch by name: 0 for `zero`, 1 for `one`
int switchFromName(const std::string& v, const char* zero, const char* one, const char* key) {
  if (v == zero) return 0;
  if (v == one) return 1;
  throw std::invalid_argument(std::string(key) + " must be one of {" + zero + ", " + one + "}");
}

}  // namespace

// ---- FrameData ------------------------------------------------------------------------------------------
std::vector<int32_t> FrameData::dynamicImage() const {
  std::vector<int32_t> d(static_cast<size_t>(input.sensor.width) * input.sensor.height);
  chk(khr_download_frame(input.ctx, input.slot, nullptr, nullptr, d.data()), "khr_download_frame");
  return d;
}

// ---- FrameDataBuffer (frame_data_buffer.cpp:57-123) ------------------------------------------------------
FrameDataBuffer::FrameDataBuffer(const Config& cfg) : config(cfg) {
  if (config.max_buffer_size == 0) throw std::invalid_argument("max_buffer_size must be > 0");  // :52
}

void FrameDataBuffer::trimBuffer(const Tracks& tracks) {
  // drop every buffered frame that no track observation refers to any more (:57-86)
  auto referenced = [&tracks](TimeStamp stamp) {
    for (const Track& t : tracks)
      for (const Observation& o : t.observations)
        if (o.stamp == stamp) return true;
    return false;
  };
  for (auto it = buffer_.begin(); it != buffer_.end();) it = referenced((*it)->input.timestamp_ns) ? it + 1 : buffer_.erase(it);
  if (!buffer_.empty()) oldest_time_stamp_ = buffer_.front()->input.timestamp_ns;
}

void FrameDataBuffer::storeData(const FrameData::Ptr& data) {
  // every n-th frame is appended, the others overwrite the newest entry (:88-109)
  const bool append = input_counter_ == 0;
  input_counter_ = (input_counter_ + 1) % std::max(1, config.store_every_n_frames);
  if (!append) {
    if (!buffer_.empty()) buffer_.pop_back();
    buffer_.push_back(data);
    return;
  }
  buffer_.push_back(data);
  if (buffer_.size() > config.max_buffer_size) {
    buffer_.pop_front();
    oldest_time_stamp_ = buffer_.front()->input.timestamp_ns;
  }
}

FrameData::Ptr FrameDataBuffer::getData(TimeStamp stamp) const {
  if (stamp < oldest_time_stamp_) return nullptr;  // :111-123
  for (const auto& f : buffer_)
    if (f->input.timestamp_ns == stamp) return f;
  return nullptr;
}

// ---- FreeSpaceMotionDetector ---------------------------------------------------------------------------------
FreeSpaceMotionDetector::FreeSpaceMotionDetector(const Config& cfg) : config(cfg) {
  // checks of free_space_motion_detector.cpp:63-67
  if (config.num_threads == 0) throw std::invalid_argument("num_threads must be > 0 (or -1)");
  if (config.neighbor_connectivity != 6 && config.neighbor_connectivity != 18 && config.neighbor_connectivity != 26)
    throw std::invalid_argument("neighbor_connectivity must be one of {6, 18, 26}");
  if (config.max_cluster_size < config.min_cluster_size)
    throw std::invalid_argument("param 'max_cluster_size' must be >= 'min_cluster_size'");
  if (!(config.max_range > 0.f)) throw std::invalid_argument("max_range must be > 0");
}

void FreeSpaceMotionDetector::processInput(const VolumetricMap& map, FrameData& data) {
  // free_space_motion_detector.cpp:73-103; the parameters were handed to the device context at creation
  const int n = khr_detect_motion(map.ctx(), data.input.slot);
  chk(n, "khr_detect_motion");
  data.num_dynamic_clusters = n;
  fetchClusters(map, data);
}

void FreeSpaceMotionDetector::fetchClusters(const VolumetricMap& map, FrameData& data) {
  data.dynamic_clusters.clear();
  if (data.num_dynamic_clusters <= 0) return;
  // ids saturate at 255 but the cluster list does not (free_space_motion_detector.cpp:384-395)
  std::vector<khr_cluster> cl(static_cast<size_t>(std::max(data.num_dynamic_clusters, 1)));
  const int n = khr_get_dynamic_clusters(map.ctx(), data.input.slot, cl.data(), static_cast<int>(cl.size()));
  chk(n, "khr_get_dynamic_clusters");
  for (int i = 0; i < n && i < static_cast<int>(cl.size()); ++i) {
    MeasurementCluster m;
    m.id = cl[i].id;
    m.num_pixels = cl[i].num_pixels_listed;
    if (cl[i].num_pixels_painted) {
      m.bounding_box.include(cl[i].bbox_min);
      m.bounding_box.include(cl[i].bbox_max);
    }
    for (int d = 0; d < 3; ++d) m.centroid[d] = cl[i].centroid[d];
    data.dynamic_clusters.push_back(m);
  }
}

// ---- MeshObjectExtractor ----------------------------------------------------------------------------------------
MeshObjectExtractor::MeshObjectExtractor(const Config& cfg, const khr_config& aw_device_config)
    : config(cfg), device_config_(aw_device_config) {
  // checks of mesh_object_extractor.cpp:66-75
  auto in01 = [](float v) { return v >= 0.f && v <= 1.f; };
  if (!in01(config.min_object_allocation_confidence) || !in01(config.min_object_reconstruction_confidence))
    throw std::invalid_argument("confidences must be in [0, 1]");
  if (config.min_object_volume < 0 || config.max_object_volume < config.min_object_volume)
    throw std::invalid_argument("object volume limits are inconsistent");
  if (config.min_dynamic_displacement < 0 || config.min_reconstruction_resolution < 0)
    throw std::invalid_argument("negative displacement / resolution");
  // The private object map is created HERE, once (hipMalloc / hipFree synchronise the whole device: creating it at the
  // first extraction stalled the active window for ~6 ms in the middle of the stream); extractions re-scale and empty it
  // (khr_reset_map).  It only grows when an object needs more blocks than it holds.
  if (aw_device_config.max_frame_pixels > 0) {
    khr_config oc = objectMapConfig(0.05f);
    // large enough for almost every object at the default resolution (2 % of the largest extent: 50 voxels = 7 blocks per box
    // side, the allocated range is twice the box: <= 15^3 blocks); growing it later means destroying and re-creating an HBM pool
    // in the middle of the stream (~5 ms during which an extraction's join waits: profiles/r05_host_input_marks.txt).  15 KB a block.
    oc.max_blocks = std::min<uint32_t>(8192u, std::max<uint32_t>(config.max_object_blocks, 1u));
    oc.max_mesh_vertices = std::max<uint64_t>(1u << 16, static_cast<uint64_t>(oc.max_blocks) * 512ull * 15ull / 4);
    if (khr_create(&oc, &object_ctx_) == KHR_OK) object_ctx_blocks_ = oc.max_blocks;
    else object_ctx_ = nullptr;  // (no device yet: created on first use)
  }
}

khr_config MeshObjectExtractor::objectMapConfig(float voxel_size) const {
  // private map (mesh_object_extractor.cpp:201-215): vps 8, truncation 2 voxels, binary semantics, no tracking
  khr_config oc = device_config_;
  oc.voxel_size = voxel_size;
  oc.voxels_per_side = 8;
  oc.truncation_distance = oc.voxel_size * 2;
  oc.with_semantics = 1;
  oc.with_tracking = 0;
  oc.num_labels = 2;
  oc.semantic_mode = 1;
  // the extractor's own integrators (mesh_object_extractor.cpp:63-64,239,267), not the window's
  const auto& q = config.projective_integrator;
  oc.use_weight_dropoff = q.use_weight_dropoff;
  oc.weight_dropoff_epsilon = q.weight_dropoff_epsilon;
  oc.use_constant_weight = q.use_constant_weight;
  oc.max_weight = q.max_weight;
  oc.interpolation_method = interpolationFromName(q.interpolation_method);
  oc.color_blend_weight = switchFromName(q.color_blend_weight, "post", "pre", "object_extractor.projective_integrator.color_blend_weight");
  oc.mesh_min_weight = config.mesh_integrator.min_weight;
  oc.mesh_attr_source = switchFromName(config.mesh_integrator.attr_source, "nearest", "containing", "object_extractor.mesh_integrator.attr_source");
  oc.mesh_degenerate_eps = config.mesh_integrator.degenerate_eps;
  oc.alloc_candidate = 0;  // (object maps are allocated explicitly, allocate = false at :242)
  oc.num_frame_slots = 1;
  oc.max_frame_pixels = 4;
  oc.relaxed_arithmetic = device_config_.relaxed_arithmetic;
  oc.rank = 0;
  oc.world_size = 1;
  return oc;
}

MeshObjectExtractor::~MeshObjectExtractor() {
  if (object_ctx_) khr_destroy(object_ctx_);
}

float MeshObjectExtractor::objectVoxelSize(const Config& config, const BoundingBox& extent) {
  // mesh_object_extractor.cpp:201-207
  if (config.object_reconstruction_resolution < 0.f)
    return std::max(extent.maxDimension() * -config.object_reconstruction_resolution, config.min_reconstruction_resolution);
  return config.object_reconstruction_resolution;
}

void MeshObjectExtractor::objectBlockRange(const BoundingBox& extent, float block_size, int32_t* mn, int32_t* mx) {
  // centre -/+ the full dimensions (mesh_object_extractor.cpp:220-221), block index = floor(p / block_size)
  const float inv = 1.f / block_size;
  for (int i = 0; i < 3; ++i) {
    mn[i] = static_cast<int32_t>(std::floor((extent.center(i) - extent.dimension(i)) * inv));
    mx[i] = static_cast<int32_t>(std::floor((extent.center(i) + extent.dimension(i)) * inv));
  }
}

std::shared_ptr<KhronosObjectAttributes> MeshObjectExtractor::extractObject(const Track& track, const FrameDataBuffer& frames) {
  // trackIsValid (mesh_object_extractor.cpp:358-...): confidence gate
  if (track.confidence <= config.min_object_allocation_confidence) return nullptr;
  auto object = track.is_dynamic ? extractDynamicObject(track, frames) : extractStaticObject(track, frames);
  if (!object) return nullptr;
  if (track.semantics) {  // mesh_object_extractor.cpp:96-99
    object->semantic_label = track.semantics->category_id;
    object->semantic_feature = track.semantics->feature;
  }
  object->first_observed_ns = {track.first_seen};
  object->last_observed_ns = {track.last_seen};
  for (int i = 0; i < 3; ++i) object->position[i] = object->bounding_box.center(i);
  return object;
}

std::shared_ptr<KhronosObjectAttributes> MeshObjectExtractor::extractDynamicObject(const Track& track,
                                                                                  const FrameDataBuffer& buffer) const {
  // mesh_object_extractor.cpp:120-172: per observation the cluster centroid -> trajectory, mean bounding-box extent,
  // maximal displacement from the first position
  auto object = std::make_shared<KhronosObjectAttributes>();
  float extent[3] = {0, 0, 0};
  float max_displacement = 0.f;
  for (const Observation& o : track.observations) {
    if (o.dynamic_cluster_id == -1) continue;
    const FrameData::Ptr f = buffer.getData(o.stamp);
    if (!f) continue;
    const MeasurementCluster* cl = nullptr;
    for (const auto& c : f->dynamic_clusters)
      if (c.id == o.dynamic_cluster_id) cl = &c;
    if (!cl) continue;
    object->trajectory_positions.push_back({cl->centroid[0], cl->centroid[1], cl->centroid[2]});
    object->trajectory_timestamps.push_back(o.stamp);
    float d2 = 0.f;
    for (int i = 0; i < 3; ++i) {
      extent[i] += cl->bounding_box.dimension(i);
      const float d = cl->centroid[i] - object->trajectory_positions.front()[i];
      d2 += d * d;
    }
    max_displacement = std::max(max_displacement, std::sqrt(d2));
  }
  if (object->trajectory_positions.empty()) return nullptr;                 // :156-160
  if (max_displacement < config.min_dynamic_displacement) return nullptr;   // :161-166
  const float n = static_cast<float>(object->trajectory_positions.size());
  const auto& c0 = object->trajectory_positions.front();
  BoundingBox bb;  // BoundingBox(mean extent, first position) (:167-168)
  const float lo[3] = {c0[0] - 0.5f * extent[0] / n, c0[1] - 0.5f * extent[1] / n, c0[2] - 0.5f * extent[2] / n};
  const float hi[3] = {c0[0] + 0.5f * extent[0] / n, c0[1] + 0.5f * extent[1] / n, c0[2] + 0.5f * extent[2] / n};
  bb.include(lo);
  bb.include(hi);
  object->bounding_box = bb;
  return object;
}

std::shared_ptr<KhronosObjectAttributes> MeshObjectExtractor::extractStaticObject(const Track& track,
                                                                                 const FrameDataBuffer& buffer) const {
  if (config.object_reconstruction_resolution == 0.f) return nullptr;
  // collectSemanticFrames + computeExtent (mesh_object_extractor.cpp:306-340)
  std::vector<std::pair<FrameData::Ptr, int>> frames;
  BoundingBox extent;
  for (const Observation& o : track.observations) {
    if (o.semantic_cluster_id == -1) continue;
    FrameData::Ptr f = buffer.getData(o.stamp);
    if (!f) continue;
    frames.emplace_back(f, o.semantic_cluster_id);
    for (const auto& cl : f->semantic_clusters)
      if (cl.id == o.semantic_cluster_id) {
        extent.merge(cl.bounding_box);
        break;
      }
  }
  if (frames.empty()) return nullptr;
  if (extent.volume() < config.min_object_volume) return nullptr;

  const float ovs = objectVoxelSize(config, extent);
  if (!(ovs > 0.f)) return nullptr;  // config::isValid(map_config)
  khr_config oc = objectMapConfig(ovs);
  int32_t mn[3], mx[3];
  objectBlockRange(extent, oc.voxel_size * 8.f, mn, mx);
  std::vector<int32_t> idx;
  for (int x = mn[0]; x <= mx[0]; ++x)
    for (int y = mn[1]; y <= mx[1]; ++y)
      for (int z = mn[2]; z <= mx[2]; ++z) {
        idx.push_back(x);
        idx.push_back(y);
        idx.push_back(z);
      }
  const size_t n_blocks = idx.size() / 3;
  if (n_blocks > config.max_object_blocks) return nullptr;
  khr_host_trace("x_begin");
  if (!object_ctx_ || n_blocks + 1 > object_ctx_blocks_) {
    if (object_ctx_) khr_destroy(object_ctx_);
    object_ctx_ = nullptr;
    oc.max_blocks = static_cast<uint32_t>(std::max<size_t>({n_blocks + 1, 2 * static_cast<size_t>(object_ctx_blocks_), 4096}));
    oc.max_mesh_vertices = std::max<uint64_t>(1u << 16, static_cast<uint64_t>(oc.max_blocks) * 512ull * 15ull / 4);
    chk(khr_create(&oc, &object_ctx_), "khr_create(object map)");
    object_ctx_blocks_ = oc.max_blocks;
  } else {
    chk(khr_reset_map(object_ctx_, oc.voxel_size, oc.truncation_distance), "khr_reset_map(object map)");
  }
  khr_ctx* octx = object_ctx_;
  khr_host_trace("x_ctx_ready");
  // the buffered frames were written on the active window's stream: one device-side dependency instead of a host wait per
  // re-integrated frame (the extraction may run on a worker thread while the window keeps queueing frames)
  chk(khr_depend_on(octx, frames.front().first->input.ctx), "khr_depend_on");
  khr_host_trace("x_depends");
  std::shared_ptr<KhronosObjectAttributes> object;
  try {
    chk(khr_allocate_blocks(octx, idx.data(), static_cast<int64_t>(n_blocks)), "khr_allocate_blocks");  // :218-228
    khr_host_trace("x_allocated");
    // projective re-integration of every buffered frame with the binary object label (:239-243): one call, the block
    // list and the per-call bookkeeping are set up once for all frames
    {
      std::vector<int> slots, ids;
      for (const auto& fr : frames) {
        slots.push_back(fr.first->input.slot);
        ids.push_back(fr.second);
      }
      chk(khr_integrate_shared_batch(octx, frames.front().first->input.ctx, slots.data(), ids.data(), static_cast<int>(slots.size()),
                                     /*allocate=*/0, /*use_mask=*/0),
          "khr_integrate_shared_batch");
    }
    khr_host_trace("x_integrated");
    // erase low-confidence voxels (:246-264)
    if (!config.visualize_classification)  // (no count requested: the call stays asynchronous)
      chk(khr_object_prune(octx, config.min_object_reconstruction_confidence, config.min_object_reconstruction_observations, nullptr),
          "khr_object_prune");
    chk(khr_generate_mesh(octx, 1, 0), "khr_generate_mesh");  // :267
    khr_host_trace("x_mesh_queued");
    object = std::make_shared<KhronosObjectAttributes>();
    khr_mesh_view view{};
    const int64_t nv = khr_fetch_mesh(octx, &view);  // one device -> host round trip for the whole object mesh
    chk(static_cast<int>(nv < 0 ? nv : 0), "khr_fetch_mesh");
    hydra::Mesh& mesh = object->mesh;
    mesh.points.resize(3 * nv);
    mesh.colors.resize(4 * nv);
    mesh.labels.resize(nv);
    mesh.first_seen_stamps.resize(nv);
    mesh.stamps.resize(nv);
    if (nv > 0)
      chk(khr_fetch_mesh_into(octx, mesh.points.data(), mesh.colors.data(), mesh.labels.data(), mesh.first_seen_stamps.data(),
                              mesh.stamps.data()),
          "khr_fetch_mesh_into");
    khr_host_trace("x_downloaded");
  } catch (...) {
    khr_destroy(object_ctx_);  // unknown state: start from a fresh context next time
    object_ctx_ = nullptr;
    object_ctx_blocks_ = 0;
    throw;
  }

  if (object->mesh.numVertices() == 0 && config.only_extract_reconstructed_objects) return nullptr;  // :271-275
  if (object->mesh.numVertices() == 0) {
    object->bounding_box = extent;
  } else {
    BoundingBox bb;
    for (size_t i = 0; i < object->mesh.numVertices(); ++i) bb.include(&object->mesh.points[3 * i]);
    object->bounding_box = bb;
  }
  const float vol = object->bounding_box.volume();
  if (vol > config.max_object_volume || vol < config.min_object_volume) return nullptr;  // :283-292
  // move the mesh to the bounding-box frame (:299-302)
  for (size_t i = 0; i < object->mesh.numVertices(); ++i)
    for (int d = 0; d < 3; ++d) object->mesh.points[3 * i + d] -= object->bounding_box.center(d);
  return object;
}

// ---- ActiveWindow::Config --------------------------------------------------------------------------------------
ActiveWindow::Config ActiveWindow::Config::fromYaml(const khronos_amd::YamlNode& n) {
  Config c;
  n.read("verbosity", c.verbosity);
  n.read("min_output_separation", c.min_output_separation);
  n.read("detach_object_extraction", c.detach_object_extraction);
  if (const auto* m = n.find("volumetric_map")) {
    m->read("voxel_size", c.volumetric_map.voxel_size);
    m->read("truncation_distance", c.volumetric_map.truncation_distance);
    m->read("voxels_per_side", c.volumetric_map.voxels_per_side);
    m->read("with_semantics", c.volumetric_map.with_semantics);
    m->read("with_tracking", c.volumetric_map.with_tracking);
  }
  if (const auto* m = n.find("projective_integrator")) {
    m->read("verbosity", c.projective_integrator.verbosity);
    m->read("use_weight_dropoff", c.projective_integrator.use_weight_dropoff);
    m->read("weight_dropoff_epsilon", c.projective_integrator.weight_dropoff_epsilon);
    m->read("use_constant_weight", c.projective_integrator.use_constant_weight);
    m->read("max_weight", c.projective_integrator.max_weight);
    m->read("interpolation_method", c.projective_integrator.interpolation_method);
    m->read("num_threads", c.projective_integrator.num_threads);
    m->read("alloc_candidate", c.projective_integrator.alloc_candidate);
    m->read("color_blend_weight", c.projective_integrator.color_blend_weight);
    if (const auto* si = m->find("semantic_integrator")) si->read("label_confidence", c.projective_integrator.label_confidence);
  }
  if (const auto* m = n.find("tracking_integrator")) {
    auto& t = c.tracking_integrator;
    m->read("verbosity", t.verbosity);
    m->read("temporal_buffer", t.temporal_buffer);
    m->read("burn_in_period", t.burn_in_period);
    m->read("tsdf_occupancy_threshold", t.tsdf_occupancy_threshold);
    m->read("neighbor_connectivity", t.neighbor_connectivity);
    m->read("temporal_window", t.temporal_window);
    m->read("num_threads", t.num_threads);
  }
  if (const auto* m = n.find("motion_detector")) {
    m->read("type", c.motion_detector_type);
    auto& d = c.motion_detector;
    m->read("verbosity", d.verbosity);
    m->read("neighbor_connectivity", d.neighbor_connectivity);
    m->read("min_cluster_size", d.min_cluster_size);
    m->read("max_cluster_size", d.max_cluster_size);
    m->read("min_separation_distance", d.min_separation_distance);
    m->read("max_range", d.max_range);
    m->read("min_z_coordinate", d.min_z_coordinate);
    m->read("num_threads", d.num_threads);
  }
  if (const auto* m = n.find("object_detector")) {
    m->read("type", c.object_detector_type);
    c.object_detector = ConnectedSemantics::Config::fromYaml(*m);
    c.instance_forwarding = InstanceForwarding::Config::fromYaml(*m);
  }
  if (const auto* m = n.find("tracker")) {
    m->read("type", c.tracker_type);
    c.tracker = MaxIoUTracker::Config::fromYaml(*m);
  }
  if (const auto* m = n.find("object_extractor")) {
    m->read("type", c.object_extractor_type);
    auto& e = c.object_extractor;
    m->read("verbosity", e.verbosity);
    m->read("min_object_allocation_confidence", e.min_object_allocation_confidence);
    m->read("min_object_volume", e.min_object_volume);
    m->read("max_object_volume", e.max_object_volume);
    m->read("only_extract_reconstructed_objects", e.only_extract_reconstructed_objects);
    m->read("min_dynamic_displacement", e.min_dynamic_displacement);
    m->read("min_object_reconstruction_confidence", e.min_object_reconstruction_confidence);
    m->read("min_object_reconstruction_observations", e.min_object_reconstruction_observations);
    m->read("object_reconstruction_resolution", e.object_reconstruction_resolution);
    m->read("min_reconstruction_resolution", e.min_reconstruction_resolution);
    m->read("visualize_classification", e.visualize_classification);
    if (const auto* pi = m->find("projective_integrator")) {  // mesh_object_extractor.cpp:63
      auto& q = e.projective_integrator;
      pi->read("verbosity", q.verbosity);
      pi->read("use_weight_dropoff", q.use_weight_dropoff);
      pi->read("weight_dropoff_epsilon", q.weight_dropoff_epsilon);
      pi->read("use_constant_weight", q.use_constant_weight);
      pi->read("max_weight", q.max_weight);
      pi->read("interpolation_method", q.interpolation_method);
      pi->read("num_threads", q.num_threads);
      pi->read("color_blend_weight", q.color_blend_weight);
    }
    if (const auto* mi = m->find("mesh_integrator")) {  // mesh_object_extractor.cpp:64
      mi->read("min_weight", e.mesh_integrator.min_weight);
      mi->read("attr_source", e.mesh_integrator.attr_source);
      mi->read("degenerate_eps", e.mesh_integrator.degenerate_eps);
    }
  }
  if (const auto* m = n.find("extraction_worker")) {
    m->read("num_workers", c.extraction_worker.num_workers);
    m->read("poll_time_us", c.extraction_worker.poll_time_us);
    m->read("verbosity", c.extraction_worker.verbosity);
  }
  if (const auto* m = n.find("mesh_integrator")) {
    m->read("min_weight", c.mesh_integrator.min_weight);
    m->read("attr_source", c.mesh_integrator.attr_source);
    m->read("degenerate_eps", c.mesh_integrator.degenerate_eps);
  }
  if (const auto* m = n.find("frame_data_buffer")) {
    auto& b = c.frame_data_buffer;
    m->read("max_buffer_size", b.max_buffer_size);
    m->read("store_every_n_frames", b.store_every_n_frames);
  }
  if (const auto* m = n.find("khronos_sinks"))  // active_window.cpp:70 ("[]" or absent: none)
    for (const khronos_amd::YamlNode* item : m->items()) c.khronos_sinks.push_back(*item);
  if (const auto* m = n.find("device")) {  // extension block (not in the reference): HBM sizing / placement
    m->read("num_labels", c.num_labels);
    int v = static_cast<int>(c.max_blocks);
    m->read("max_blocks", v);
    c.max_blocks = static_cast<uint32_t>(v);
    m->read("frame_slot_headroom", c.frame_slot_headroom);
    v = static_cast<int>(c.max_snapshot_blocks);
    m->read("max_snapshot_blocks", v);
    c.max_snapshot_blocks = static_cast<uint32_t>(v);
    v = static_cast<int>(c.max_frame_pixels);
    m->read("max_frame_pixels", v);
    c.max_frame_pixels = static_cast<uint32_t>(v);
    m->read("device", c.device);
    m->read("rank", c.rank);
    m->read("world_size", c.world_size);
    m->read("exact_arithmetic", c.exact_arithmetic);
    m->read("timing_sync_device", c.timing_sync_device);
    m->read("fuse_device_stages", c.fuse_device_stages);
  }
  return c;
}

ActiveWindow::Config ActiveWindow::Config::fromYamlString(const std::string& text) {
  const khronos_amd::YamlNode root = khronos_amd::parseYaml(text);
  const khronos_amd::YamlNode* aw = root.find("active_window");
  Config c = fromYaml(aw ? *aw : root);
  if (aw) {
    std::string type;
    aw->read("type", type);
    if (!type.empty() && type != "ActiveWindow") throw std::invalid_argument("active_window.type must be 'ActiveWindow'");
  }
  return c;
}

void ActiveWindow::Config::checkValid() const {
  const auto& t = tracking_integrator;  // tracking_integrator.cpp:61-65
  if (t.neighbor_connectivity != 6 && t.neighbor_connectivity != 18 && t.neighbor_connectivity != 26)
    throw std::invalid_argument("tracking_integrator.neighbor_connectivity must be one of {6, 18, 26}");
  if (t.num_threads == 0) throw std::invalid_argument("tracking_integrator.num_threads must be >= 1 (or -1)");
  if (!(t.temporal_buffer > 0)) throw std::invalid_argument("tracking_integrator.temporal_buffer must be > 0");
  if (t.tsdf_occupancy_threshold == 0) throw std::invalid_argument("tracking_integrator.tsdf_occupancy_threshold must be != 0");
  if (!(t.temporal_window > 0)) throw std::invalid_argument("tracking_integrator.temporal_window must be > 0");
  if (!(volumetric_map.voxel_size > 0) || !(volumetric_map.truncation_distance > 0))
    throw std::invalid_argument("volumetric_map.voxel_size / truncation_distance must be > 0");
  if (volumetric_map.voxels_per_side != 16 && volumetric_map.voxels_per_side != 8)
    throw std::invalid_argument("volumetric_map.voxels_per_side must be 8 or 16 on this device backend");
  if (!motion_detector_type.empty() && motion_detector_type != "FreeSpaceMotionDetector")
    throw std::invalid_argument("unknown motion_detector type '" + motion_detector_type + "'");
  if (!object_extractor_type.empty() && object_extractor_type != "MeshObjectExtractor")
    throw std::invalid_argument("unknown object_extractor type '" + object_extractor_type + "'");
  if (!object_detector_type.empty() && object_detector_type != "ConnectedSemantics" && object_detector_type != "InstanceForwarding")
    throw std::invalid_argument("unknown object_detector type '" + object_detector_type + "'");
  if (!tracker_type.empty() && tracker_type != "MaxIouTracker" && tracker_type != "ExternalTracker")
    throw std::invalid_argument("unknown tracker type '" + tracker_type + "'");
  interpolationFromName(projective_integrator.interpolation_method);
  interpolationFromName(object_extractor.projective_integrator.interpolation_method);
  switchFromName(projective_integrator.alloc_candidate, "block_centre", "camera_offset", "projective_integrator.alloc_candidate");
  switchFromName(projective_integrator.color_blend_weight, "post", "pre", "projective_integrator.color_blend_weight");
  switchFromName(mesh_integrator.attr_source, "nearest", "containing", "mesh_integrator.attr_source");
  switchFromName(object_extractor.projective_integrator.color_blend_weight, "post", "pre", "object_extractor.projective_integrator.color_blend_weight");
  switchFromName(object_extractor.mesh_integrator.attr_source, "nearest", "containing", "object_extractor.mesh_integrator.attr_source");
}

// ---- ActiveWindow ------------------------------------------------------------------------------------------------
static std::vector<ActiveWindow::KhronosSink> instantiateSinks(const std::vector<khronos_amd::YamlNode>& configs);

ActiveWindow::ActiveWindow(const Config& cfg, const OutputQueue::Ptr& output_queue)
    : hydra::ActiveWindowModule(output_queue), config(cfg), frame_data_buffer_(cfg.frame_data_buffer) {
  config.checkValid();
  sinks_ = instantiateSinks(config.khronos_sinks);  // active_window.cpp:80
  khr_config& d = device_config_;
  khr_default_config(&d);
  d.voxel_size = config.volumetric_map.voxel_size;
  d.voxels_per_side = config.volumetric_map.voxels_per_side;
  d.truncation_distance = config.volumetric_map.truncation_distance;
  d.with_semantics = config.volumetric_map.with_semantics;
  d.with_tracking = config.volumetric_map.with_tracking;
  d.num_labels = config.num_labels;
  d.use_weight_dropoff = config.projective_integrator.use_weight_dropoff;
  d.weight_dropoff_epsilon = config.projective_integrator.weight_dropoff_epsilon;
  d.use_constant_weight = config.projective_integrator.use_constant_weight;
  d.max_weight = config.projective_integrator.max_weight;
  d.interpolation_method = interpolationFromName(config.projective_integrator.interpolation_method);
  d.label_confidence = config.projective_integrator.label_confidence;
  d.temporal_buffer = config.tracking_integrator.temporal_buffer;
  d.tsdf_occupancy_threshold = config.tracking_integrator.tsdf_occupancy_threshold;
  d.neighbor_connectivity = config.tracking_integrator.neighbor_connectivity;
  d.temporal_window = config.tracking_integrator.temporal_window;
  d.md_neighbor_connectivity = config.motion_detector.neighbor_connectivity;
  d.md_min_cluster_size = config.motion_detector.min_cluster_size;
  d.md_max_cluster_size = config.motion_detector.max_cluster_size;
  d.md_min_separation_distance = config.motion_detector.min_separation_distance;
  d.md_max_range = config.motion_detector.max_range;
  d.md_min_z_coordinate = config.motion_detector.min_z_coordinate;
  d.mesh_min_weight = config.mesh_integrator.min_weight;
  d.alloc_candidate = switchFromName(config.projective_integrator.alloc_candidate, "block_centre", "camera_offset", "projective_integrator.alloc_candidate");
  d.color_blend_weight = switchFromName(config.projective_integrator.color_blend_weight, "post", "pre", "projective_integrator.color_blend_weight");
  d.mesh_attr_source = switchFromName(config.mesh_integrator.attr_source, "nearest", "containing", "mesh_integrator.attr_source");
  d.mesh_degenerate_eps = config.mesh_integrator.degenerate_eps;
  d.max_blocks = config.max_blocks;
  d.max_frame_pixels = config.max_frame_pixels;
  d.max_mesh_vertices = config.max_mesh_vertices;
  // with an object extractor the buffered frames stay resident in the device ring (FrameDataBuffer role)
  // (+ headroom: a queued or running extraction request holds a COPY of the buffer, i.e. leases on frames the window
  // may already have popped; when the ring is exhausted anyway spinOnce waits for the worker and retries)
  d.num_frame_slots = config.object_extractor_type.empty()
                          ? 2u
                          : static_cast<uint32_t>(config.frame_data_buffer.max_buffer_size + 1 +
                                                  (config.frame_slot_headroom >= 0 ? config.frame_slot_headroom
                                                                                   : 16 * std::max(1, config.extraction_worker.num_workers)));
  d.device = config.device;
  d.rank = config.rank;
  d.world_size = config.world_size;
  d.relaxed_arithmetic = config.exact_arithmetic ? 0 : 1;
  d.max_snapshot_blocks = config.max_snapshot_blocks;
  chk(khr_create(&d, &ctx_), "khr_create");
  if (config.timing_sync_device) {
    khr_ctx* const sc = ctx_;
    hydra::timing::ElapsedTimeRecorder::instance().sync_device = [sc]() { khr_sync(sc); };
  }
  map_ = VolumetricMap(config.volumetric_map, ctx_);

  // member processors as specified in the config; absent ones are the no-op bases (active_window.cpp:83-99)
  if (config.motion_detector_type == "FreeSpaceMotionDetector")
    motion_detector_ = std::make_unique<FreeSpaceMotionDetector>(config.motion_detector);
  else
    motion_detector_ = std::make_unique<MotionDetector>();
  if (config.object_detector_type == "ConnectedSemantics")
    object_detector_ = std::make_unique<ConnectedSemantics>(config.object_detector, map_);
  else if (config.object_detector_type == "InstanceForwarding")
    object_detector_ = std::make_unique<InstanceForwarding>(config.instance_forwarding);
  else
    object_detector_ = std::make_unique<ObjectDetector>();
  if (config.tracker_type == "MaxIouTracker") {
    tracker_ = std::make_unique<MaxIoUTracker>(config.tracker);
  } else if (config.tracker_type == "ExternalTracker") {
    ExternalTracker::Config ec;
    ec.verbosity = config.tracker.verbosity;
    ec.temporal_window = config.tracker.temporal_window;
    ec.min_num_observations = config.tracker.min_num_observations;
    tracker_ = std::make_unique<ExternalTracker>(ec);
  } else {
    tracker_ = std::make_unique<Tracker>();
  }
  if (config.object_extractor_type == "MeshObjectExtractor") {
    const MeshObjectExtractor::Config oec = config.object_extractor;
    extraction_worker_ = std::make_unique<ObjectWorkerPool>(
        config.extraction_worker, [oec, d]() -> std::unique_ptr<ObjectExtractor> { return std::make_unique<MeshObjectExtractor>(oec, d); });
  }
}

ActiveWindow::~ActiveWindow() {
  hydra::timing::ElapsedTimeRecorder::instance().sync_device = nullptr;
  // buffered frames hold leases on frame slots of the context: they go first (frames a sink copied must not outlive the
  // window either)
  if (extraction_worker_) extraction_worker_->stop();  // (requests hold copies of the frame buffer: they go with it)
  pending_frame_.reset();
  frame_data_buffer_.clear();
  extraction_worker_.reset();
  if (ctx_) khr_destroy(ctx_);
}

std::string ActiveWindow::printInfo() const {
  std::ostringstream o;
  o << "ActiveWindow::Config: voxel_size=" << config.volumetric_map.voxel_size
    << " truncation_distance=" << config.volumetric_map.truncation_distance
    << " voxels_per_side=" << config.volumetric_map.voxels_per_side << " with_semantics=" << config.volumetric_map.with_semantics
    << " min_output_separation=" << config.min_output_separation << " motion_detector='" << config.motion_detector_type
    << "' object_extractor='" << config.object_extractor_type << "' temporal_window=" << config.tracking_integrator.temporal_window
    << " sinks=" << sinks_.size();
  return o.str();
}

void ActiveWindow::addKhronosSink(const KhronosSink& sink) {
  if (sink) sinks_.push_back(sink);
}

namespace {
std::mutex& sinkRegistryMutex() {
  static std::mutex mu;
  return mu;
}
std::map<std::string, ActiveWindow::KhronosSinkFactory>& sinkRegistry() {
  static std::map<std::string, ActiveWindow::KhronosSinkFactory> reg;
  return reg;
}
}  // namespace

bool ActiveWindow::registerKhronosSink(const std::string& type, KhronosSinkFactory factory) {
  std::lock_guard<std::mutex> lock(sinkRegistryMutex());
  return sinkRegistry().emplace(type, std::move(factory)).second;
}

// KhronosSink::instantiate(config.khronos_sinks) (active_window.cpp:80)
static std::vector<ActiveWindow::KhronosSink> instantiateSinks(const std::vector<khronos_amd::YamlNode>& configs) {
  std::vector<ActiveWindow::KhronosSink> out;
  for (const auto& node : configs) {
    std::string type;
    node.read("type", type);
    ActiveWindow::KhronosSinkFactory factory;
    {
      std::lock_guard<std::mutex> lock(sinkRegistryMutex());
      auto it = sinkRegistry().find(type);
      if (it != sinkRegistry().end()) factory = it->second;
    }
    if (!factory) {
      std::fprintf(stderr, "[Khronos Active Window] khronos_sinks: no sink type '%s' is registered; entry skipped\n", type.c_str());
      continue;
    }
    if (auto sink = factory(node)) out.push_back(std::move(sink));
  }
  return out;
}

using Timer = hydra::timing::ScopedTimer;

std::shared_ptr<FrameData> ActiveWindow::createData(const hydra::InputPacket& input) const {
  Timer timer("active_window/create_data", latest_stamp_);  // active_window.cpp:269
  // active_window.cpp:268-286: normalise the packet (device: range image, rgba, tiles) and allocate the
  // dynamic / object images (zeroed in the frame slot)
  auto data = std::make_shared<FrameData>();
  InputData& in = data->input;
  in.timestamp_ns = input.timestamp_ns;
  std::memcpy(in.world_T_body, input.world_T_body, sizeof(in.world_T_body));
  hydra::mul4(input.world_T_body, input.body_T_sensor, in.world_T_sensor);
  in.sensor = input.sensor;
  in.ctx = ctx_;
  khr_sensor s{input.sensor.width, input.sensor.height, input.sensor.fx, input.sensor.fy, input.sensor.cx, input.sensor.cy,
               input.sensor.min_range, input.sensor.max_range};
  khr_frame f{};
  f.timestamp_ns = input.timestamp_ns;
  std::memcpy(f.world_T_sensor, in.world_T_sensor, sizeof(f.world_T_sensor));
  f.depth = input.depth;
  f.color = input.color;
  f.label = input.labels;
  in.label_features = input.label_features;
  in.slot = khr_upload_frame(ctx_, &s, &f, input.on_device ? 1 : 0);
  if (in.slot == KHR_ENOMEM && extraction_worker_) {  // ring exhausted by frames pending extractions hold: wait and retry once
    extraction_worker_->join();
    ++num_ring_waits_;
    in.slot = khr_upload_frame(ctx_, &s, &f, input.on_device ? 1 : 0);
  }
  if (in.slot < 0) return nullptr;  // "Input packet preprocessing failed. Skipping frame." (:276-279)
  in.retainSlot();
  return data;
}

void ActiveWindow::updateMap(const FrameData& data) {
  // active_window.cpp:203-215: mask = dynamic_image != 0, integrate with allocation, then tracking update
  Timer timer("active_window/update_map", latest_stamp_, config.timing_sync_device);  // :204
  chk(khr_integrate(ctx_, data.input.slot, /*allocate=*/1, /*use_mask=*/1, /*object_id=*/-1), "khr_integrate");
  {
    Timer t2("integration/tracking", data.input.timestamp_ns, config.timing_sync_device);  // tracking_integrator.cpp:72
    chk(khr_update_tracking(ctx_, data.input.timestamp_ns), "khr_update_tracking");
  }
}

hydra::ActiveWindowOutput::Ptr ActiveWindow::spinOnce(const hydra::InputPacket& input) {
  std::lock_guard<std::mutex> lock(mutex_);
  latest_stamp_ = input.timestamp_ns;
  Timer timer("active_window/all", latest_stamp_, config.timing_sync_device);  // active_window.cpp:121
  // Reference order (active_window.cpp:124-137): data -> motion -> objects -> tracker -> updateMap.  The object
  // detector and tracker are host plugins that neither read nor write what the integration reads, so the
  // device work (normalise, motion detection, integration, tracking) is queued first and they run while the
  // GPU is busy.
  std::shared_ptr<FrameData> data;
  bool fused_output = false;
  const bool plain_motion = !motion_detector_->isDeviceBacked() && config.motion_detector_type.empty();
  if (motion_detector_->isDeviceBacked() || plain_motion) {
    // fused device step (khr_process_frame): the motion detector's host round trip hides behind allocation
    data = std::make_shared<FrameData>();
    InputData& in = data->input;
    in.timestamp_ns = input.timestamp_ns;
    std::memcpy(in.world_T_body, input.world_T_body, sizeof(in.world_T_body));
    hydra::mul4(input.world_T_body, input.body_T_sensor, in.world_T_sensor);
    in.sensor = input.sensor;
    in.ctx = ctx_;
    khr_sensor s{input.sensor.width, input.sensor.height, input.sensor.fx, input.sensor.fy, input.sensor.cx, input.sensor.cy,
                 input.sensor.min_range, input.sensor.max_range};
    khr_frame f{};
    f.timestamp_ns = input.timestamp_ns;
    std::memcpy(f.world_T_sensor, in.world_T_sensor, sizeof(f.world_T_sensor));
    f.depth = input.depth;
    f.color = input.color;
    f.label = input.labels;
    uint32_t flags = KHR_PF_TRACKING | (motion_detector_->isDeviceBacked() ? KHR_PF_MOTION : 0u);
    // (round 6) ConnectedSemantics' kernels only read the frame: queued by the same call on the context's second stream, they run
    // beside the update; object_detector_->processInput below then finds the frame's clusters ready (before: launched and awaited
    // after the fused call had returned, 0.19 ms of every 0.43 ms frame at 1280 x 720)
    if (config.fuse_device_stages && dynamic_cast<ConnectedSemantics*>(object_detector_.get())) flags |= KHR_PF_OBJECTS;
    // the output's device stages (marching cubes beside the snapshot of the updated blocks, archival, flag clearing; :217-237, :169-171)
    // in the same call at the frames where an output is due.  They do not read anything the object detector or the tracker write;
    // a sink is handed the map BEFORE archival in the reference (:152-153 come before :163), so with sinks the stages stay where
    // the reference has them.
    fused_output = config.fuse_device_stages && sinks_.empty() &&
                   !(last_full_upated_ + fromSeconds(config.min_output_separation) > latest_stamp_);
    if (fused_output) flags |= KHR_PF_OUTPUT | KHR_PF_SNAPSHOT;
    if (input.on_device && input.buffers_complete) flags |= KHR_PF_INPUT_READY;
    int n_clusters = 0;
    // create_data + motion_detection/all + update_map (+ integration/tracking) of the reference are ONE fused device call
    // here; the scope is recorded under the reference's outer name
    Timer t_map("active_window/update_map", latest_stamp_, config.timing_sync_device);
    in.label_features = input.label_features;
    in.slot = khr_process_frame(ctx_, &s, &f, input.on_device ? 1 : 0, flags, &n_clusters);
    if (in.slot == KHR_ENOMEM && extraction_worker_) {
      // every frame slot is leased: pending extractions pin frames the window has already dropped.  The reference has no
      // fixed ring and cannot lose a frame this way: wait for the worker (its requests release their frames) and retry once.
      extraction_worker_->join();
      ++num_ring_waits_;
      in.slot = khr_process_frame(ctx_, &s, &f, input.on_device ? 1 : 0, flags, &n_clusters);
    }
    t_map.stop();
    if (in.slot < 0) return nullptr;  // "Input packet preprocessing failed. Skipping frame." (:276-279)
    in.retainSlot();
    data->num_dynamic_clusters = n_clusters;
    if (n_clusters > 0) FreeSpaceMotionDetector::fetchClusters(map_, *data);
  } else {
    data = createData(input);
    if (!data) return nullptr;  // the reference dereferences unconditionally here (latent crash)
    {
      Timer t("motion_detection/all", latest_stamp_, config.timing_sync_device);  // free_space_motion_detector.cpp:75
      motion_detector_->processInput(map_, *data);
    }
    updateMap(*data);
  }
  last_num_dynamic_ = data->num_dynamic_clusters;
  khr_host_trace("aw_device_queued");
  // the previous frame's association (see below), while this frame's kernels run
  completePendingFrame();
  khr_host_trace("aw_prev_associated");
  {
    Timer t("object_detection/all", latest_stamp_);  // connected_semantics.cpp:61, instance_forwarding.cpp:75
    object_detector_->processInput(map_, *data);
  }
  auto* const iou = dynamic_cast<MaxIoUTracker*>(tracker_.get());
  if (config.fuse_device_stages && iou && sinks_.empty()) {
    // (round 6) MaxIoUTracker in two halves (the software pipeline of kop_launch_frame / kop_finish_frame, object_pipeline.cpp): the
    // voxel-set passes of this frame's clusters are queued now; their results are collected, the tracks associated and the frame
    // stored (:130-145) when the NEXT spinOnce has queued its device work -- or before anything looks at the tracks or the frame
    // buffer (getTracks, getLatestFrameData, finishMapping, extractObjects; an output: behind its wait for the archived blocks).  The host no longer waits a device round
    // trip per frame for them.  Sinks are handed the tracks of THIS frame (:153): with sinks the tracker runs in line.
    Timer t("tracking/all", latest_stamp_);
    iou->beginInput(*data);
    pending_frame_ = data;
  } else {
    {
      Timer t("tracking/all", latest_stamp_);  // max_iou_tracker.cpp:200, external_tracker.cpp:68
      tracker_->processInput(*data);
    }
    frame_data_buffer_.trimBuffer(tracker_->getTracks());
    frame_data_buffer_.storeData(data);
  }
  ++num_frames_processed_;
  khr_host_trace("aw_tracker_queued");
  {
    Timer sink_timer("active_window/sinks", latest_stamp_);  // active_window.cpp:152
    for (const auto& sink : sinks_) sink(*data, map_, tracker_->getTracks());
  }

  if (last_full_upated_ + fromSeconds(config.min_output_separation) > latest_stamp_) return nullptr;  // :158-160
  auto output = extractOutputData(*data, config.detach_object_extraction, fused_output);
  // (active_window.cpp:165) the output's copy of the InputData: stamp, poses, sensor, label features AND the images -- as a
  // device-side copy of its own (16 bytes per pixel, one kernel in stream order), not as a lease on the frame's ring slot: outputs
  // wait in the consumer's queue for an unbounded time and the ring is finite
  khr_host_trace("aw_output_extracted");
  output->sensor_data = std::make_shared<InputData>(data->input);
  output->sensor_data->detachFromRing();
  khr_host_trace("aw_sensor_data_copied");
  last_full_upated_ = latest_stamp_;
  if (!fused_output) chk(khr_clear_updated(ctx_), "khr_clear_updated");  // :169-171
  return output;
}

void ActiveWindow::completePendingFrame() const {
  if (!pending_frame_) return;
  std::shared_ptr<FrameData> data = std::move(pending_frame_);
  pending_frame_.reset();
  {
    Timer t("tracking/associate_deferred", data->input.timestamp_ns);
    static_cast<MaxIoUTracker*>(tracker_.get())->completeInput(*data);
  }
  frame_data_buffer_.trimBuffer(tracker_->getTracks());
  frame_data_buffer_.storeData(data);
}

hydra::ActiveWindowOutput::Ptr ActiveWindow::extractOutputData(const FrameData& data, bool threaded, bool device_stages_queued) {
  // active_window.cpp:217-249
  Timer timer("active_window/extract_output", latest_stamp_, config.timing_sync_device);  // :220
  if (!device_stages_queued) chk(khr_generate_mesh(ctx_, 1, 1), "khr_generate_mesh");
  auto output = std::make_shared<hydra::ActiveWindowOutput>();
  output->timestamp_ns = data.input.timestamp_ns;
  for (int r = 0; r < 3; ++r) {
    output->world_t_body[r] = data.input.world_T_body[4 * r + 3];
    for (int c = 0; c < 3; ++c) output->world_R_body[3 * r + c] = data.input.world_T_body[4 * r + c];
  }
  output->map_ctx = ctx_;
  {  // output->setMap(map.cloneUpdated()) (:229): snapshot on the device, in stream order, no host round trip
    khr_snapshot* snap = nullptr;
    if (device_stages_queued) chk(khr_take_snapshot(ctx_, &snap), "khr_take_snapshot");
    else chk(khr_snapshot_updated(ctx_, KHR_SNAP_ALL, config.max_snapshot_blocks, &snap), "khr_snapshot_updated");
    output->setMap(snap);
    output->snapshot_capacity = std::min<int64_t>(config.max_snapshot_blocks ? config.max_snapshot_blocks : config.max_blocks, config.max_blocks);
  }
  // archive after cloning / meshing (:231-237)
  if (config.volumetric_map.with_tracking) {
    std::vector<int32_t>& removed = removed_scratch_;  // (kept: 12 bytes x max_blocks, not allocated and zeroed per output)
    removed.resize(3 * static_cast<size_t>(config.max_blocks));
    int64_t n = 0;
    if (device_stages_queued) chk(khr_last_removed(ctx_, removed.data(), config.max_blocks, &n), "khr_last_removed");
    else chk(khr_reset_inactive(ctx_, removed.data(), config.max_blocks, &n), "khr_reset_inactive");
    output->archived_mesh_indices.resize(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) output->archived_mesh_indices[i] = {removed[3 * i], removed[3 * i + 1], removed[3 * i + 2]};
  }
  // :238-247: inactive tracks go to the worker pool; a blocking call (finishMapping, detach_object_extraction: false) waits
  // for them; the output carries whatever has finished by now (with detached extraction: objects of earlier outputs too)
  khr_host_trace("aw_archived_fetched");
  completePendingFrame();  // (the tracks as of this frame)
  extractInactiveObjects();
  if (extraction_worker_) {
    if (!threaded) extraction_worker_->join();
    extraction_worker_->fill(output->graph_update);
  }
  return output;
}

void ActiveWindow::extractInactiveObjects() {
  // active_window.cpp:251-266: inactive tracks leave the tracker; the track is moved and the frame buffer copied to the
  // worker (the copy keeps the relevant frames -- and their device frame slots -- alive while the window moves on)
  Tracks& tracks = tracker_->getTracks();
  for (auto it = tracks.begin(); it != tracks.end();) {
    if (it->is_active) {
      ++it;
      continue;
    }
    if (extraction_worker_) extraction_worker_->submit(latest_stamp_, std::move(*it), frame_data_buffer_);
    it = tracks.erase(it);
  }
}

void ActiveWindow::finishMapping() {
  std::lock_guard<std::mutex> lock(mutex_);
  // active_window.cpp:176-189: everything inactive, then a blocking output extraction
  completePendingFrame();
  chk(khr_mark_all_inactive(ctx_), "khr_mark_all_inactive");
  for (Track& t : tracker_->getTracks()) t.is_active = false;
  if (!frame_data_buffer_.empty()) extractOutputData(frame_data_buffer_.getLatestData(), false);
}

std::vector<std::shared_ptr<KhronosObjectAttributes>> ActiveWindow::extractObjects() {
  std::vector<std::shared_ptr<KhronosObjectAttributes>> result;
  std::lock_guard<std::mutex> lock(mutex_);
  if (!extraction_worker_) return result;
  completePendingFrame();
  for (const Track& t : tracker_->getTracks()) {  // active_window.cpp:191-201
    auto obj = extraction_worker_->runBlocking(t, frame_data_buffer_);
    if (obj) result.push_back(obj);
  }
  return result;
}

// ---- ObjectWorkerPool (object_worker_pool.cpp:56-146) ---------------------------------------------------------------
ObjectWorkerPool::ObjectWorkerPool(const Config& cfg, const ExtractorFactory& make_extractor) : config(cfg) {
  if (!make_extractor) return;
  const int n = std::max(1, std::min(kMaxWorkers, cfg.num_workers > 0 ? cfg.num_workers : kMaxWorkers));
  for (int i = 0; i < n; ++i) extractors_.push_back(make_extractor());
  for (size_t w = 0; w < extractors_.size(); ++w) workers_.emplace_back([this, w] { workerLoop(w); });
}

ObjectWorkerPool::~ObjectWorkerPool() { stop(); }

void ObjectWorkerPool::stop() {
  {
    std::lock_guard<std::mutex> lock(mutex_);
    should_shutdown_ = true;
  }
  cv_work_.notify_all();
  for (auto& t : workers_)
    if (t.joinable()) t.join();
  workers_.clear();
  std::lock_guard<std::mutex> lock(mutex_);
  queue_.clear();  // (frames of unworked requests are released here)
  outstanding_.store(in_work_, std::memory_order_release);
}

void ObjectWorkerPool::join() {
  // the caller blocks anyway: poll for up to 3 ms before sleeping (an extraction is ~1 ms; waking a thread that sleeps on
  // a condition variable costs 50 - 100 us here, which a caller that joins at output cadence would pay every time)
  const auto t0 = std::chrono::steady_clock::now();
  while (outstanding_.load(std::memory_order_acquire) != 0 &&
         std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(3))
    std::this_thread::yield();
  std::unique_lock<std::mutex> lock(mutex_);
  cv_idle_.wait(lock, [&] { return (queue_.empty() && in_work_ == 0) || should_shutdown_; });
  if (!error_.empty()) {
    const std::string e = error_;
    error_.clear();
    throw std::runtime_error("object extraction worker: " + e);
  }
}

size_t ObjectWorkerPool::numRunning() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return queue_.size() + in_work_;
}

void ObjectWorkerPool::submit(TimeStamp stamp, Track&& track, const FrameDataBuffer& frame_data) {
  if (extractors_.empty()) return;  // :93-95
  auto req = std::unique_ptr<Request>(new Request{stamp, std::move(track), frame_data});
  {
    std::lock_guard<std::mutex> lock(mutex_);
    outstanding_.fetch_add(1, std::memory_order_relaxed);
    queue_.push_back(std::move(req));
  }
  cv_work_.notify_one();
}

std::shared_ptr<KhronosObjectAttributes> ObjectWorkerPool::runBlocking(const Track& track, const FrameDataBuffer& data) {
  if (extractors_.empty()) return nullptr;  // :102-104
  // worker 0's extractor, when that worker is not using it (a device mini-map context serves one extraction at a time)
  std::lock_guard<std::mutex> lock(blocking_mutex_);
  return extractors_[0]->extractObject(track, data);
}

size_t ObjectWorkerPool::fill(std::vector<std::shared_ptr<KhronosObjectAttributes>>& out) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!error_.empty()) {
    const std::string e = error_;
    error_.clear();
    throw std::runtime_error("object extraction worker: " + e);
  }
  const size_t n = output_.size();
  std::move(output_.begin(), output_.end(), std::back_inserter(out));
  output_.clear();
  return n;
}

void ObjectWorkerPool::workerLoop(size_t worker) {
  try {
    extractors_[worker]->prepareThread();
  } catch (...) {
  }
  while (true) {
    std::unique_ptr<Request> req;
    {
      std::unique_lock<std::mutex> lock(mutex_);
      cv_work_.wait(lock, [&] { return should_shutdown_ || !queue_.empty(); });
      if (should_shutdown_) return;
      req = std::move(queue_.front());
      queue_.pop_front();
      ++in_work_;
    }
    std::shared_ptr<KhronosObjectAttributes> attrs;
    std::string err;
    const auto start = std::chrono::steady_clock::now();
    try {
      khr_host_trace("worker_job_begin");
      if (test_delay_ms_ > 0) std::this_thread::sleep_for(std::chrono::milliseconds(test_delay_ms_));  // (tests: a slow extractor)
      if (worker == 0) {
        std::lock_guard<std::mutex> lock(blocking_mutex_);
        attrs = extractors_[0]->extractObject(req->track, req->frame_data);
      } else {
        attrs = extractors_[worker]->extractObject(req->track, req->frame_data);
      }
      khr_host_trace("worker_job_end");
    } catch (const std::exception& e) {
      err = e.what();
    }
    hydra::timing::ElapsedTimeRecorder::instance().record(
        "active_window/extract_object", std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count());  // :139
    req.reset();  // the request's frames (and their device slots) are released before the pool reports idle
    {
      std::lock_guard<std::mutex> lock(mutex_);
      --in_work_;
      if (attrs) output_.emplace_back(std::move(attrs));
      if (!err.empty()) error_ = err;
      outstanding_.fetch_sub(1, std::memory_order_release);
    }
    cv_idle_.notify_all();
  }
}

}  // namespace khronos
