// object_tracking.cpp — host plugins next to the fusion path, restated from
//   khronos/src/active_window/object_detection/connected_semantics.cpp   (clustering itself: device, khr_detect_objects)
//   khronos/src/active_window/tracking/max_iou_tracker.cpp
//   khronos/src/active_window/tracking/external_tracker.cpp
//   khronos/src/active_window/data/track.cpp
// The data-parallel parts (pixel -> voxel grouping, connected components, per-cluster boxes and voxel sets) are
// C-ABI calls into the gfx950 kernels; what remains here is the sequential association logic over a handful of
// clusters and tracks, which the reference also runs on one thread.
#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

#include "active_window.h"

namespace khronos {

namespace {
void chk(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + khr_last_error());
}

std::vector<int> parseIntList(const std::string& scalar) {
  // inline YAML sequence "[1, 2, 3]" (the only form the mapper configs use for label lists)
  std::vector<int> out;
  std::string tok;
  auto flush = [&]() {
    if (!tok.empty()) {
      out.push_back(std::stoi(tok));
      tok.clear();
    }
  };
  for (char ch : scalar) {
    if ((ch >= '0' && ch <= '9') || ch == '-') tok.push_back(ch);
    else flush();
  }
  flush();
  return out;
}
}  // namespace

// ---- Track ------------------------------------------------------------------------------------------------------------
void Track::updateSemantics(const std::optional<SemanticClusterInfo>& other) {
  if (!other) return;
  if (!semantics) {
    semantics = other;
    return;
  }
  const bool other_has_feature = other->feature.size() != 1;
  const bool has_feature = semantics->feature.size() != 1;
  if (has_feature && !other_has_feature) return;  // an existing feature is kept
  if (other_has_feature && !has_feature) {
    semantics->feature = other->feature;
    ++num_features;
    return;
  }
  // running mean of the features (track.cpp:63-69)
  const float total = static_cast<float>(num_features + 1);
  const float w_prev = static_cast<float>(num_features) / total, w_new = 1.0f / total;
  const size_t n = std::min(semantics->feature.size(), other->feature.size());
  for (size_t i = 0; i < n; ++i) semantics->feature[i] = w_prev * semantics->feature[i] + w_new * other->feature[i];
  ++num_features;
}

// ---- ConnectedSemantics -----------------------------------------------------------------------------------------------
ConnectedSemantics::Config ConnectedSemantics::Config::fromYaml(const khronos_amd::YamlNode& m) {
  Config c;
  m.read("verbosity", c.verbosity);
  m.read("use_full_connectivity", c.use_full_connectivity);
  m.read("min_cluster_size", c.min_cluster_size);
  m.read("max_cluster_size", c.max_cluster_size);
  m.read("use_3d", c.use_3d);
  m.read("grid_size", c.grid_size);
  m.read("max_range", c.max_range);
  std::string labels;
  m.read("object_labels", labels);  // extension: the reference reads them from hydra's global label space
  if (!labels.empty()) c.object_labels = parseIntList(labels);
  return c;
}

ConnectedSemantics::ConnectedSemantics(const Config& cfg, const VolumetricMap& map) : config(cfg) {
  if (config.use_3d && !(config.grid_size > 0.f)) throw std::invalid_argument("object_detector.grid_size must be > 0");
  khr_object_detector_config d{};
  d.use_full_connectivity = config.use_full_connectivity;
  d.min_cluster_size = config.min_cluster_size;
  d.max_cluster_size = config.max_cluster_size;
  d.use_3d = config.use_3d;
  d.grid_size = config.grid_size;
  d.max_range = config.max_range;
  std::vector<int32_t> labels(config.object_labels.begin(), config.object_labels.end());
  d.object_labels = labels.data();
  d.n_object_labels = static_cast<int32_t>(labels.size());
  chk(khr_configure_object_detector(map.ctx(), &d), "khr_configure_object_detector");
}

void ConnectedSemantics::processInput(const VolumetricMap& map, FrameData& data) {
  // connected_semantics.cpp:59-69; FrameData::object_image stays in the device frame slot
  const int n = khr_detect_objects(map.ctx(), data.input.slot);
  chk(n, "khr_detect_objects");
  data.semantic_clusters.clear();
  if (n == 0) return;
  std::vector<khr_cluster> cl(static_cast<size_t>(n));
  chk(khr_get_semantic_clusters(map.ctx(), data.input.slot, cl.data(), n), "khr_get_semantic_clusters");
  data.semantic_clusters.reserve(cl.size());
  for (const khr_cluster& k : cl) {
    MeasurementCluster m;
    m.id = k.id;
    m.num_pixels = static_cast<size_t>(k.num_pixels_listed);
    m.bounding_box.include(k.bbox_min);
    m.bounding_box.include(k.bbox_max);
    for (int d = 0; d < 3; ++d) m.centroid[d] = k.centroid[d];
    m.semantics = SemanticClusterInfo(k.semantic_id);
    data.semantic_clusters.push_back(std::move(m));
  }
}

// ---- InstanceForwarding -----------------------------------------------------------------------------------------------
InstanceForwarding::Config InstanceForwarding::Config::fromYaml(const khronos_amd::YamlNode& m) {
  Config c;  // declare_config, instance_forwarding.cpp:47-61
  m.read("verbosity", c.verbosity);
  m.read("max_range", c.max_range);
  m.read("min_cluster_size", c.min_cluster_size);
  m.read("max_cluster_size", c.max_cluster_size);
  m.read("min_object_volume", c.min_object_volume);
  m.read("max_object_volume", c.max_object_volume);
  m.read("max_background_score", c.max_background_score);
  m.read("max_instance_id", c.max_instance_id);
  return c;
}

InstanceForwarding::InstanceForwarding(const Config& cfg)
    : config(cfg), filter_by_volume_(cfg.min_object_volume > 0.0 || cfg.max_object_volume > 0.0) {
  if (config.max_instance_id < 1 || config.max_instance_id > 65535) throw std::invalid_argument("object_detector.max_instance_id must be in [1, 65535]");
}

float InstanceForwarding::bestBackgroundScore(const std::vector<std::vector<float>>& background, const std::vector<float>& feature) {
  // EmbeddingGroup::getBestScore with hydra::CosineDistance (un-vendored, ASSUMPTIONS.md A.9): the largest cosine
  // similarity between the feature and a background prompt
  float best = -std::numeric_limits<float>::infinity();
  for (const auto& b : background) {
    float dot = 0.f, na = 0.f, nb = 0.f;
    const size_t n = std::min(b.size(), feature.size());
    for (size_t i = 0; i < n; ++i) {
      dot += b[i] * feature[i];
      na += b[i] * b[i];
      nb += feature[i] * feature[i];
    }
    const float s = dot / (std::sqrt(na) * std::sqrt(nb));
    if (s > best) best = s;
  }
  return best;
}

void InstanceForwarding::processInput(const VolumetricMap& map, FrameData& data) {
  // instance_forwarding.cpp:73-149
  data.semantic_clusters.clear();
  // background filter (:93-103): an id without a feature, or whose best score exceeds the limit, contributes no pixels.
  // The ids a label image can carry are bounded by the table size, so "no feature" ids are listed too.
  std::vector<int32_t> background;
  if (!config.background_embeddings.empty()) {
    for (int id = 1; id <= config.max_instance_id; ++id) {
      const auto f = data.input.label_features.find(id);
      if (f == data.input.label_features.end() ||
          static_cast<double>(bestBackgroundScore(config.background_embeddings, f->second)) > config.max_background_score)
        background.push_back(id);
    }
  }
  std::vector<khr_cluster> table(static_cast<size_t>(config.max_instance_id) + 1);
  const int n = khr_forward_instances(map.ctx(), data.input.slot, config.max_range, background.empty() ? nullptr : background.data(),
                                      static_cast<int>(background.size()), config.max_instance_id, table.data());
  chk(n, "khr_forward_instances");
  // clusters in ascending id order (the reference iterates an unordered_map, :117; ASSUMPTIONS.md C.4)
  for (const khr_cluster& k : table) {
    if (k.num_pixels_listed == 0) continue;
    const int size = static_cast<int>(k.num_pixels_listed);
    if (size < config.min_cluster_size || (config.max_cluster_size > 0 && size > config.max_cluster_size)) continue;  // :119-122
    MeasurementCluster m;
    m.id = k.id;
    m.num_pixels = static_cast<size_t>(k.num_pixels_listed);
    m.bounding_box.include(k.bbox_min);
    m.bounding_box.include(k.bbox_max);
    for (int d = 0; d < 3; ++d) m.centroid[d] = k.centroid[d];
    if (filter_by_volume_) {  // :128-135
      const double volume = m.bounding_box.volume();
      if (volume < config.min_object_volume || (config.max_object_volume > 0.0 && volume > config.max_object_volume)) continue;
    }
    if (data.input.label_features.empty()) m.semantics = SemanticClusterInfo(k.id);  // closed set (:138-140)
    const auto f = data.input.label_features.find(k.id);
    if (f != data.input.label_features.end()) m.semantics = SemanticClusterInfo(k.id, f->second);  // (:142-145)
    data.semantic_clusters.push_back(std::move(m));
  }
}

// ---- MaxIoUTracker ---------------------------------------------------------------------------------------------------------
MaxIoUTracker::Config MaxIoUTracker::Config::fromYaml(const khronos_amd::YamlNode& m) {
  Config c;
  m.read("verbosity", c.verbosity);
  std::string s;
  m.read("track_by", s);
  if (!s.empty()) {
    if (s == "pixels") c.track_by = TrackBy::kPixels;
    else if (s == "voxels") c.track_by = TrackBy::kVoxels;
    else if (s == "bounding_box") c.track_by = TrackBy::kBouningBox;
    else throw std::invalid_argument("tracker.track_by must be one of 'pixels', 'voxels', 'bounding_box'");
  }
  s.clear();
  m.read("semantic_association", s);
  if (!s.empty()) {
    if (s == "assign_cluster") c.semantic_association = SemanticAssociation::kAssignCluster;
    else if (s == "assign_track") c.semantic_association = SemanticAssociation::kAssignTrack;
    else throw std::invalid_argument("tracker.semantic_association must be 'assign_cluster' or 'assign_track'");
  }
  m.read("min_semantic_iou", c.min_semantic_iou);
  m.read("min_cosine_sim", c.min_cosine_sim);
  m.read("min_cross_iou", c.min_cross_iou);
  m.read("max_dynamic_distance", c.max_dynamic_distance);
  m.read("temporal_window", c.temporal_window);
  m.read("min_num_observations", c.min_num_observations);
  m.read("voxel_size", c.voxel_size);
  return c;
}

MaxIoUTracker::MaxIoUTracker(const Config& cfg) : config(cfg) {
  // max_iou_tracker.cpp:143-147
  auto in = [](float v, float lo, float hi) { return v >= lo && v <= hi; };
  if (!in(config.min_cross_iou, 0.f, 1.f)) throw std::invalid_argument("tracker.min_cross_iou must be in [0, 1]");
  if (!in(config.min_semantic_iou, 0.f, 1.f)) throw std::invalid_argument("tracker.min_semantic_iou must be in [0, 1]");
  if (!in(config.min_cosine_sim, -1.f, 1.f)) throw std::invalid_argument("tracker.min_cosine_sim must be in [-1, 1]");
  if (!(config.temporal_window > 0.f)) throw std::invalid_argument("tracker.temporal_window must be > 0");
  if (!(config.voxel_size > 0.f)) throw std::invalid_argument("tracker.voxel_size must be > 0");
}

void MaxIoUTracker::processInput(FrameData& data) {
  beginInput(data);  // max_iou_tracker.cpp:187-203
  completeInput(data);
}

void MaxIoUTracker::beginInput(FrameData& data) { launchTrackMeasurements(data); }

void MaxIoUTracker::completeInput(FrameData& data) {
  processing_stamp_ = data.input.timestamp_ns;
  finishTrackMeasurements(data);
  current_ = &data;
  pix_inter_.clear();
  pix_key_.clear();
  pix_max_id_ = 0;
  for (const auto& c : data.semantic_clusters) pix_max_id_ = std::max(pix_max_id_, c.id);
  associateTracks(data);
  current_ = nullptr;
  updateTrackingDuration();
}

void MaxIoUTracker::setupTrackMeasurements(FrameData& data) const {
  launchTrackMeasurements(data);
  finishTrackMeasurements(data);
}

// max_iou_tracker.cpp:464-487.  Bounding boxes arrive with the clusters (device reduction); the voxel sets of ALL
// clusters of an id image are one device pass, enqueued here and collected in finishTrackMeasurements.  Without a
// device frame (unit tests) the clusters are used as given.
void MaxIoUTracker::launchTrackMeasurements(FrameData& data) const {
  if (config.track_by != Config::TrackBy::kVoxels || !data.input.ctx || data.input.slot < 0) return;
  if (!data.semantic_clusters.empty())
    chk(khr_cluster_voxels_launch(data.input.ctx, data.input.slot, 1, config.voxel_size), "khr_cluster_voxels_launch");
  if (!data.dynamic_clusters.empty())
    chk(khr_cluster_voxels_launch(data.input.ctx, data.input.slot, 0, config.voxel_size), "khr_cluster_voxels_launch");
}

void MaxIoUTracker::finishTrackMeasurements(FrameData& data) const {
  if (config.track_by != Config::TrackBy::kVoxels || !data.input.ctx || data.input.slot < 0) return;
  auto fill = [&](std::vector<MeasurementCluster>& clusters, int which) {
    for (auto& c : clusters) c.voxels.clear();
    if (clusters.empty()) return;
    if (scratch_ids_.empty()) {
      scratch_ids_.resize(1u << 16);
      scratch_voxels_.resize(3u << 16);
    }
    int64_t n = khr_cluster_voxels_fetch(data.input.ctx, which, scratch_ids_.data(), scratch_voxels_.data(),
                                         static_cast<int64_t>(scratch_ids_.size()));
    chk(static_cast<int>(std::min<int64_t>(n, 0)), "khr_cluster_voxels_fetch");
    if (n > static_cast<int64_t>(scratch_ids_.size())) {
      scratch_ids_.resize(static_cast<size_t>(n));
      scratch_voxels_.resize(static_cast<size_t>(3 * n));
      n = khr_cluster_voxels_fetch(data.input.ctx, which, scratch_ids_.data(), scratch_voxels_.data(), n);
      chk(static_cast<int>(std::min<int64_t>(n, 0)), "khr_cluster_voxels_fetch");
    }
    // pairs are sorted by (id, x, y, z)
    const auto ib = scratch_ids_.begin(), ie = scratch_ids_.begin() + n;
    for (auto& c : clusters) {
      const auto lo = std::lower_bound(ib, ie, c.id), hi = std::upper_bound(ib, ie, c.id);
      c.voxels.reserve(static_cast<size_t>(hi - lo));
      for (auto it = lo; it != hi; ++it) {
        const size_t k = static_cast<size_t>(it - ib);
        c.voxels.push_back({scratch_voxels_[3 * k], scratch_voxels_[3 * k + 1], scratch_voxels_[3 * k + 2]});
      }
    }
  };
  fill(data.semantic_clusters, 1);
  fill(data.dynamic_clusters, 0);
}

void MaxIoUTracker::associateTracks(const FrameData& data) {
  hydra::timing::ScopedTimer timer("tracking/associate", processing_stamp_);  // max_iou_tracker.cpp:217
  associateDynamicTracks(data);   // max_iou_tracker.cpp:205-218
  associateSemanticTracks(data);
}

void MaxIoUTracker::computeCentroid(const MeasurementCluster& cluster, float* centroid) const {
  // max_iou_tracker.cpp:540-563
  centroid[0] = centroid[1] = centroid[2] = 0.f;
  if (config.track_by == Config::TrackBy::kBouningBox) {
    for (int d = 0; d < 3; ++d) centroid[d] = cluster.bounding_box.center(d);
    return;
  }
  if (config.track_by == Config::TrackBy::kPixels) {
    // mean of the cluster's vertices (:543-549); reduced on the device with the cluster (float sums in reduction order)
    for (int d = 0; d < 3; ++d) centroid[d] = cluster.centroid[d];
    return;
  }
  // voxels: mean of the voxel centres (grid_.toPoint); summed in sorted voxel order (the reference iterates an
  // unordered set, ASSUMPTIONS.md C.4)
  for (const GlobalIndex& v : cluster.voxels)
    for (int d = 0; d < 3; ++d) centroid[d] += (static_cast<float>(v[d]) + 0.5f) * config.voxel_size;
  const float n = static_cast<float>(cluster.voxels.size());
  for (int d = 0; d < 3; ++d) centroid[d] = centroid[d] / n;
}

float MaxIoUTracker::computeIoUVoxels(const std::vector<GlobalIndex>& a, const std::vector<GlobalIndex>& b) {
  // max_iou_tracker.cpp:565-576 on two sorted sets
  float intersection = 0.f;
  size_t i = 0, j = 0;
  while (i < a.size() && j < b.size()) {
    if (a[i] == b[j]) {
      intersection += 1.f;
      ++i;
      ++j;
    } else if (a[i] < b[j]) {
      ++i;
    } else {
      ++j;
    }
  }
  return intersection / (static_cast<float>(a.size() + b.size()) - intersection);
}

float MaxIoUTracker::computeIoUBoundingBox(const BoundingBox& a, const BoundingBox& b) {
  // spark_dsg::BoundingBox::computeIoU for axis-aligned boxes (un-vendored; ASSUMPTIONS.md A.7)
  if (!a.valid || !b.valid) return 0.f;
  float inter = 1.f;
  for (int d = 0; d < 3; ++d) {
    const float lo = std::max(a.min[d], b.min[d]), hi = std::min(a.max[d], b.max[d]);
    if (!(hi > lo)) return 0.f;
    inter *= hi - lo;
  }
  const float uni = a.volume() + b.volume() - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

// track_by = pixels: intersections of the re-projected points of tracks_[track_index] with the object-image clusters of the
// frame being associated (computeIoUPixels, max_iou_tracker.cpp:578-600).  Rows are computed in batches of up to 32 tracks
// per device call; a track created or updated during this frame's association gets a fresh row on demand.
void MaxIoUTracker::ensurePixelIntersections(size_t track_index) const {
  const auto keyOf = [](const Track& t) { return (static_cast<uint64_t>(static_cast<uint32_t>(t.id)) << 40) ^ t.last_seen; };
  if (pix_inter_.size() < tracks_.size()) {
    pix_inter_.resize(tracks_.size());
    pix_key_.resize(tracks_.size(), ~0ull);
  }
  const Track& want = tracks_[track_index];
  if (pix_key_[track_index] == keyOf(want) && !pix_inter_[track_index].empty()) return;
  if (!current_ || !current_->input.ctx || current_->input.slot < 0 || pix_max_id_ < 1) return;
  // the wanted track plus every other track whose row is stale, up to the per-call limit
  std::vector<size_t> batch = {track_index};
  for (size_t t = 0; t < tracks_.size() && batch.size() < 32; ++t)
    if (t != track_index && tracks_[t].last_pixels.slot >= 0 && pix_key_[t] != keyOf(tracks_[t])) batch.push_back(t);
  std::vector<khr_pixel_ref> refs;
  std::vector<size_t> owners;
  for (size_t t : batch) {
    const Track::PixelRef& r = tracks_[t].last_pixels;
    if (r.slot < 0 || r.ctx != current_->input.ctx) continue;
    refs.push_back({r.slot, r.which, r.id});
    owners.push_back(t);
  }
  if (refs.empty()) return;
  const size_t stride = static_cast<size_t>(pix_max_id_) + 1;
  std::vector<uint32_t> n_points(refs.size()), inter(refs.size() * stride);
  chk(khr_pixel_iou(current_->input.ctx, current_->input.slot, refs.data(), static_cast<int>(refs.size()), pix_max_id_, n_points.data(),
                    inter.data()),
      "khr_pixel_iou");
  for (size_t k = 0; k < owners.size(); ++k) {
    pix_inter_[owners[k]].assign(inter.begin() + static_cast<std::ptrdiff_t>(k * stride), inter.begin() + static_cast<std::ptrdiff_t>((k + 1) * stride));
    pix_inter_[owners[k]].push_back(n_points[k]);  // last entry: last_points.size() as the device counted it
    pix_key_[owners[k]] = keyOf(tracks_[owners[k]]);
  }
}

float MaxIoUTracker::computeIoU(const MeasurementCluster& cluster, const Track& track) const {
  if (config.track_by == Config::TrackBy::kPixels) {
    const size_t ti = static_cast<size_t>(&track - tracks_.data());
    ensurePixelIntersections(ti);
    float intersection = 0.f;
    size_t n_points = track.last_pixels.num_points;
    if (ti < pix_inter_.size() && !pix_inter_[ti].empty()) {
      const std::vector<uint32_t>& row = pix_inter_[ti];
      if (cluster.id >= 0 && static_cast<size_t>(cluster.id) + 1 < row.size()) intersection = static_cast<float>(row[static_cast<size_t>(cluster.id)]);
      n_points = row.back();  // the pixels that carry the id in the source image (dynamic clusters: the painted set)
    }
    // intersection / (cluster.pixels.size() + track.last_points.size() - intersection)  (:599)
    return intersection / (static_cast<float>(cluster.num_pixels + n_points) - intersection);
  }
  return config.track_by == Config::TrackBy::kVoxels ? computeIoUVoxels(cluster.voxels, track.last_voxels)
                                                     : computeIoUBoundingBox(track.last_bounding_box, cluster.bounding_box);
}

void MaxIoUTracker::associateDynamicTracks(const FrameData& data) {
  // max_iou_tracker.cpp:220-272: nearest unassigned dynamic cluster within max_dynamic_distance, per track in order
  std::unordered_set<int> associated;
  for (Track& track : tracks_) {
    if (!track.is_dynamic) continue;
    float best_distance = config.max_dynamic_distance;
    const MeasurementCluster* best = nullptr;
    float best_centroid[3] = {0, 0, 0};
    for (const auto& cluster : data.dynamic_clusters) {
      if (associated.count(cluster.id)) continue;
      float c[3];
      computeCentroid(cluster, c);
      const float dx = c[0] - track.last_centroid[0], dy = c[1] - track.last_centroid[1], dz = c[2] - track.last_centroid[2];
      const float distance = std::sqrt(dx * dx + dy * dy + dz * dz);
      if (distance < best_distance) {
        best = &cluster;
        best_distance = distance;
        for (int d = 0; d < 3; ++d) best_centroid[d] = c[d];
      }
    }
    if (best) {
      associated.insert(best->id);
      updateTrack(*best, track, true);
      for (int d = 0; d < 3; ++d) track.last_centroid[d] = best_centroid[d];
    }
  }
  for (const auto& cluster : data.dynamic_clusters) {
    if (associated.count(cluster.id)) continue;
    Track& track = addNewTrack(cluster, true);
    computeCentroid(cluster, track.last_centroid);
  }
}

namespace {
// semanticsMatch (max_iou_tracker.cpp:102-135)
bool semanticsMatch(const std::optional<SemanticClusterInfo>& lhs, const std::optional<SemanticClusterInfo>& rhs, float min_cosine_sim) {
  if (lhs.has_value() != rhs.has_value()) return false;
  if (!lhs && !rhs) return true;
  if (lhs->category_id != rhs->category_id) return false;
  if (lhs->feature.size() != rhs->feature.size()) return false;
  if (lhs->feature.size() == 1) return true;  // closed set: no features
  // hydra::CosineDistance::score (un-vendored): a.b / (|a| |b|)
  float dot = 0.f, na = 0.f, nb = 0.f;
  for (size_t i = 0; i < lhs->feature.size(); ++i) {
    dot += lhs->feature[i] * rhs->feature[i];
    na += lhs->feature[i] * lhs->feature[i];
    nb += rhs->feature[i] * rhs->feature[i];
  }
  const float sim = dot / (std::sqrt(na) * std::sqrt(nb));
  return !(sim < min_cosine_sim);
}
}  // namespace

void MaxIoUTracker::associateSemanticTracks(const FrameData& data) {
  // max_iou_tracker.cpp:274-333: semantic clusters first go to dynamic tracks (cross IoU), then to static tracks
  std::unordered_set<int> associated;
  for (Track& track : tracks_) {
    if (!track.is_dynamic) continue;
    float best_iou = config.min_cross_iou;
    const MeasurementCluster* best = nullptr;
    for (const auto& cluster : data.semantic_clusters) {
      if (associated.count(cluster.id)) continue;
      const float iou = computeIoU(cluster, track);
      if (iou > best_iou) {
        best = &cluster;
        best_iou = iou;
      }
    }
    if (best) {
      associated.insert(best->id);
      if (track.last_seen < processing_stamp_) updateTrack(*best, track, false);
      else track.observations.back().semantic_cluster_id = best->id;  // already updated by the dynamic association
    }
  }
  if (config.semantic_association == Config::SemanticAssociation::kAssignCluster) assignClustersToStaticTrack(data, associated);
  else assignStaticTracksToCluster(data, associated);
}

void MaxIoUTracker::assignClustersToStaticTrack(const FrameData& data, std::unordered_set<int>& associated) {
  // max_iou_tracker.cpp:335-386: every static track greedily takes its best matching free cluster
  for (Track& track : tracks_) {
    if (track.is_dynamic) continue;
    float best_iou = config.min_semantic_iou;
    const MeasurementCluster* best = nullptr;
    for (const auto& cluster : data.semantic_clusters) {
      if (associated.count(cluster.id)) continue;
      if (!semanticsMatch(cluster.semantics, track.semantics, config.min_cosine_sim)) continue;
      const float iou = computeIoU(cluster, track);
      if (iou > best_iou) {
        best = &cluster;
        best_iou = iou;
      }
    }
    if (best) {
      associated.insert(best->id);
      updateTrack(*best, track, false);
    }
  }
  for (const auto& cluster : data.semantic_clusters)
    if (!associated.count(cluster.id)) addNewTrack(cluster, false);
}

void MaxIoUTracker::assignStaticTracksToCluster(const FrameData& data, std::unordered_set<int>& associated) {
  // max_iou_tracker.cpp:388-433: every free cluster takes the FIRST static track that matches well enough
  for (const auto& cluster : data.semantic_clusters) {
    if (associated.count(cluster.id)) continue;
    bool assigned = false;
    // tracks created for earlier clusters of this frame are candidates too, as in the reference
    for (size_t t = 0; t < tracks_.size(); ++t) {
      Track& track = tracks_[t];
      if (track.is_dynamic) continue;
      if (!semanticsMatch(cluster.semantics, track.semantics, config.min_cosine_sim)) continue;
      const float iou = computeIoU(cluster, track);
      if (iou < config.min_semantic_iou) continue;
      assigned = true;
      associated.insert(cluster.id);
      updateTrack(cluster, track, false);
      break;
    }
    if (!assigned) addNewTrack(cluster, false);
  }
}

Track& MaxIoUTracker::addNewTrack(const MeasurementCluster& observation, bool is_dynamic) {
  tracks_.emplace_back();  // max_iou_tracker.cpp:489-500
  Track& track = tracks_.back();
  track.is_dynamic = is_dynamic;
  track.id = current_track_id_++;
  track.first_seen = processing_stamp_;
  updateTrack(observation, track, is_dynamic);
  return track;
}

void MaxIoUTracker::updateTrack(const MeasurementCluster& observation, Track& track, bool is_observation_dynamic) const {
  // max_iou_tracker.cpp:502-546
  if (config.track_by == Config::TrackBy::kPixels) {
    // track.last_points = the vertices of observation.pixels (:504-510): named, not copied
    Track::PixelRef r;
    if (current_ && current_->input.ctx && current_->input.slot >= 0) {
      r.ctx = current_->input.ctx;
      r.slot = current_->input.slot;
      r.which = is_observation_dynamic ? 0 : 1;
      r.id = observation.id;
      r.lease = current_->input.slot_lease;  // the frame's own lease object (FrameData keeps the slot; so does this copy)
    }
    r.num_points = observation.num_pixels;
    track.last_pixels = std::move(r);
  } else if (config.track_by == Config::TrackBy::kVoxels) {
    track.last_voxels = observation.voxels;
    track.last_voxel_size = config.voxel_size;
  }
  track.last_bounding_box = observation.bounding_box;
  if (!is_observation_dynamic) {  // dynamic observations never overwrite the track's semantics
    if (!track.semantics) track.semantics = observation.semantics;
    else track.updateSemantics(observation.semantics);
  }
  track.last_seen = processing_stamp_;
  track.observations.push_back({processing_stamp_, !is_observation_dynamic ? observation.id : -1,
                                is_observation_dynamic ? observation.id : -1});
  track.confidence = std::min(static_cast<float>(track.observations.size()) / static_cast<float>(config.min_num_observations * 2), 1.f);
}

void MaxIoUTracker::updateTrackingDuration() {
  // max_iou_tracker.cpp:548-554 (unsigned stamps, as in the reference)
  const TimeStamp min_time = processing_stamp_ - fromSeconds(config.temporal_window);
  for (Track& track : tracks_) track.is_active = track.last_seen >= min_time;
}

// ---- ExternalTracker ------------------------------------------------------------------------------------------------------
ExternalTracker::Config ExternalTracker::Config::fromYaml(const khronos_amd::YamlNode& m) {
  Config c;
  m.read("verbosity", c.verbosity);
  m.read("temporal_window", c.temporal_window);
  m.read("min_num_observations", c.min_num_observations);
  return c;
}

ExternalTracker::ExternalTracker(const Config& cfg) : config(cfg) {
  if (!(config.temporal_window > 0.f)) throw std::invalid_argument("tracker.temporal_window must be > 0");
}

void ExternalTracker::processInput(FrameData& data) {
  processing_stamp_ = data.input.timestamp_ns;  // external_tracker.cpp:59-77 (bounding boxes arrive with the clusters)
  auto update = [&](const MeasurementCluster& observation, Track& track) {  // :126-134
    track.updateSemantics(observation.semantics);
    track.last_seen = processing_stamp_;
    track.observations.push_back({processing_stamp_, observation.id, -1});
    track.confidence = std::min(static_cast<float>(track.observations.size()) / static_cast<float>(config.min_num_observations * 2), 1.f);
  };
  // associateTracks (:79-115): a track follows the cluster that carries its id
  std::unordered_set<int> associated;
  for (Track& track : tracks_) {
    if (track.is_dynamic) continue;
    for (const auto& cluster : data.semantic_clusters) {
      if (associated.count(cluster.id)) continue;
      if (track.id == cluster.id) {
        associated.insert(cluster.id);
        update(cluster, track);
        break;
      }
    }
  }
  for (const auto& cluster : data.semantic_clusters) {
    if (associated.count(cluster.id)) continue;
    tracks_.emplace_back();  // addNewTrack (:117-123)
    Track& track = tracks_.back();
    track.is_dynamic = false;
    track.id = cluster.id;
    track.first_seen = processing_stamp_;
    update(cluster, track);
  }
  const TimeStamp min_time = processing_stamp_ - fromSeconds(config.temporal_window);  // :136-141
  for (Track& track : tracks_) track.is_active = track.last_seen >= min_time;
}

}  // namespace khronos
