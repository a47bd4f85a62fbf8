// change_detection.cpp — see change_detection.h.
#include "change_detection.h"

#include <algorithm>
#include <limits>
#include <unordered_set>

namespace khronos {

std::vector<ObjectChange>::iterator ObjectChanges::find(NodeId id) {
  return std::find_if(begin(), end(), [id](const ObjectChange& c) { return c.node_id == id; });
}

RayBackgroundChangeDetector::RayBackgroundChangeDetector(const Config& cfg, std::shared_ptr<const RayVerificator> ray_verificator,
                                                         std::shared_ptr<const RayChangeDetector> ray_change_detector)
    : config(cfg),
      ray_verificator_(std::move(ray_verificator)),
      ray_change_detector_(std::move(ray_change_detector)),
      time_filtering_threshold_ns_(static_cast<uint64_t>(cfg.time_filtering_threshold * 1e9)) {}

namespace {
ChangeState stateOf(const RayChangeDetector::ChangeResult& r) {  // ray_background_change_detector.cpp:96-102
  if (r.closest_absent) return ChangeState::kAbsent;
  if (r.furthest_persistent) return ChangeState::kPersistent;
  return ChangeState::kUnobserved;
}
}  // namespace

size_t RayBackgroundChangeDetector::detectChanges(const std::vector<float>& vertex_positions, const std::vector<uint64_t>& vertex_stamps,
                                                  const std::vector<size_t>& reobserved_vertices, BackgroundChanges& changes) const {
  const size_t n = vertex_stamps.size();
  if (vertex_positions.size() != 3 * n) throw std::invalid_argument("RayBackgroundChangeDetector: 3 coordinates per vertex expected");
  // the queries of this call: the new vertices (:66-68), then the re-observed ones that exist (:72-81) -- ONE device pass
  std::vector<size_t> which;
  for (size_t i = changes.size(); i < n; ++i) which.push_back(i);
  const size_t n_new = which.size();
  for (const size_t i : reobserved_vertices)
    if (i < n) which.push_back(i);
  std::vector<float> pts(3 * which.size());
  std::vector<uint64_t> earliest(which.size()), latest(which.size(), std::numeric_limits<uint64_t>::max());
  std::vector<uint8_t> forward(which.size(), 1);  // detectChanges(check_result, true) (:94)
  for (size_t q = 0; q < which.size(); ++q) {
    const size_t i = which[q];
    std::copy(vertex_positions.begin() + 3 * i, vertex_positions.begin() + 3 * i + 3, pts.begin() + 3 * q);
    earliest[q] = vertex_stamps[i] + time_filtering_threshold_ns_;  // check(pos, timestamp + threshold) (:91-92)
  }
  const auto results = ray_change_detector_->detectChangesMany(*ray_verificator_, pts, earliest, latest, forward);
  changes.reserve(n);
  for (size_t q = 0; q < n_new; ++q) changes.push_back(stateOf(results[q]));
  size_t num_updates = 0;
  for (size_t q = n_new; q < which.size(); ++q) {
    const ChangeState s = stateOf(results[q]);
    if (s != changes.at(which[q])) ++num_updates;
    changes.at(which[q]) = s;
  }
  return num_updates;
}

ChangeState RayBackgroundChangeDetector::checkVertex(const float* position, uint64_t stamp) const {
  const RayVerificator::CheckResult check = ray_verificator_->check(position, stamp + time_filtering_threshold_ns_);
  return stateOf(ray_change_detector_->detectChanges(check, true));
}

RayObjectChangeDetector::RayObjectChangeDetector(const Config& cfg, std::shared_ptr<const RayVerificator> ray_verificator,
                                                 std::shared_ptr<const RayChangeDetector> ray_change_detector)
    : config(cfg),
      ray_verificator_(std::move(ray_verificator)),
      ray_change_detector_(std::move(ray_change_detector)),
      time_filtering_threshold_ns_(static_cast<uint64_t>(cfg.time_filtering_threshold * 1e9)) {
  if (cfg.query_subsampling < 1) throw std::invalid_argument("RayObjectChangeDetector: query_subsampling must be >= 1");
}

void RayObjectChangeDetector::detectChanges(const std::vector<Object>& objects, const std::vector<NodeId>& reobserved_objects,
                                            const RPGOMerges& rpgo_merges, ObjectChanges& changes) const {
  for (const NodeId id : reobserved_objects) {  // (:66-72) their states are recomputed
    auto it = changes.find(id);
    if (it != changes.end()) changes.erase(it);
  }
  std::unordered_set<NodeId> existing;
  for (const ObjectChange& c : changes) existing.insert(c.node_id);
  for (const Object& o : objects) {
    if (existing.count(o.node_id) || !o.attributes) continue;
    if (!o.attributes->trajectory_positions.empty()) continue;  // dynamic objects (:86-89)
    ObjectChange& change = changes.emplace_back();
    change.node_id = o.node_id;
    checkObjectMerge(rpgo_merges, change);
    checkObjectObservation(*o.attributes, change);
  }
}

void RayObjectChangeDetector::checkObjectMerge(const RPGOMerges& rpgo_merges, ObjectChange& change) const {
  const auto it = std::find_if(rpgo_merges.begin(), rpgo_merges.end(), [&](const RPGOMerge& m) { return m.from_node == change.node_id; });
  if (it != rpgo_merges.end() && it->is_valid) change.merged_id = it->to_node;
}

void RayObjectChangeDetector::checkObjectObservation(const hydra::KhronosObjectAttributes& attrs, ObjectChange& change) const {
  if (attrs.mesh.points.empty() || attrs.first_observed_ns.empty() || attrs.last_observed_ns.empty()) return;
  // every sub-sampled vertex is queried twice (:127-134): before the object was first seen and after it was last seen.  Both
  // sets go to the device as ONE batch of 2 m points; the stamp lists of each half are merged in query order (CheckResult::merge)
  std::vector<float> pts;
  // (mesh.pos(i) + bounding_box.world_P_center: the extractor stores the vertices in the bounding-box frame, mesh_object_extractor.cpp:299-302)
  const size_t n_vertices = attrs.mesh.points.size() / 3;
  for (size_t i = 0; i < n_vertices; i += static_cast<size_t>(config.query_subsampling))
    for (int d = 0; d < 3; ++d) pts.push_back(attrs.mesh.points[3 * i + static_cast<size_t>(d)] + attrs.bounding_box.center(d));
  const size_t m = pts.size() / 3;
  std::vector<float> both(pts);
  both.insert(both.end(), pts.begin(), pts.end());
  std::vector<uint64_t> earliest(2 * m, 0ul), latest(2 * m, std::numeric_limits<uint64_t>::max());
  for (size_t q = 0; q < m; ++q) {
    latest[q] = attrs.first_observed_ns.front() - time_filtering_threshold_ns_;      // check(point, 0, first - threshold)
    earliest[m + q] = attrs.last_observed_ns.back() + time_filtering_threshold_ns_;  // check(point, last + threshold)
  }
  const std::vector<RayVerificator::CheckResult> res = ray_verificator_->checkMany(both, earliest, latest);
  RayVerificator::CheckResult before, after;
  for (size_t q = 0; q < m; ++q) {
    before.merge(res[q]);
    after.merge(res[m + q]);
  }
  const auto before_result = ray_change_detector_->detectChanges(before, false);
  const auto after_result = ray_change_detector_->detectChanges(after, true);
  change.first_absent = before_result.closest_absent.value_or(0ul);
  change.last_absent = after_result.closest_absent.value_or(0ul);
  change.first_persistent = before_result.furthest_persistent.value_or(0ul);
  change.last_persistent = after_result.furthest_persistent.value_or(0ul);
}

}  // namespace khronos

// ---- C entry points for the bindings / tests (khronos_amd/host_capi.py) ------------------------------------------------------
extern "C" void khr_set_last_error(const char* text);

namespace {
std::shared_ptr<const khronos::RayChangeDetector> makeVote(float temporal_resolution, int64_t window_size, int use_relative_confidence,
                                                           float absence_confidence, float presence_confidence) {
  khronos::RayChangeDetector::Config cfg;
  cfg.temporal_resolution = temporal_resolution;
  cfg.window_size = static_cast<size_t>(window_size);
  cfg.use_relative_confidence = use_relative_confidence != 0;
  cfg.absence_confidence = absence_confidence;
  cfg.presence_confidence = presence_confidence;
  return std::make_shared<khronos::RayChangeDetector>(cfg);
}
}  // namespace

extern "C" {

// RayBackgroundChangeDetector::detectChanges over arrays.  rv: a khr_rayver index filled by the caller (khr_rv_add_rays).
// states_io[n_vertices]: the first n_known entries are the states so far (0 unobserved, 1 persistent, 2 absent: ChangeState);
// on return all n_vertices are set.  Returns the number of re-observed vertices whose state changed, < 0 on error.
int64_t khr_host_background_changes(khr_rayver* rv, int64_t n_vertices, const float* positions, const uint64_t* stamps, int64_t n_known,
                                    const int64_t* reobserved, int64_t n_reobserved, float time_filtering_threshold, float temporal_resolution,
                                    int64_t window_size, int use_relative_confidence, float absence_confidence, float presence_confidence,
                                    uint8_t* states_io) {
  if (!rv || n_vertices < 0 || n_known < 0 || n_known > n_vertices || n_reobserved < 0 || (n_vertices > 0 && (!positions || !stamps || !states_io)) ||
      (n_reobserved > 0 && !reobserved) || window_size < 0)
    return KHR_EINVAL;
  try {
    auto ver = std::make_shared<const khronos::RayVerificator>(khronos::RayVerificator::Config(), rv);
    khronos::RayBackgroundChangeDetector::Config cfg;
    cfg.time_filtering_threshold = time_filtering_threshold;
    const khronos::RayBackgroundChangeDetector det(cfg, ver, makeVote(temporal_resolution, window_size, use_relative_confidence, absence_confidence,
                                                                      presence_confidence));
    khronos::BackgroundChanges changes;
    for (int64_t i = 0; i < n_known; ++i) changes.push_back(static_cast<khronos::ChangeState>(states_io[i]));
    std::vector<size_t> re;
    for (int64_t i = 0; i < n_reobserved; ++i)
      if (reobserved[i] >= 0) re.push_back(static_cast<size_t>(reobserved[i]));
    const size_t updates = det.detectChanges(std::vector<float>(positions, positions + 3 * n_vertices), std::vector<uint64_t>(stamps, stamps + n_vertices),
                                             re, changes);
    for (int64_t i = 0; i < n_vertices; ++i) states_io[i] = static_cast<uint8_t>(changes[static_cast<size_t>(i)]);
    return static_cast<int64_t>(updates);
  } catch (const std::exception& e) {
    khr_set_last_error(e.what());
    return KHR_EINVAL;
  }
}

// RayObjectChangeDetector::checkObjectObservation for one object: vertices in the bounding-box frame + the box (min, max).
// out[4] = first_absent, last_absent, first_persistent, last_persistent.
int khr_host_object_change(khr_rayver* rv, int64_t n_vertices, const float* vertices, const float* bbox_min, const float* bbox_max,
                           uint64_t first_observed, uint64_t last_observed, float time_filtering_threshold, int query_subsampling,
                           float temporal_resolution, int64_t window_size, int use_relative_confidence, float absence_confidence,
                           float presence_confidence, uint64_t* out) {
  if (!rv || n_vertices < 0 || (n_vertices > 0 && !vertices) || !bbox_min || !bbox_max || !out || window_size < 0) return KHR_EINVAL;
  try {
    auto ver = std::make_shared<const khronos::RayVerificator>(khronos::RayVerificator::Config(), rv);
    khronos::RayObjectChangeDetector::Config cfg;
    cfg.time_filtering_threshold = time_filtering_threshold;
    cfg.query_subsampling = query_subsampling;
    const khronos::RayObjectChangeDetector det(cfg, ver, makeVote(temporal_resolution, window_size, use_relative_confidence, absence_confidence,
                                                                  presence_confidence));
    hydra::KhronosObjectAttributes attrs;
    attrs.mesh.points.assign(vertices, vertices + 3 * n_vertices);
    for (int d = 0; d < 3; ++d) {
      attrs.bounding_box.min[d] = bbox_min[d];
      attrs.bounding_box.max[d] = bbox_max[d];
    }
    attrs.bounding_box.valid = true;
    attrs.first_observed_ns = {first_observed};
    attrs.last_observed_ns = {last_observed};
    khronos::ObjectChange change;
    det.checkObjectObservation(attrs, change);
    out[0] = change.first_absent;
    out[1] = change.last_absent;
    out[2] = change.first_persistent;
    out[3] = change.last_persistent;
    return KHR_OK;
  } catch (const std::exception& e) {
    khr_set_last_error(e.what());
    return KHR_EINVAL;
  }
}

}  // extern "C"
