// ray_verificator.cpp — see ray_verificator.h.  Restated from khronos/src/backend/change_detection/ray_verificator.cpp;
// the ray march and the ray-point tests run on the device (khr_rv_add_rays / khr_rv_check).
#include "ray_verificator.h"

#include <algorithm>
#include <ctime>
#include <stdexcept>

namespace khronos {

namespace {
void chk(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + khr_last_error());
}
}  // namespace

RayVerificator::Config RayVerificator::Config::fromYaml(const khronos_amd::YamlNode& m) {
  Config c;
  m.read("verbosity", c.verbosity);
  m.read("block_size", c.block_size);
  m.read("radial_tolerance", c.radial_tolerance);
  m.read("depth_tolerance", c.depth_tolerance);
  std::string p;
  m.read("ray_policy", p);
  if (!p.empty()) {
    static const char* names[] = {"First", "Last", "FirstAndLast", "Middle", "All", "Random", "Random3"};
    bool found = false;
    for (int i = 0; i < 7; ++i)
      if (p == names[i]) {
        c.ray_policy = static_cast<RayPolicy>(i);
        found = true;
      }
    if (!found) throw std::invalid_argument("ray_policy must be one of First, Last, FirstAndLast, Middle, All, Random, Random3");
  }
  m.read("active_window_duration", c.active_window_duration);
  return c;
}

RayVerificator::RayVerificator(const Config& cfg) : config(cfg), seed_(static_cast<unsigned int>(time(nullptr))) {
  if (!(config.block_size > 0.f)) throw std::invalid_argument("block_size must be > 0");
  if (!(config.radial_tolerance > 0.f)) throw std::invalid_argument("radial_tolerance must be > 0");
  if (!(config.depth_tolerance > 0.f)) throw std::invalid_argument("depth_tolerance must be > 0");
  chk(khr_rv_create(config.block_size, config.radial_tolerance, config.depth_tolerance, config.device, &rv_), "khr_rv_create");
}

RayVerificator::RayVerificator(const Config& cfg, khr_rayver* borrowed)
    : config(cfg), rv_(borrowed), owns_(false), seed_(static_cast<unsigned int>(time(nullptr))) {
  if (!borrowed) throw std::invalid_argument("null ray index");
}

RayVerificator::~RayVerificator() {
  if (rv_ && owns_) khr_rv_destroy(rv_);
}

void RayVerificator::clear() {
  chk(khr_rv_clear(rv_), "khr_rv_clear");
  timestamps_.clear();
  positions_.clear();
  previous_vertex_index_ = 0;
}

size_t RayVerificator::numRays() const { return static_cast<size_t>(khr_rv_num_rays(rv_)); }

std::unordered_set<size_t> RayVerificator::computeVertexSources(uint64_t first_seen, uint64_t last_seen) {
  std::unordered_set<size_t> result;
  using P = Config::RayPolicy;
  const auto b = timestamps_.begin(), e = timestamps_.end();
  if (config.ray_policy == P::kFirst || config.ray_policy == P::kFirstAndLast) {
    const auto it = std::upper_bound(b, e, first_seen);
    if (it != e) result.insert(static_cast<size_t>(it - b));
  }
  if (config.ray_policy == P::kLast || config.ray_policy == P::kFirstAndLast) {
    const auto it = std::lower_bound(b, e, last_seen);
    if (it != e) result.insert(static_cast<size_t>(it - b));
  }
  if (config.ray_policy == P::kMiddle) {
    const uint64_t stamp = (last_seen + first_seen) / 2;
    const auto it = std::lower_bound(b, e, stamp);
    if (it != e) result.insert(static_cast<size_t>(it - b));
  }
  if (config.ray_policy == P::kAll) {
    const auto lo = std::upper_bound(b, e, first_seen), hi = std::lower_bound(b, e, last_seen);
    for (auto it = lo; it < hi; ++it) result.insert(static_cast<size_t>(it - b));
  }
  if (config.ray_policy == P::kRandom || config.ray_policy == P::kRandom3) {
    const auto lo = std::upper_bound(b, e, first_seen), hi = std::lower_bound(b, e, last_seen);
    const size_t range = lo < hi ? static_cast<size_t>(hi - lo) : 0, start = static_cast<size_t>(lo - b);
    if (range > 0)
      for (int i = 0; i < (config.ray_policy == P::kRandom3 ? 3 : 1); ++i) {
        const size_t index = start + static_cast<size_t>(rand_r(&seed_)) % range;
        if (index < timestamps_.size()) result.insert(index);
      }
  }
  return result;
}

void RayVerificator::updateData(const std::vector<uint64_t>& pose_stamps, const std::vector<float>& pose_positions,
                                const std::vector<float>& vertices, const std::vector<uint64_t>& first_seen,
                                const std::vector<uint64_t>& last_seen_in) {
  if (pose_positions.size() != 3 * pose_stamps.size()) throw std::invalid_argument("pose arrays have inconsistent sizes");
  // addPoseNodes (:178-209): new sensor poses extend the sorted stamp list
  for (size_t i = timestamps_.size(); i < pose_stamps.size(); ++i) {
    timestamps_.push_back(pose_stamps[i]);
    positions_.insert(positions_.end(), pose_positions.begin() + 3 * i, pose_positions.begin() + 3 * i + 3);
  }
  // addVertices (:211-264)
  const size_t nv = first_seen.size();
  if (vertices.size() != 3 * nv || last_seen_in.size() != nv) return;  // "Mesh arrays have inconsistent sizes"
  if (nv == 0) return;
  const uint64_t offset_ns = config.active_window_duration > 0 ? static_cast<uint64_t>(config.active_window_duration * 1e9) : 0;
  std::vector<uint64_t> stamps;
  std::vector<float> sources, targets;
  for (size_t i = previous_vertex_index_; i < nv; ++i) {
    std::unordered_set<size_t> src = computeVertexSources(first_seen[i], last_seen_in[i] - offset_ns);
    std::vector<size_t> ordered(src.begin(), src.end());
    std::sort(ordered.begin(), ordered.end());  // the reference iterates the unordered set; ascending pose index here
    for (size_t s : ordered) {
      if (s >= timestamps_.size()) continue;
      stamps.push_back(timestamps_[s]);
      sources.insert(sources.end(), positions_.begin() + 3 * s, positions_.begin() + 3 * s + 3);
      targets.insert(targets.end(), vertices.begin() + 3 * i, vertices.begin() + 3 * i + 3);
    }
  }
  previous_vertex_index_ = nv;
  if (!stamps.empty())
    chk(khr_rv_add_rays(rv_, static_cast<int64_t>(stamps.size()), stamps.data(), sources.data(), targets.data()), "khr_rv_add_rays");
}

std::vector<RayVerificator::CheckResult> RayVerificator::checkMany(const std::vector<float>& points, const std::vector<uint64_t>& earliest,
                                                                   const std::vector<uint64_t>& latest) const {
  const size_t m = earliest.size();
  if (points.size() != 3 * m || latest.size() != m) throw std::invalid_argument("query arrays have inconsistent sizes");
  std::vector<CheckResult> out(m);
  if (m == 0) return out;
  std::vector<uint32_t> np(m), na(m);
  uint64_t tp = 0, ta = 0;
  chk(khr_rv_check(rv_, static_cast<int64_t>(m), points.data(), earliest.data(), latest.data(), np.data(), na.data(), &tp, &ta),
      "khr_rv_check");
  std::vector<uint64_t> pres(std::max<uint64_t>(tp, 1)), absn(std::max<uint64_t>(ta, 1));
  chk(khr_rv_check_stamps(rv_, pres.data(), absn.data()), "khr_rv_check_stamps");
  size_t op = 0, oa = 0;
  for (size_t i = 0; i < m; ++i) {
    out[i].present.assign(pres.begin() + op, pres.begin() + op + np[i]);
    out[i].absent.assign(absn.begin() + oa, absn.begin() + oa + na[i]);
    op += np[i];
    oa += na[i];
  }
  return out;
}

RayVerificator::CheckResult RayVerificator::check(const float* point, uint64_t earliest, uint64_t latest) const {
  return checkMany({point[0], point[1], point[2]}, {earliest}, {latest})[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// RayChangeDetector (ray_change_detector.cpp)
// ---------------------------------------------------------------------------------------------------------------------
RayChangeDetector::Config RayChangeDetector::Config::fromYaml(const khronos_amd::YamlNode& m) {
  Config c;
  m.read("verbosity", c.verbosity);
  m.read("temporal_resolution", c.temporal_resolution);
  int w = static_cast<int>(c.window_size);
  m.read("window_size", w);
  if (w < 0) throw std::invalid_argument("window_size must be > 0");
  c.window_size = static_cast<size_t>(w);
  m.read("use_relative_confidence", c.use_relative_confidence);
  m.read("absence_confidence", c.absence_confidence);
  m.read("presence_confidence", c.presence_confidence);
  return c;
}

void RayChangeDetector::Config::checkValid() const {  // ray_change_detector.cpp:51-60
  if (!(temporal_resolution > 0.f)) throw std::invalid_argument("RayChangeDetector: temporal_resolution must be > 0");
  if (!(window_size > 0)) throw std::invalid_argument("RayChangeDetector: window_size must be > 0");
  if (use_relative_confidence) {
    if (!(absence_confidence >= 0.f && absence_confidence <= 1.f)) throw std::invalid_argument("RayChangeDetector: absence_confidence must be in [0, 1]");
    if (!(presence_confidence >= 0.f && presence_confidence <= 1.f)) throw std::invalid_argument("RayChangeDetector: presence_confidence must be in [0, 1]");
  } else {
    if (!(absence_confidence > 0.f)) throw std::invalid_argument("RayChangeDetector: absence_confidence must be > 0");
    if (!(presence_confidence > 0.f)) throw std::invalid_argument("RayChangeDetector: presence_confidence must be > 0");
  }
}

static const RayChangeDetector::Config& validated(const RayChangeDetector::Config& c) {
  c.checkValid();
  return c;
}

// resolution_ns_(config.temporal_resolution * 1e9): float * double -> double -> uint64_t (ray_change_detector.cpp:63-64)
RayChangeDetector::RayChangeDetector(const Config& cfg)
    : config(validated(cfg)), resolution_ns_(static_cast<uint64_t>(static_cast<double>(cfg.temporal_resolution) * 1e9)) {}

RayChangeDetector::ChangeResult RayChangeDetector::detectChanges(const RayVerificator::CheckResult& check, bool forward) const {
  return detectChanges(check.present.data(), check.present.size(), check.absent.data(), check.absent.size(), forward);
}

RayChangeDetector::ChangeResult RayChangeDetector::detectChanges(const uint64_t* present, size_t n_present, const uint64_t* absent,
                                                                 size_t n_absent, bool forward) const {
  // series[time_index] = {num_present, num_absent} (:72-81)
  std::unordered_map<size_t, std::pair<unsigned, unsigned>> series;
  for (size_t i = 0; i < n_present; ++i) series[present[i] / resolution_ns_].first++;
  for (size_t i = 0; i < n_absent; ++i) series[absent[i] / resolution_ns_].second++;
  // directional list of the occupied bins (:83-93)
  std::vector<size_t> idx;
  idx.reserve(series.size());
  for (const auto& kv : series) idx.push_back(kv.first);
  if (forward) std::sort(idx.begin(), idx.end());
  else std::sort(idx.begin(), idx.end(), std::greater<size_t>());
  ChangeResult result;
  for (const size_t ti : idx) {
    // the window always extends towards LATER bins, in both directions (:101-107)
    unsigned np = 0, na = 0;
    for (size_t i = 0; i < config.window_size; ++i) {
      const auto it = series.find(ti + i);
      if (it != series.end()) {
        np += it->second.first;
        na += it->second.second;
      }
    }
    if (config.use_relative_confidence) {  // (:110-119)
      const float absence = na / static_cast<float>(np + na);
      if (absence > config.absence_confidence) {
        result.closest_absent = ti * resolution_ns_;
        return result;  // once an absence is found the search stops
      }
      if (1.f - absence > config.presence_confidence) result.furthest_persistent = ti * resolution_ns_;
    } else {  // (:120-129) counts against float thresholds
      if (na > config.absence_confidence) {
        result.closest_absent = ti * resolution_ns_;
        return result;
      }
      if (np > config.presence_confidence) result.furthest_persistent = ti * resolution_ns_;
    }
  }
  return result;
}

std::vector<RayChangeDetector::ChangeResult> RayChangeDetector::detectChangesMany(const RayVerificator& verificator,
                                                                                 const std::vector<float>& points,
                                                                                 const std::vector<uint64_t>& earliest,
                                                                                 const std::vector<uint64_t>& latest,
                                                                                 const std::vector<uint8_t>& forward) const {
  const size_t m = earliest.size();
  if (points.size() != 3 * m || latest.size() != m || forward.size() != m) throw std::invalid_argument("query arrays have inconsistent sizes");
  std::vector<ChangeResult> out(m);
  if (m == 0) return out;
  uint64_t tp = 0, ta = 0;
  chk(khr_rv_check(verificator.handle(), static_cast<int64_t>(m), points.data(), earliest.data(), latest.data(), nullptr, nullptr, &tp, &ta),
      "khr_rv_check");
  std::vector<uint64_t> ca(m), fp(m);
  std::vector<uint8_t> fl(m);
  chk(khr_rv_detect_changes(verificator.handle(), config.temporal_resolution, static_cast<int64_t>(config.window_size),
                            config.use_relative_confidence ? 1 : 0, config.absence_confidence, config.presence_confidence, forward.data(), 0,
                            ca.data(), fp.data(), fl.data()),
      "khr_rv_detect_changes");
  for (size_t i = 0; i < m; ++i) {
    if (fl[i] & 0x80) {  // too many time bins for the device histogram: vote here on the point's own lists
      out[i] = detectChanges(verificator.check(&points[3 * i], earliest[i], latest[i]), forward[i] != 0);
      continue;
    }
    if (fl[i] & 1) out[i].closest_absent = ca[i];
    if (fl[i] & 2) out[i].furthest_persistent = fp[i];
  }
  return out;
}

}  // namespace khronos
