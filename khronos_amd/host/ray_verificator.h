// ray_verificator.h — host-side mirror of khronos::RayVerificator
// (khronos/include/khronos/backend/change_detection/ray_verificator.h:60-240) over the device index of
// include/khronos_amd.h (khr_rv_*).  The reference reads the agent poses and the background mesh out of the scene
// graph (setDsg / updateDsg); spark_dsg is not available here, so the same data arrives as arrays (updateData).
#pragma once
#include <cstdint>
#include <limits>
#include <memory>
#include <optional>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/khronos_amd.h"
#include "mini_yaml.h"

namespace khronos {

class RayVerificator {
 public:
  struct Config {  // ray_verificator.h:67-98, checks ray_verificator.cpp:59-61
    int verbosity = 0;
    float block_size = 1.f;
    float radial_tolerance = 0.1f;
    float depth_tolerance = 0.1f;
    enum class RayPolicy { kFirst, kLast, kFirstAndLast, kMiddle, kAll, kRandom, kRandom3 } ray_policy = RayPolicy::kMiddle;
    float active_window_duration = 0.f;
    int device = 0;  // extension: HIP device of the index
    static Config fromYaml(const khronos_amd::YamlNode& node);
  } const config;

  struct CheckResult {  // ray_verificator.h:104-112
    std::vector<uint64_t> absent, present;
    void merge(const CheckResult& other) {
      absent.insert(absent.end(), other.absent.begin(), other.absent.end());
      present.insert(present.end(), other.present.begin(), other.present.end());
    }
  };

  explicit RayVerificator(const Config& config);
  // a view of a device index somebody else owns (bindings that hold a khr_rayver of their own): never destroys it
  RayVerificator(const Config& config, khr_rayver* borrowed);
  ~RayVerificator();
  RayVerificator(const RayVerificator&) = delete;
  RayVerificator& operator=(const RayVerificator&) = delete;

  // setDsg(): forget everything; updateDsg() / addPoseNodes() + addVertices(): the caller passes ALL agent poses
  // (sorted by time, as the agent layer is) and ALL mesh vertices so far; only the new ones are processed.
  void clear();
  void updateData(const std::vector<uint64_t>& pose_stamps, const std::vector<float>& pose_positions /*3 per pose*/,
                  const std::vector<float>& vertices /*3 per vertex*/, const std::vector<uint64_t>& first_seen,
                  const std::vector<uint64_t>& last_seen);

  // check() (ray_verificator.cpp:66-145); checkMany answers many points with one device pass
  CheckResult check(const float* point, uint64_t earliest = 0ul, uint64_t latest = std::numeric_limits<uint64_t>::max()) const;
  std::vector<CheckResult> checkMany(const std::vector<float>& points, const std::vector<uint64_t>& earliest,
                                     const std::vector<uint64_t>& latest) const;

  size_t numRays() const;
  khr_rayver* handle() const { return rv_; }  // the device index (for RayChangeDetector::detectChangesMany)
  // computeVertexSources (ray_verificator.cpp:266-325): indices into the pose list
  std::unordered_set<size_t> computeVertexSources(uint64_t first_seen, uint64_t last_seen);

 private:
  khr_rayver* rv_ = nullptr;
  bool owns_ = true;
  std::vector<uint64_t> timestamps_;
  std::vector<float> positions_;
  size_t previous_vertex_index_ = 0;
  unsigned int seed_;
};

// Host-side mirror of khronos::RayChangeDetector (khronos/include/khronos/backend/change_detection/ray_change_detector.h:
// 64-115, khronos/src/backend/change_detection/ray_change_detector.cpp:40-133): the time-bin majority vote over the
// presence / absence observations RayVerificator::check returns for one point.  Pure host logic; the callers are
// ray_background_change_detector.cpp:92-103 and ray_object_change_detector.cpp:127-160.
class RayChangeDetector {
 public:
  struct Config {  // ray_change_detector.h:68-87, checks ray_change_detector.cpp:51-60
    int verbosity = 0;
    float temporal_resolution = 1.f;  // [s]
    size_t window_size = 5;
    bool use_relative_confidence = true;
    float absence_confidence = 0.5f;
    float presence_confidence = 0.5f;
    static Config fromYaml(const khronos_amd::YamlNode& node);
    void checkValid() const;
  } const config;

  struct ChangeResult {  // ray_change_detector.h:94-100
    std::optional<uint64_t> closest_absent;
    std::optional<uint64_t> furthest_persistent;
  };

  explicit RayChangeDetector(const Config& config);
  virtual ~RayChangeDetector() = default;

  // detectChanges (ray_change_detector.cpp:66-133)
  ChangeResult detectChanges(const RayVerificator::CheckResult& check, bool forward) const;
  ChangeResult detectChanges(const uint64_t* present, size_t n_present, const uint64_t* absent, size_t n_absent, bool forward) const;

  // check + detectChanges for many points in one device pass (khr_rv_check + khr_rv_detect_changes): what the loops of
  // ray_background_change_detector.cpp:92-103 / ray_object_change_detector.cpp:127-160 do point by point.  forward[i] != 0:
  // search towards the future.  Points whose observations span more time bins than the device histogram holds are
  // voted on here from their stamp lists.
  std::vector<ChangeResult> detectChangesMany(const RayVerificator& verificator, const std::vector<float>& points,
                                              const std::vector<uint64_t>& earliest, const std::vector<uint64_t>& latest,
                                              const std::vector<uint8_t>& forward) const;

 protected:
  const uint64_t resolution_ns_;
};

}  // namespace khronos
