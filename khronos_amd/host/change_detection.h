// change_detection.h — host-side mirrors of the two callers of the ray verificator in the Khronos backend (SURVEY.md section 8 f4,
// BASELINE configs[4] "full spatio-temporal change reconciliation"):
//   khronos::RayBackgroundChangeDetector  khronos/src/backend/change_detection/background/ray_background_change_detector.cpp:59-103
//   khronos::RayObjectChangeDetector      khronos/src/backend/change_detection/objects/ray_object_change_detector.cpp:62-160
// The reference walks the scene graph and asks RayVerificator::check + RayChangeDetector::detectChanges once per mesh vertex
// (background) or twice per sub-sampled vertex (objects); here every batch of queries is ONE device pass over the ray index
// (khr_rv_check, and khr_rv_detect_changes for the per-vertex votes).  spark_dsg is not available offline, so the mesh / the
// objects arrive as arrays and as KhronosObjectAttributes (hydra_compat.h) instead of a DynamicSceneGraph.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "hydra_compat.h"
#include "ray_verificator.h"

namespace khronos {

using NodeId = uint64_t;
enum class ChangeState { kUnobserved, kPersistent, kAbsent };  // change_state.h:124
using BackgroundChanges = std::vector<ChangeState>;            // change_state.h:129: one entry per background-mesh vertex

struct RPGOMerge {  // change_state.h:54-62
  NodeId from_node = 0, to_node = 0;
  bool is_valid = false;
};
using RPGOMerges = std::vector<RPGOMerge>;

struct ObjectChange {  // change_state.h:76-103
  NodeId node_id = 0;
  NodeId merged_id = 0;
  uint64_t first_absent = 0, last_absent = 0, first_persistent = 0, last_persistent = 0;
};
struct ObjectChanges : std::vector<ObjectChange> {
  std::vector<ObjectChange>::iterator find(NodeId id);
};

class RayBackgroundChangeDetector {
 public:
  struct Config {  // ray_background_change_detector.h:57-62
    int verbosity = 0;
    float time_filtering_threshold = 5.f;  // [s]
  } const config;

  RayBackgroundChangeDetector(const Config& config, std::shared_ptr<const RayVerificator> ray_verificator,
                              std::shared_ptr<const RayChangeDetector> ray_change_detector);

  // detectChanges (:59-88): `changes` holds the states of the vertices seen so far; vertices [changes.size(), n) are new and
  // get their first state, the re-observed ones (RayVerificator::getReobservedVertices) are recomputed.  Returns how many
  // re-observed vertices changed state.  vertex_positions: 3 per vertex; vertex_stamps: mesh.timestamp(i).
  size_t detectChanges(const std::vector<float>& vertex_positions, const std::vector<uint64_t>& vertex_stamps,
                       const std::vector<size_t>& reobserved_vertices, BackgroundChanges& changes) const;
  // checkVertex (:90-103), one vertex (a batch of one)
  ChangeState checkVertex(const float* position, uint64_t stamp) const;

 private:
  std::shared_ptr<const RayVerificator> ray_verificator_;
  std::shared_ptr<const RayChangeDetector> ray_change_detector_;
  const uint64_t time_filtering_threshold_ns_;
};

class RayObjectChangeDetector {
 public:
  struct Config {  // ray_object_change_detector.h:57-64
    float time_filtering_threshold = 5.f;  // [s]
    int query_subsampling = 100;
  } const config;
  struct Object {  // an object node of the DsgLayers::OBJECTS layer
    NodeId node_id = 0;
    const hydra::KhronosObjectAttributes* attributes = nullptr;
  };

  RayObjectChangeDetector(const Config& config, std::shared_ptr<const RayVerificator> ray_verificator,
                          std::shared_ptr<const RayChangeDetector> ray_change_detector);

  // detectChanges (:62-102): the states of re-observed objects are dropped and recomputed, objects that already have a state are
  // left alone, dynamic objects (with a trajectory) are skipped
  void detectChanges(const std::vector<Object>& objects, const std::vector<NodeId>& reobserved_objects, const RPGOMerges& rpgo_merges,
                     ObjectChanges& changes) const;
  void checkObjectMerge(const RPGOMerges& rpgo_merges, ObjectChange& change) const;                 // :104-115
  void checkObjectObservation(const hydra::KhronosObjectAttributes& attrs, ObjectChange& change) const;   // :117-160

 private:
  std::shared_ptr<const RayVerificator> ray_verificator_;
  std::shared_ptr<const RayChangeDetector> ray_change_detector_;
  const uint64_t time_filtering_threshold_ns_;
};

}  // namespace khronos
