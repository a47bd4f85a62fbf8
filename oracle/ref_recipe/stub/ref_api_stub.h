// ref_api_stub.h -- DECLARATIONS ONLY, for `g++ -fsyntax-only` of oracle/ref_recipe/dump_vectors.cpp in a container that has
// neither Hydra nor Eigen nor OpenCV.  It states the upstream API surface the harness relies on, one line of evidence each
// (file:line under /root/reference where Khronos itself uses the symbol that way).  Nothing here is ever linked or run, and
// nothing here is derived from upstream sources (they are not on this machine): a mismatch with the real headers shows up as a
// compile error when build.sh runs against real checkouts, which is where makeInput() says "ADAPT HERE".
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

// ---- Eigen / OpenCV: only what the harness touches ---------------------------------------------------------------------------
namespace Eigen {
constexpr int RowMajor = 1;
template <typename T, int R, int C, int O = 0> struct Matrix {};
using Matrix4d = Matrix<double, 4, 4>;
template <typename M> struct Map {
  explicit Map(const double*) {}
  operator Matrix4d() const { return {}; }
};
struct Isometry3d {
  Isometry3d() = default;
  explicit Isometry3d(const Matrix4d&) {}
};
struct Vector3f {
  float x() const { return 0; }
  float y() const { return 0; }
  float z() const { return 0; }
};
struct Vector3i {
  int x() const { return 0; }
  int y() const { return 0; }
  int z() const { return 0; }
};
}  // namespace Eigen
constexpr int CV_32FC1 = 5, CV_8UC3 = 16, CV_32SC1 = 4;
namespace cv {
struct Mat {
  Mat() = default;
  Mat(int, int, int, void*) {}
  Mat clone() const { return *this; }
  static Mat zeros(int, int, int) { return {}; }
};
inline int countNonZero(const Mat&) { return 0; }
}  // namespace cv

namespace spatial_hash {
using BlockIndex = Eigen::Vector3i;                 // common_types.h:79-83 (hydra::BlockIndex)
using BlockIndices = std::vector<BlockIndex>;       // tracking_integrator.h: resetInactive(.., spatial_hash::BlockIndices*)
}  // namespace spatial_hash

namespace hydra {
using BlockIndex = spatial_hash::BlockIndex;
struct Color { uint8_t r = 0, g = 0, b = 0, a = 255; };                                  // tsdf_voxel.color (mesh_object_extractor.cpp:259)
struct TsdfVoxel { float distance = 0, weight = 0; Color color; };                       // .distance mesh_object_extractor.cpp:249
struct TrackingVoxel { uint64_t last_observed = 0, last_occupied = 0; bool ever_free = false, active = false, to_remove = false; };  // tracking_integrator.cpp:224-252
struct SemanticVoxel { uint32_t semantic_label = 0; bool empty = true; };                // mesh_object_extractor.cpp:344-355
template <typename V> struct Block {
  using Ptr = std::shared_ptr<Block>;
  BlockIndex index;                                                                     // tsdf_block.index (mesh_object_extractor.cpp:247)
  size_t numVoxels() const { return 0; }                                                // mesh_object_extractor.cpp:248
  V& getVoxel(size_t) { static V v; return v; }                                         // mesh_object_extractor.cpp:249
  const V& getVoxel(size_t) const { static V v; return v; }
  void clearUpdated() const {}                                                          // active_window.cpp:170
};
using TsdfBlock = Block<TsdfVoxel>;
template <typename V> struct Layer {
  std::vector<BlockIndex> allocatedBlockIndices() const { return {}; }                  // tracking_integrator.cpp:75
  typename Block<V>::Ptr getBlockPtr(const BlockIndex&) const { return nullptr; }       // tracking_integrator.cpp:142
  const Block<V>& getBlock(const BlockIndex&) const { static Block<V> b; return b; }    // mesh_object_extractor.cpp:247
  Block<V>* begin() { return nullptr; }                                                 // active_window.cpp:169 (range-for over a layer)
  Block<V>* end() { return nullptr; }
};
struct MeshBlock {};
struct MeshLayer {};
struct VolumetricMap {
  struct Config { float voxel_size = 0.1f; int voxels_per_side = 16; float truncation_distance = 0.3f; bool with_semantics = false, with_tracking = true; };  // mesh_object_extractor.cpp:201-211
  explicit VolumetricMap(const Config&) {}                                              // mesh_object_extractor.cpp:215
  Layer<TsdfVoxel>& getTsdfLayer() { static Layer<TsdfVoxel> l; return l; }             // mesh_object_extractor.cpp:218
  Layer<TrackingVoxel>* getTrackingLayer() { static Layer<TrackingVoxel> l; return &l; } // tracking_integrator.cpp:109
  Layer<SemanticVoxel>* getSemanticLayer() { static Layer<SemanticVoxel> l; return &l; } // mesh_object_extractor.cpp:219
  MeshLayer& getMeshLayer() { static MeshLayer l; return l; }                            // mesh_object_extractor.cpp:269
};
struct Sensor { virtual ~Sensor() = default; };
struct Camera : Sensor {
  struct Config { int width = 0, height = 0; float fx = 0, fy = 0, cx = 0, cy = 0; double min_range = 0, max_range = 0; };  // khronos_ros/config/vio/jackal/LeftCameraParams.yaml
  explicit Camera(const Config&) {}
};
struct InputData {                                                                       // frame_data.h:66; fields: free_space_motion_detector.cpp:169-175
  explicit InputData(std::shared_ptr<const Sensor>) {}
  uint64_t timestamp_ns = 0;
  Eigen::Isometry3d world_T_body;                                                        // active_window.cpp:227-228
  cv::Mat depth_image, color_image, label_image, range_image, vertex_map;
};
namespace conversions {
bool normalizeData(InputData&, bool);     // the steps parseInputPacket runs (active_window.cpp:275); names as recalled from Hydra main
bool convertVertexMap(InputData&, bool);
}  // namespace conversions
void maskNonZero(const cv::Mat&, cv::Mat&);                                              // active_window.cpp:209
struct ProjectiveIntegrator {
  struct Config {};
  explicit ProjectiveIntegrator(const Config&) {}
  void updateMap(const InputData&, VolumetricMap&, bool allocate_blocks, const cv::Mat& mask) const;  // active_window.cpp:210
};
struct MeshIntegratorConfig {};
struct MeshIntegrator {
  explicit MeshIntegrator(const MeshIntegratorConfig&) {}
  void generateMesh(VolumetricMap&, bool only_mesh_updated_blocks, bool clear_updated_flag) const;    // active_window.cpp:223
};
}  // namespace hydra

namespace khronos {
using hydra::VolumetricMap;
struct MeasurementCluster {};
struct FrameData {                                                                       // frame_data.h:59-83
  explicit FrameData(const hydra::InputData& in) : input(in) {}
  const hydra::InputData input;
  std::vector<MeasurementCluster> dynamic_clusters;
  cv::Mat dynamic_image, object_image;
};
struct TrackingIntegrator {                                                              // tracking_integrator.h:59-107
  struct Config { float temporal_buffer = 1.f, burn_in_period = 1.f, tsdf_occupancy_threshold = -1.5f; int neighbor_connectivity = 18; float temporal_window = 3.f; int num_threads = 1; };
  explicit TrackingIntegrator(const Config&) {}
  void updateBlocks(const FrameData&, VolumetricMap&) const;
  void resetInactive(VolumetricMap&, spatial_hash::BlockIndices* removed = nullptr) const;
};
struct FreeSpaceMotionDetector {                                                         // free_space_motion_detector.h:72-118
  struct Config { int neighbor_connectivity = 26, min_cluster_size = 0, max_cluster_size = 1000000; float min_separation_distance = 1.f, max_range = 10000.f, min_z_coordinate = -10000.f; int num_threads = 1; };
  explicit FreeSpaceMotionDetector(const Config&) {}
  void processInput(const VolumetricMap&, FrameData&);
};
struct Color { uint8_t r = 0, g = 0, b = 0, a = 255; };
struct Mesh {                                                                            // geometry_utils.cpp:66-83
  std::vector<Eigen::Vector3f> points;
  std::vector<Color> colors;
  std::vector<uint32_t> labels;
  std::vector<uint64_t> first_seen_stamps, stamps;
};
namespace utils { Mesh combineMeshLayer(const hydra::MeshLayer&); }                      // geometry_utils.cpp:61
}  // namespace khronos
