// dump_vectors.cpp -- REFERENCE VECTOR GENERATOR (test infrastructure; cannot be built in the offline container).
//
// Pins the CPU oracle of this repository (oracle/oracle.cpp) against UPSTREAM code: the seeded frame sequence of
// tests/golden/make_golden.py is pushed through the real
//     hydra::ProjectiveIntegrator::updateMap      (called at khronos/src/active_window/active_window.cpp:210)
//     khronos::FreeSpaceMotionDetector            (khronos/src/active_window/motion_detection/free_space_motion_detector.cpp:73-399)
//     khronos::TrackingIntegrator::updateBlocks / resetInactive  (khronos/src/active_window/integration/tracking_integrator.cpp:71-131)
//     hydra::MeshIntegrator::generateMesh         (called at active_window.cpp:223)
// in the order of khronos::ActiveWindow::spinOnce / extractOutputData (active_window.cpp:118-174, 217-249), and the map, the
// per-frame dynamic-pixel counts, the archived block counts and the mesh are written as .npy files that
// oracle/ref_recipe/build.sh zips into oracle/_ref/ref_small.npz -- the file tests/test_golden.py accepts IN PLACE of the
// oracle-generated tests/golden/aw_small.npz (same keys).  With it the parity chain reads HIP == oracle == Hydra; without it
// the claim stays "HIP == our restatement" (parity unpinned).
//
// Needs checkouts of Hydra, Spatial-Hash, config_utilities, Spark-DSG (install/https.rosinstall:1-8,29-36 of the reference) and
// their own dependencies (Eigen, OpenCV, glog, ...): see build.sh.  Every upstream symbol used below is one the reference
// itself uses; the line that shows it is cited.  The ONE place that depends on Hydra types the reference never constructs
// (hydra::InputData from raw images) is makeInput() -- adapt it there if Hydra's main branch has moved.
//
// Compile check without the checkouts (what the offline test does): g++ -fsyntax-only -Ioracle/ref_recipe/stub ... against
// stand-in headers that declare exactly this API surface (oracle/ref_recipe/stub/README.md).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include <hydra/reconstruction/mesh_integrator.h>       // active_window.h:52
#include <hydra/reconstruction/projective_integrator.h> // active_window.h:53
#include <hydra/reconstruction/volumetric_map.h>        // VolumetricMap (mesh_object_extractor.cpp:201-213)
#include <hydra/input/input_data.h>                     // hydra::InputData (frame_data.h:66)
#include <hydra/input/camera.h>                         // hydra::Camera: the pinhole sensor model behind InputData::getSensor()

#include "khronos/active_window/data/frame_data.h"
#include "khronos/active_window/integration/tracking_integrator.h"
#include "khronos/active_window/motion_detection/free_space_motion_detector.h"
#include "khronos/utils/geometry_utils.h"

// the synthetic stream of this repository (khronos_amd/synth/synth.cpp, compiled into the harness by build.sh)
extern "C" {
void* synth_create(uint32_t seed, int num_static, int with_mover);
void synth_destroy(void* scene);
void synth_render(void* scene, int W, int H, float fx, float fy, float cx, float cy, const double* T, double t_sec, float max_depth,
                  float noise_sigma_rel, uint32_t noise_seed, float* depth, uint8_t* rgb, int32_t* label, int num_threads);
}

namespace {

// ---- tests/golden/make_golden.py: W, H, N, CFG ------------------------------------------------------------------------------
constexpr int kW = 96, kH = 72, kN = 16;
constexpr float kVoxelSize = 0.2f, kTruncation = 0.4f, kTemporalWindow = 0.9f;
constexpr int kMinClusterSize = 5;
constexpr float kMinSeparation = 2.0f, kMaxRange = 5.0f;
constexpr uint32_t kSeed = 1234;

// khronos_amd/synth.py: circle_pose / camera_pose (optical frame: x right, y down, z forward; world z up)
void circlePose(double t_sec, double* T /* row-major 4x4 world_T_sensor */) {
  const double th = 2.0 * M_PI * t_sec / 10.0, yaw = th + M_PI / 2;
  const double f[3] = {std::cos(yaw), std::sin(yaw), 0.0}, r[3] = {f[1], -f[0], 0.0}, d[3] = {0.0, 0.0, -1.0};
  const double p[3] = {1.5 * std::cos(th), 1.5 * std::sin(th), 1.5};
  for (int i = 0; i < 3; ++i) {
    T[4 * i + 0] = r[i];
    T[4 * i + 1] = d[i];
    T[4 * i + 2] = f[i];
    T[4 * i + 3] = p[i];
  }
  T[12] = T[13] = T[14] = 0.0;
  T[15] = 1.0;
}

// ---- minimal .npy writer (version 1.0, C order, little endian) ----------------------------------------------------------------
template <typename T>
const char* npyType();
template <> const char* npyType<float>() { return "<f4"; }
template <> const char* npyType<double>() { return "<f8"; }
template <> const char* npyType<uint8_t>() { return "|u1"; }
template <> const char* npyType<int32_t>() { return "<i4"; }
template <> const char* npyType<int64_t>() { return "<i8"; }
template <> const char* npyType<uint64_t>() { return "<u8"; }
template <typename T>
void writeNpy(const std::string& dir, const std::string& name, const std::vector<T>& data, const std::vector<size_t>& shape) {
  std::string sh = "(";
  for (size_t i = 0; i < shape.size(); ++i) sh += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
  sh += ")";
  std::string hdr = std::string("{'descr': '") + npyType<T>() + "', 'fortran_order': False, 'shape': " + sh + ", }";
  while ((10 + hdr.size() + 1) % 64 != 0) hdr += ' ';
  hdr += '\n';
  std::ofstream f(dir + "/" + name + ".npy", std::ios::binary);
  const uint16_t n = static_cast<uint16_t>(hdr.size());
  f.write("\x93NUMPY\x01\x00", 8);
  f.write(reinterpret_cast<const char*>(&n), 2);
  f.write(hdr.data(), static_cast<std::streamsize>(hdr.size()));
  f.write(reinterpret_cast<const char*>(data.data()), static_cast<std::streamsize>(data.size() * sizeof(T)));
}

// ---- ADAPT HERE: raw images + pose -> hydra::InputData -------------------------------------------------------------------------
// The reference only ever receives InputData from hydra::conversions::parseInputPacket (active_window.cpp:275); what it READS
// is: depth_image CV_32FC1, range_image CV_32FC1, vertex_map CV_32FC3 in the WORLD frame (free_space_motion_detector.cpp:169-175),
// label_image CV_32SC1 (instance_forwarding.cpp:83), color_image CV_8UC3, timestamp_ns, world_T_body, getSensor() /
// getSensorPose() (free_space_motion_detector.cpp:80).  Written against Hydra main as of 2025-12; the conversion of depth to
// range / vertices is the upstream one (hydra::conversions), NOT re-implemented here -- that is the point of the exercise.
std::shared_ptr<hydra::InputData> makeInput(const std::shared_ptr<const hydra::Sensor>& camera, uint64_t stamp_ns, const double* world_T_sensor,
                                            const std::vector<float>& depth, const std::vector<uint8_t>& rgb, const std::vector<int32_t>& label) {
  auto data = std::make_shared<hydra::InputData>(camera);
  data->timestamp_ns = stamp_ns;
  Eigen::Matrix4d T = Eigen::Map<const Eigen::Matrix<double, 4, 4, Eigen::RowMajor>>(world_T_sensor);
  data->world_T_body = Eigen::Isometry3d(T);  // body == sensor: the camera's body_T_sensor is the identity (build.sh: camera config)
  data->depth_image = cv::Mat(kH, kW, CV_32FC1, const_cast<float*>(depth.data())).clone();
  data->color_image = cv::Mat(kH, kW, CV_8UC3, const_cast<uint8_t*>(rgb.data())).clone();
  data->label_image = cv::Mat(kH, kW, CV_32SC1, const_cast<int32_t*>(label.data())).clone();
  // range image + world-frame vertex map: the upstream conversion parseInputPacket applies (input_conversion.h)
  if (!hydra::conversions::normalizeData(*data, /*normalize_labels=*/false) || !hydra::conversions::convertVertexMap(*data, /*in_world_frame=*/true))
    return nullptr;
  return data;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string out_dir = argc > 1 ? argv[1] : "oracle/_ref/npy";
  // sensor: pinhole, fx = fy = W / 2, principal point at the centre, range 0.1 .. 5 m (khronos_amd/synth.py; launch/khronos.launch.yaml:10-11)
  hydra::Camera::Config cam;
  cam.width = kW;
  cam.height = kH;
  cam.fx = cam.fy = kW / 2.0f;
  cam.cx = kW / 2.0f;
  cam.cy = kH / 2.0f;
  cam.min_range = 0.1;
  cam.max_range = 5.0;
  const std::shared_ptr<const hydra::Sensor> camera = std::make_shared<hydra::Camera>(cam);

  // map + integrators configured as tests/golden/make_golden.py configures the oracle (defaults elsewhere: Appendix B of SURVEY.md)
  hydra::VolumetricMap::Config map_config;  // fields: mesh_object_extractor.cpp:201-211
  map_config.voxel_size = kVoxelSize;
  map_config.voxels_per_side = 16;
  map_config.truncation_distance = kTruncation;
  map_config.with_semantics = true;
  map_config.with_tracking = true;
  hydra::VolumetricMap map(map_config);
  hydra::ProjectiveIntegrator integrator{hydra::ProjectiveIntegrator::Config()};
  hydra::MeshIntegrator mesher{hydra::MeshIntegratorConfig()};
  khronos::TrackingIntegrator::Config tc;  // tracking_integrator.h:59-83
  tc.temporal_window = kTemporalWindow;
  khronos::TrackingIntegrator tracking(tc);
  khronos::FreeSpaceMotionDetector::Config mc;  // free_space_motion_detector.h:72-97
  mc.min_cluster_size = kMinClusterSize;
  mc.min_separation_distance = kMinSeparation;
  mc.max_range = kMaxRange;
  khronos::FreeSpaceMotionDetector motion(mc);

  void* scene = synth_create(kSeed, 12, 1);
  std::vector<int64_t> n_clusters, dyn_pixels, removed_counts;
  std::vector<uint64_t> stamps;
  std::vector<double> poses, depth_sums;
  for (int i = 0; i < kN; ++i) {
    double T[16];
    circlePose(0.1 * i, T);
    std::vector<float> depth(kW * kH);
    std::vector<uint8_t> rgb(3 * kW * kH);
    std::vector<int32_t> label(kW * kH);
    synth_render(scene, kW, kH, kW / 2.f, kW / 2.f, kW / 2.f, kH / 2.f, T, 0.1 * i, 5.0f, 0.f, kSeed + 7919u * i, depth.data(), rgb.data(),
                 label.data(), 1);
    const uint64_t stamp = static_cast<uint64_t>(std::llround((1.0 + 0.1 * i) * 1e9));
    stamps.push_back(stamp);
    poses.insert(poses.end(), T, T + 16);
    double s = 0;
    for (float d : depth) s += d;
    depth_sums.push_back(s);
    const auto input = makeInput(camera, stamp, T, depth, rgb, label);
    if (!input) {
      std::fprintf(stderr, "frame %d: input conversion failed\n", i);
      return 2;
    }
    // ActiveWindow::createData (active_window.cpp:268-286)
    khronos::FrameData data(*input);
    data.dynamic_image = cv::Mat::zeros(kH, kW, CV_32SC1);
    data.object_image = cv::Mat::zeros(kH, kW, CV_32SC1);
    // spinOnce order (active_window.cpp:127-145): motion detection, then the map update with the dynamic mask
    motion.processInput(map, data);
    n_clusters.push_back(static_cast<int64_t>(data.dynamic_clusters.size()));
    dyn_pixels.push_back(cv::countNonZero(data.dynamic_image));
    cv::Mat mask;
    hydra::maskNonZero(data.dynamic_image, mask);  // active_window.cpp:209
    integrator.updateMap(data.input, map, true, mask);  // active_window.cpp:210
    tracking.updateBlocks(data, map);                   // active_window.cpp:214
    if (i % 5 == 4) {  // extractOutputData (active_window.cpp:217-237) + clearing of the updated flags (:169-171)
      mesher.generateMesh(map, true, true);
      spatial_hash::BlockIndices removed;
      tracking.resetInactive(map, &removed);
      removed_counts.push_back(static_cast<int64_t>(removed.size()));
      for (auto& block : map.getTsdfLayer()) block.clearUpdated();  // (const in the reference's loop: the flag is mutable upstream)
    }
  }
  synth_destroy(scene);

  // ---- the map, block by block in sorted index order (the keys of tests/golden/aw_small.npz) --------------------------------
  auto indices = map.getTsdfLayer().allocatedBlockIndices();  // tracking_integrator.cpp:75
  std::sort(indices.begin(), indices.end(), [](const auto& a, const auto& b) {
    return a.x() != b.x() ? a.x() < b.x() : (a.y() != b.y() ? a.y() < b.y() : a.z() < b.z());
  });
  const size_t nb = indices.size(), nv = 4096;
  std::vector<int32_t> idx;
  std::vector<float> distance, weight;
  std::vector<uint8_t> flags, sem_label, color;
  std::vector<uint64_t> last_observed;
  for (const auto& bi : indices) {
    idx.insert(idx.end(), {bi.x(), bi.y(), bi.z()});
    const auto tsdf = map.getTsdfLayer().getBlockPtr(bi);            // tracking_integrator.cpp:142
    const auto trk = map.getTrackingLayer()->getBlockPtr(bi);        // tracking_integrator.cpp:147
    const auto& sem = map.getSemanticLayer()->getBlock(bi);          // mesh_object_extractor.cpp:219,247
    for (size_t v = 0; v < nv; ++v) {
      const auto& t = tsdf->getVoxel(v);
      const auto& k = trk->getVoxel(v);
      const auto& s = sem.getVoxel(v);
      distance.push_back(t.distance);
      weight.push_back(t.weight);
      color.insert(color.end(), {t.color.r, t.color.g, t.color.b, t.color.a});
      last_observed.push_back(k.last_observed);
      // bit 0 active, 1 ever_free, 2 to_remove (tracking_integrator.cpp:224-246), 3 semantic entry non-empty (mesh_object_extractor.cpp:344)
      flags.push_back(static_cast<uint8_t>((k.active ? 1 : 0) | (k.ever_free ? 2 : 0) | (k.to_remove ? 4 : 0) | (s.empty ? 0 : 8)));
      sem_label.push_back(static_cast<uint8_t>(s.semantic_label));
    }
  }
  const khronos::Mesh mesh = khronos::utils::combineMeshLayer(map.getMeshLayer());  // geometry_utils.cpp:61-86
  double checksum = 0;
  for (const auto& p : mesh.points) checksum += static_cast<double>(p.x()) + static_cast<double>(p.y()) + static_cast<double>(p.z());
  // vertex attributes (geometry_utils.cpp:66-72): sums that tell khr_config.mesh_attr_source 0 / 1 apart (oracle/ref_recipe/match_switches.py)
  uint64_t color_sum = 0, label_sum = 0, stamp_sum = 0;
  for (const auto& c : mesh.colors) color_sum += static_cast<uint64_t>(c.r) + c.g + c.b;
  for (const auto l : mesh.labels) label_sum += l;
  for (const auto t : mesh.stamps) stamp_sum += t;  // (mod 2^64)

  writeNpy<int32_t>(out_dir, "block_indices", idx, {nb, 3});
  writeNpy<float>(out_dir, "distance", distance, {nb, nv});
  writeNpy<float>(out_dir, "weight", weight, {nb, nv});
  writeNpy<uint8_t>(out_dir, "flags", flags, {nb, nv});
  writeNpy<uint8_t>(out_dir, "sem_label", sem_label, {nb, nv});
  writeNpy<uint64_t>(out_dir, "last_observed", last_observed, {nb, nv});
  writeNpy<uint8_t>(out_dir, "color", color, {nb, nv, 4});
  writeNpy<int64_t>(out_dir, "n_clusters", n_clusters, {static_cast<size_t>(kN)});
  writeNpy<int64_t>(out_dir, "dyn_pixels", dyn_pixels, {static_cast<size_t>(kN)});
  writeNpy<int64_t>(out_dir, "removed_counts", removed_counts, {removed_counts.size()});
  writeNpy<int64_t>(out_dir, "mesh_vertices", {static_cast<int64_t>(mesh.points.size())}, {});
  writeNpy<double>(out_dir, "mesh_checksum", {checksum}, {});
  writeNpy<uint64_t>(out_dir, "mesh_color_checksum", {color_sum}, {});
  writeNpy<uint64_t>(out_dir, "mesh_label_checksum", {label_sum}, {});
  writeNpy<uint64_t>(out_dir, "mesh_stamp_checksum", {stamp_sum}, {});
  writeNpy<uint64_t>(out_dir, "stamps", stamps, {static_cast<size_t>(kN)});
  writeNpy<double>(out_dir, "poses", poses, {static_cast<size_t>(kN), 4, 4});
  writeNpy<double>(out_dir, "depth", depth_sums, {static_cast<size_t>(kN)});
  std::printf("wrote %zu blocks, %zu mesh vertices to %s\n", nb, mesh.points.size(), out_dir.c_str());
  return 0;
}
