// host_selftest.cpp — GPU-free checks of the host-side logic (run by tests/test_cpu_host.py):
// YAML-subset loader against the reference's mapper config keys, config validation, FrameDataBuffer
// store / trim known answers (frame_data_buffer.cpp:57-123), object-map sizing (mesh_object_extractor.cpp:201-228).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "active_window.h"

using namespace khronos;

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static FrameData::Ptr frame(uint64_t stamp) {
  auto f = std::make_shared<FrameData>();
  f->input.timestamp_ns = stamp;
  return f;
}

int main(int argc, char** argv) {
  // ---- YAML ----
  if (argc > 1) {
    std::ifstream in(argv[1]);
    std::stringstream ss;
    ss << in.rdbuf();
    const auto cfg = ActiveWindow::Config::fromYamlString(ss.str());
    cfg.checkValid();
    std::printf("{\"voxel_size\": %g, \"truncation_distance\": %g, \"voxels_per_side\": %d, \"with_semantics\": %d, "
                "\"min_output_separation\": %g, \"motion_detector\": \"%s\", \"md_min_cluster_size\": %d, "
                "\"md_min_separation_distance\": %g, \"md_max_range\": %g, \"temporal_window\": %g, \"object_extractor\": \"%s\", "
                "\"min_object_volume\": %g, \"max_buffer_size\": %zu, \"object_detector\": \"%s\", \"tracker\": \"%s\", "
                "\"only_extract_reconstructed_objects\": %d, \"object_reconstruction_resolution\": %g}\n",
                cfg.volumetric_map.voxel_size, cfg.volumetric_map.truncation_distance, cfg.volumetric_map.voxels_per_side,
                int(cfg.volumetric_map.with_semantics), cfg.min_output_separation, cfg.motion_detector_type.c_str(),
                cfg.motion_detector.min_cluster_size, cfg.motion_detector.min_separation_distance, cfg.motion_detector.max_range,
                cfg.tracking_integrator.temporal_window, cfg.object_extractor_type.c_str(), cfg.object_extractor.min_object_volume,
                cfg.frame_data_buffer.max_buffer_size, cfg.object_detector_type.c_str(), cfg.tracker_type.c_str(),
                int(cfg.object_extractor.only_extract_reconstructed_objects), cfg.object_extractor.object_reconstruction_resolution);
    return 0;
  }
  // ---- config validation (tracking_integrator.cpp:61-65) ----
  {
    ActiveWindow::Config c;
    c.checkValid();
    c.tracking_integrator.neighbor_connectivity = 7;
    bool threw = false;
    try { c.checkValid(); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  // ---- FrameDataBuffer: capped FIFO (frame_data_buffer.cpp:88-109) ----
  {
    FrameDataBuffer::Config bc;
    bc.max_buffer_size = 3;
    FrameDataBuffer b(bc);
    for (uint64_t s = 1; s <= 5; ++s) b.storeData(frame(s));
    CHECK(b.size() == 3);
    CHECK(b.getData(1) == nullptr && b.getData(2) == nullptr);  // older than the oldest stamp
    CHECK(b.getData(3) && b.getData(5) && b.getLatestData().input.timestamp_ns == 5);
    // trim keeps only frames referenced by a track observation (:57-86)
    Tracks tracks(1);
    tracks[0].observations.push_back({4, 1, -1});
    b.trimBuffer(tracks);
    CHECK(b.size() == 1 && b.getData(4) && !b.getData(5));
  }
  // ---- FrameDataBuffer: store_every_n_frames overwrites the newest entry in between ----
  {
    FrameDataBuffer::Config bc;
    bc.max_buffer_size = 10;
    bc.store_every_n_frames = 3;
    FrameDataBuffer b(bc);
    for (uint64_t s = 1; s <= 7; ++s) b.storeData(frame(s));
    // appended: 1, 4, 7; frames 2,3 / 5,6 overwrote the newest entry and were overwritten in turn
    CHECK(b.size() == 3);
    CHECK(b.getData(7) && b.getLatestData().input.timestamp_ns == 7);
  }
  // ---- object map sizing (mesh_object_extractor.cpp:201-228) ----
  {
    MeshObjectExtractor::Config oc;
    BoundingBox e;
    const float a[3] = {1.f, 2.f, 0.f}, bq[3] = {1.5f, 2.2f, 1.0f};
    e.include(a);
    e.include(bq);
    CHECK(e.maxDimension() == 1.0f);
    CHECK(MeshObjectExtractor::objectVoxelSize(oc, e) == 0.02f);  // -0.02 => 2 % of the max extent
    oc.object_reconstruction_resolution = 0.05f;
    CHECK(MeshObjectExtractor::objectVoxelSize(oc, e) == 0.05f);
    int32_t mn[3], mx[3];
    MeshObjectExtractor::objectBlockRange(e, 0.16f, mn, mx);  // centre -/+ FULL dimensions (2x the box)
    CHECK(mn[2] == static_cast<int32_t>(std::floor((0.5f - 1.0f) / 0.16f)) && mx[2] == static_cast<int32_t>(std::floor(1.5f / 0.16f)));
    CHECK(mn[0] == static_cast<int32_t>(std::floor((1.25f - 0.5f) * (1.f / 0.16f))));
  }
  std::printf("host selftest ok\n");
  return 0;
}
