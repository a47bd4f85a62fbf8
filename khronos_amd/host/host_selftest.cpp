// host_selftest.cpp — GPU-free checks of the host-side logic (run by tests/test_cpu_host.py):
// YAML-subset loader against the reference's mapper config keys, config validation, FrameDataBuffer
// store / trim known answers (frame_data_buffer.cpp:57-123), object-map sizing (mesh_object_extractor.cpp:201-228).
#include <type_traits>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
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

// Tracker replay (tests/test_cpu_host.py): a scenario of frames with semantic / dynamic clusters (voxel sets and
// boxes given) on stdin, the track list after every frame as JSON lines on stdout.  Format:
//   C <tracker> <track_by> <association> <min_semantic_iou> <min_cosine_sim> <min_cross_iou> <max_dynamic_distance>
//     <temporal_window> <min_num_observations> <voxel_size>
//   F <stamp>  |  S <id> <category> <min3> <max3> <nvox> <xyz>*  |  D <id> <min3> <max3> <nvox> <xyz>*  |  E (end of frame)
static int trackerReplay() {
  std::unique_ptr<Tracker> tracker;
  auto data = std::make_shared<FrameData>();
  std::string tok;
  auto readCluster = [&](bool semantic) {
    MeasurementCluster c;
    std::cin >> c.id;
    if (semantic) {
      int cat;
      std::cin >> cat;
      c.semantics = SemanticClusterInfo(cat);
    }
    float lo[3], hi[3];
    for (float& v : lo) std::cin >> v;
    for (float& v : hi) std::cin >> v;
    c.bounding_box.include(lo);
    c.bounding_box.include(hi);
    size_t n;
    std::cin >> n;
    c.voxels.resize(n);
    for (auto& v : c.voxels) std::cin >> v[0] >> v[1] >> v[2];
    std::sort(c.voxels.begin(), c.voxels.end());
    c.num_pixels = n;
    return c;
  };
  while (std::cin >> tok) {
    if (tok == "C") {
      std::string kind, by, assoc;
      MaxIoUTracker::Config c;
      std::cin >> kind >> by >> assoc >> c.min_semantic_iou >> c.min_cosine_sim >> c.min_cross_iou >> c.max_dynamic_distance >>
          c.temporal_window >> c.min_num_observations >> c.voxel_size;
      c.track_by = by == "voxels" ? MaxIoUTracker::Config::TrackBy::kVoxels : MaxIoUTracker::Config::TrackBy::kBouningBox;
      c.semantic_association = assoc == "assign_track" ? MaxIoUTracker::Config::SemanticAssociation::kAssignTrack
                                                       : MaxIoUTracker::Config::SemanticAssociation::kAssignCluster;
      if (kind == "external") {
        ExternalTracker::Config e;
        e.temporal_window = c.temporal_window;
        e.min_num_observations = c.min_num_observations;
        tracker = std::make_unique<ExternalTracker>(e);
      } else {
        tracker = std::make_unique<MaxIoUTracker>(c);
      }
    } else if (tok == "F") {
      data = std::make_shared<FrameData>();
      std::cin >> data->input.timestamp_ns;
    } else if (tok == "S") {
      data->semantic_clusters.push_back(readCluster(true));
    } else if (tok == "D") {
      data->dynamic_clusters.push_back(readCluster(false));
    } else if (tok == "E") {
      tracker->processInput(*data);
      std::printf("[");
      bool first = true;
      for (const Track& t : tracker->getTracks()) {
        const Observation& o = t.observations.back();
        std::printf("%s{\"id\": %d, \"dyn\": %d, \"active\": %d, \"conf\": %.9g, \"first\": %llu, \"last\": %llu, \"cat\": %d, "
                    "\"n_obs\": %zu, \"obs\": [%llu, %d, %d], \"n_vox\": %zu, \"centroid\": [%.9g, %.9g, %.9g]}",
                    first ? "" : ", ", t.id, int(t.is_dynamic), int(t.is_active), t.confidence,
                    static_cast<unsigned long long>(t.first_seen), static_cast<unsigned long long>(t.last_seen),
                    t.semantics ? t.semantics->category_id : -1, t.observations.size(), static_cast<unsigned long long>(o.stamp),
                    o.semantic_cluster_id, o.dynamic_cluster_id, t.last_voxels.size(), t.last_centroid[0], t.last_centroid[1],
                    t.last_centroid[2]);
        first = false;
      }
      std::printf("]\n");
    }
  }
  return 0;
}

// FrameDataBuffer replay (tests/test_cpu_ref_pin.py): a script of operations on stdin, after each one a line
//   <size> <stamp of the latest frame, 0 when empty> <one 0/1 per stamp stored so far: getData(stamp) != nullptr>
// Format:  B <max_buffer_size> <store_every_n_frames>  |  S <stamp>  |  T <n_tracks> { <n_observations> <stamp>* }*
static int bufferReplay() {
  std::unique_ptr<FrameDataBuffer> buffer;
  std::vector<TimeStamp> stamps;
  std::string tok;
  while (std::cin >> tok) {
    if (tok == "B") {
      FrameDataBuffer::Config c;
      std::cin >> c.max_buffer_size >> c.store_every_n_frames;
      buffer = std::make_unique<FrameDataBuffer>(c);
      continue;
    }
    if (tok == "S") {
      auto f = std::make_shared<FrameData>();
      std::cin >> f->input.timestamp_ns;
      stamps.push_back(f->input.timestamp_ns);
      buffer->storeData(f);
    } else if (tok == "T") {
      size_t n_tracks;
      std::cin >> n_tracks;
      Tracks tracks(n_tracks);
      for (Track& t : tracks) {
        size_t n_obs;
        std::cin >> n_obs;
        t.observations.resize(n_obs);
        for (Observation& o : t.observations) std::cin >> o.stamp;
      }
      buffer->trimBuffer(tracks);
    }
    std::printf("%zu %llu", buffer->size(), static_cast<unsigned long long>(buffer->size() ? buffer->getLatestData().input.timestamp_ns : 0));
    for (TimeStamp st : stamps) std::printf(" %d", buffer->getData(st) ? 1 : 0);
    std::printf("\n");
  }
  return 0;
}

// Dynamic-object replay (tests/test_cpu_ref_pin.py): frames with dynamic clusters given as boxes, tracks to extract; one line per
// track: "null" or the object's trajectory length, first / last observed, box (min, max).  A cluster's centroid is the mean of its
// two box corners, accumulated the way utils::computeCentroid does (geometry_utils.cpp:44-50).  Format:
//   X <min_dynamic_displacement> <min_object_allocation_confidence>
//   F <stamp> <n> { <id> <lo3> <hi3> }*   |   K <confidence> <first_seen> <last_seen> <n_obs> { <stamp> <dynamic_cluster_id> }*
static int dynamicObjectReplay() {
  MeshObjectExtractor::Config cfg;
  FrameDataBuffer::Config bc;
  bc.max_buffer_size = 4096;
  FrameDataBuffer buffer(bc);
  std::unique_ptr<MeshObjectExtractor> extractor;
  std::string tok;
  while (std::cin >> tok) {
    if (tok == "X") {
      std::cin >> cfg.min_dynamic_displacement >> cfg.min_object_allocation_confidence;
      extractor = std::make_unique<MeshObjectExtractor>(cfg, khr_config{});
    } else if (tok == "F") {
      auto f = std::make_shared<FrameData>();
      size_t n;
      std::cin >> f->input.timestamp_ns >> n;
      for (size_t k = 0; k < n; ++k) {
        MeasurementCluster c;
        float lo[3], hi[3];
        std::cin >> c.id;
        for (float& v : lo) std::cin >> v;
        for (float& v : hi) std::cin >> v;
        c.bounding_box.include(lo);
        c.bounding_box.include(hi);
        for (int i = 0; i < 3; ++i) c.centroid[i] = ((0.f + lo[i]) + hi[i]) / 2;
        f->dynamic_clusters.push_back(c);
      }
      f->num_dynamic_clusters = static_cast<int>(n);
      buffer.storeData(f);
    } else if (tok == "K") {
      Track t;
      t.is_dynamic = true;
      size_t n;
      std::cin >> t.confidence >> t.first_seen >> t.last_seen >> n;
      t.observations.resize(n);
      for (Observation& o : t.observations) std::cin >> o.stamp >> o.dynamic_cluster_id;
      const auto obj = extractor->extractObject(t, buffer);
      if (!obj) {
        std::printf("null\n");
      } else {
        std::printf("%zu %llu %llu %.9g %.9g %.9g %.9g %.9g %.9g\n", obj->trajectory_positions.size(),
                    static_cast<unsigned long long>(obj->first_observed_ns[0]), static_cast<unsigned long long>(obj->last_observed_ns[0]),
                    obj->bounding_box.min[0], obj->bounding_box.min[1], obj->bounding_box.min[2], obj->bounding_box.max[0], obj->bounding_box.max[1],
                    obj->bounding_box.max[2]);
      }
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::strcmp(argv[1], "--tracker") == 0) return trackerReplay();
  if (argc > 1 && std::strcmp(argv[1], "--dynobj") == 0) return dynamicObjectReplay();
  if (argc > 1 && std::strcmp(argv[1], "--buffer") == 0) return bufferReplay();
  if (argc > 2 && std::strcmp(argv[1], "--yaml-items") == 0) {
    // the loader's own parser on a file: prints every entry of the top-level sequence `items` -- a scalar as "S <text>", a mapping
    // as "M k=v k=v" (tests/test_cpu_host.py: scalar list items that contain colons)
    std::ifstream in(argv[2]);
    std::stringstream ss;
    ss << in.rdbuf();
    const khronos_amd::YamlNode root = khronos_amd::parseYaml(ss.str());
    for (const khronos_amd::YamlNode* it : root.at("items").items()) {
      if (!it->is_map || it->children.empty()) {
        std::printf("S %s\n", it->scalar.c_str());
      } else {
        std::printf("M");
        for (const auto& kv : it->children) std::printf(" %s=%s", kv.first.c_str(), kv.second.scalar.c_str());
        std::printf("\n");
      }
    }
    return 0;
  }
  // ---- YAML ----
  if (argc > 1) {
    std::ifstream in(argv[1]);
    std::stringstream ss;
    ss << in.rdbuf();
    const auto cfg = ActiveWindow::Config::fromYamlString(ss.str());
    cfg.checkValid();
    {  // the sub-modules the ActiveWindow constructor would build from this config (active_window.cpp:83-99), those that need no
       // device: their constructors hold the reference's checkValid constraints (tests/test_cpu_ref_pin.py feeds violations)
      FrameDataBuffer buffer(cfg.frame_data_buffer);
      if (cfg.motion_detector_type == "FreeSpaceMotionDetector") FreeSpaceMotionDetector detector(cfg.motion_detector);
      if (cfg.tracker_type == "MaxIouTracker") {
        MaxIoUTracker tracker(cfg.tracker);
      } else if (cfg.tracker_type == "ExternalTracker") {
        ExternalTracker::Config ec;
        ec.temporal_window = cfg.tracker.temporal_window;
        ec.min_num_observations = cfg.tracker.min_num_observations;
        ExternalTracker tracker(ec);
      }
      if (cfg.object_extractor_type == "MeshObjectExtractor") MeshObjectExtractor extractor(cfg.object_extractor, khr_config{});
    }
    std::printf("{\"voxel_size\": %g, \"truncation_distance\": %g, \"voxels_per_side\": %d, \"with_semantics\": %d, "
                "\"min_output_separation\": %g, \"motion_detector\": \"%s\", \"md_min_cluster_size\": %d, "
                "\"md_min_separation_distance\": %g, \"md_max_range\": %g, \"temporal_window\": %g, \"object_extractor\": \"%s\", "
                "\"min_object_volume\": %g, \"max_buffer_size\": %zu, \"object_detector\": \"%s\", \"tracker\": \"%s\", "
                "\"only_extract_reconstructed_objects\": %d, \"object_reconstruction_resolution\": %g, "
                "\"alloc_candidate\": \"%s\", \"color_blend_weight\": \"%s\", \"mesh_attr_source\": \"%s\", \"mesh_degenerate_eps\": %g, "
                "\"object_interpolation_method\": \"%s\", \"object_max_weight\": %g, \"object_use_weight_dropoff\": %d, "
                "\"object_color_blend_weight\": \"%s\", \"object_mesh_min_weight\": %g, \"object_mesh_attr_source\": \"%s\"}\n",
                cfg.volumetric_map.voxel_size, cfg.volumetric_map.truncation_distance, cfg.volumetric_map.voxels_per_side,
                int(cfg.volumetric_map.with_semantics), cfg.min_output_separation, cfg.motion_detector_type.c_str(),
                cfg.motion_detector.min_cluster_size, cfg.motion_detector.min_separation_distance, cfg.motion_detector.max_range,
                cfg.tracking_integrator.temporal_window, cfg.object_extractor_type.c_str(), cfg.object_extractor.min_object_volume,
                cfg.frame_data_buffer.max_buffer_size, cfg.object_detector_type.c_str(), cfg.tracker_type.c_str(),
                int(cfg.object_extractor.only_extract_reconstructed_objects), cfg.object_extractor.object_reconstruction_resolution,
                cfg.projective_integrator.alloc_candidate.c_str(), cfg.projective_integrator.color_blend_weight.c_str(),
                cfg.mesh_integrator.attr_source.c_str(), cfg.mesh_integrator.degenerate_eps,
                cfg.object_extractor.projective_integrator.interpolation_method.c_str(), cfg.object_extractor.projective_integrator.max_weight,
                int(cfg.object_extractor.projective_integrator.use_weight_dropoff),
                cfg.object_extractor.projective_integrator.color_blend_weight.c_str(), cfg.object_extractor.mesh_integrator.min_weight,
                cfg.object_extractor.mesh_integrator.attr_source.c_str());
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
  // ---- plugin surface (active_window.h:67,99,134,190-192): a hydra::ActiveWindowModule registered as "ActiveWindow" ----
  {
    static_assert(std::is_base_of<hydra::ActiveWindowModule, ActiveWindow>::value, "ActiveWindow must be an ActiveWindowModule");
    static_assert(std::is_constructible<ActiveWindow, const ActiveWindow::Config&, const ActiveWindow::OutputQueue::Ptr&>::value,
                  "ActiveWindow(const Config&, const OutputQueue::Ptr&)");
    CHECK(hydra::ActiveWindowFactory::has("ActiveWindow"));
    CHECK(!hydra::ActiveWindowFactory::has("NoSuchWindow"));
    CHECK(hydra::ActiveWindowFactory::create("NoSuchWindow", "", nullptr) == nullptr);
  }
  // ---- khronos_sinks (active_window.cpp:70,80): a YAML list of {type, ...} entries, instantiated through the sink registry ----
  {
    const char* yaml =
        "active_window:\n"
        "  type: \"ActiveWindow\"\n"
        "  min_output_separation: 0.4\n"
        "  khronos_sinks:\n"
        "    - type: CountingSink\n"
        "      every_n: 3\n"
        "    - type: NobodyRegisteredThis\n"
        "    - type: CountingSink\n"
        "      every_n: 5\n"
        "  frame_data_buffer:\n"
        "    max_buffer_size: 7\n";
    const auto cfg = ActiveWindow::Config::fromYamlString(yaml);
    CHECK(cfg.khronos_sinks.size() == 3);
    CHECK(cfg.frame_data_buffer.max_buffer_size == 7 && cfg.min_output_separation == 0.4f);  // (the keys behind the list still parse)
    std::string t0, t1;
    int n0 = 0, n2 = 0;
    cfg.khronos_sinks[0].read("type", t0);
    cfg.khronos_sinks[0].read("every_n", n0);
    cfg.khronos_sinks[1].read("type", t1);
    cfg.khronos_sinks[2].read("every_n", n2);
    CHECK(t0 == "CountingSink" && n0 == 3 && t1 == "NobodyRegisteredThis" && n2 == 5);
    CHECK(ActiveWindow::Config::fromYamlString("active_window:\n  khronos_sinks: []\n").khronos_sinks.empty());
    static int made = 0;
    CHECK(ActiveWindow::registerKhronosSink("CountingSink", [](const khronos_amd::YamlNode& n) -> ActiveWindow::KhronosSink {
      int every = 1;
      n.read("every_n", every);
      made += every;
      return [](const FrameData&, const VolumetricMap&, const Tracks&) {};
    }));
    CHECK(!ActiveWindow::registerKhronosSink("CountingSink", nullptr));  // the name is taken
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
  // ---- MaxIoUTracker pieces (max_iou_tracker.cpp:565-576, track.cpp:42-70) ----
  {
    const std::vector<GlobalIndex> a = {{0, 0, 0}, {0, 0, 1}, {1, 0, 0}}, b = {{0, 0, 1}, {1, 0, 0}, {2, 2, 2}, {3, 0, 0}};
    CHECK(MaxIoUTracker::computeIoUVoxels(a, b) == 2.f / 5.f);
    CHECK(MaxIoUTracker::computeIoUVoxels(a, a) == 1.f);
    BoundingBox p, q;
    const float p0[3] = {0, 0, 0}, p1[3] = {2, 2, 2}, q0[3] = {1, 1, 1}, q1[3] = {3, 3, 3};
    p.include(p0); p.include(p1); q.include(q0); q.include(q1);
    CHECK(MaxIoUTracker::computeIoUBoundingBox(p, q) == 1.f / 15.f);
    Track t;
    t.updateSemantics(SemanticClusterInfo(3));
    CHECK(t.semantics && t.semantics->category_id == 3 && t.num_features == 0);
    t.updateSemantics(SemanticClusterInfo(3, {1.f, 0.f}));
    CHECK(t.num_features == 1 && t.semantics->feature.size() == 2);
    t.updateSemantics(SemanticClusterInfo(3, {0.f, 1.f}));
    CHECK(t.num_features == 2 && t.semantics->feature[0] == 0.5f && t.semantics->feature[1] == 0.5f);
    t.updateSemantics(SemanticClusterInfo(3));  // a feature is never replaced by "no feature"
    CHECK(t.num_features == 2 && t.semantics->feature.size() == 2);
    bool threw = false;
    try { MaxIoUTracker::Config c; c.track_by = MaxIoUTracker::Config::TrackBy::kVoxels; c.min_cross_iou = 1.5f; MaxIoUTracker m(c); }
    catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  // ---- hydra::timing stand-in: the reference's scope names, stats.csv rows ----
  {
    auto& rec = hydra::timing::ElapsedTimeRecorder::instance();
    rec.reset();
    rec.record("active_window/all", 0.25);
    rec.record("active_window/all", 0.75);
    { hydra::timing::ScopedTimer t("active_window/update_map", 0); }
    const auto st = rec.stats();
    CHECK(st.at("active_window/all").count == 2 && st.at("active_window/all").min == 0.25 && st.at("active_window/all").max == 0.75);
    CHECK(st.at("active_window/all").sum == 1.0 && st.count("active_window/update_map") == 1);
    const std::string path = "/tmp/khr_selftest_stats.csv";
    CHECK(rec.logStats(path));
    std::ifstream f(path);
    std::string header, row;
    std::getline(f, header);
    std::getline(f, row);
    CHECK(header == "name,mean[s],min[s],max[s],std-dev[s],count");
    CHECK(row.rfind("active_window/all,0.5,0.25,0.75,", 0) == 0);
    rec.reset();
  }
  // ---- ObjectWorkerPool (object_worker_pool.cpp:56-146) with a device-free extractor ----
  {
    struct FakeExtractor : ObjectExtractor {
      std::atomic<int>* concurrent;
      std::atomic<int>* peak;
      explicit FakeExtractor(std::atomic<int>* c, std::atomic<int>* p) : concurrent(c), peak(p) {}
      std::shared_ptr<KhronosObjectAttributes> extractObject(const Track& track, const FrameDataBuffer&) override {
        const int now = ++*concurrent;
        int seen = peak->load();
        while (now > seen && !peak->compare_exchange_weak(seen, now)) {}
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        --*concurrent;
        if (track.id % 5 == 4) return nullptr;  // (an extractor may decline a track: confidence gate, empty mesh)
        auto o = std::make_shared<KhronosObjectAttributes>();
        o->semantic_label = track.id;
        return o;
      }
    };
    std::atomic<int> concurrent{0}, peak{0};
    ObjectWorkerPool::Config pc;
    pc.num_workers = 3;
    ObjectWorkerPool pool(pc, [&]() -> std::unique_ptr<ObjectExtractor> { return std::make_unique<FakeExtractor>(&concurrent, &peak); });
    FrameDataBuffer buffer{FrameDataBuffer::Config{}};
    for (int i = 0; i < 10; ++i) {
      Track t;
      t.id = i;
      pool.submit(static_cast<TimeStamp>(i), std::move(t), buffer);
    }
    Track blocking;
    blocking.id = 100;
    const auto direct = pool.runBlocking(blocking, buffer);  // shares extractor 0 with worker 0, serialised
    CHECK(direct && direct->semantic_label == 100);
    pool.join();
    CHECK(pool.numRunning() == 0);
    std::vector<std::shared_ptr<KhronosObjectAttributes>> out;
    CHECK(pool.fill(out) == 8);  // ids 4 and 9 were declined
    std::vector<int> ids;
    for (const auto& o : out) ids.push_back(o->semantic_label);
    std::sort(ids.begin(), ids.end());
    CHECK((ids == std::vector<int>{0, 1, 2, 3, 5, 6, 7, 8}));
    CHECK(peak.load() >= 2 && peak.load() <= 3);  // several workers at once, never more than num_workers
    CHECK(pool.fill(out) == 0);
    // stop with work queued: the pool comes down without running it
    for (int i = 0; i < 4; ++i) {
      Track t;
      t.id = 200 + i;
      pool.submit(0, std::move(t), buffer);
    }
    pool.stop();
    // no extractor configured: submit / runBlocking are no-ops (:93-95, :102-104)
    ObjectWorkerPool none(pc, nullptr);
    Track t2;
    none.submit(0, std::move(t2), buffer);
    none.join();
    CHECK(!none.hasExtractor() && none.runBlocking(blocking, buffer) == nullptr && none.fill(out) == 0);
  }
  std::printf("host selftest ok\n");
  return 0;
}
