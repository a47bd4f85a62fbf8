// aw_demo.cpp — drives khronos::ActiveWindow (host mirror) over a synthetic stream, the way the Hydra module
// thread drives the reference (spinOnce per InputPacket, finishMapping at shutdown), and prints a JSON
// summary that tests/test_gpu_host.py compares with the step-wise C-ABI path and the oracle.
// usage: aw_demo <config.yaml> <width> <height> <frames> [object_label]
//   object_label >= 0: the stand-in detector / tracker below; otherwise the plugins named in the config
#include <execinfo.h>
#include <csignal>
#include <unistd.h>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include <chrono>

#include <hip/hip_runtime_api.h>

#include "active_window.h"
#include "ray_verificator.h"

extern "C" {
void* synth_create(uint32_t seed, int num_static, int with_mover);
void synth_destroy(void* s);
void synth_render(void* sp, int W, int H, float fx, float fy, float cx, float cy, const double* T, double t_sec,
                  float max_depth, float noise_sigma_rel, uint32_t noise_seed, float* depth, uint8_t* rgb, int32_t* label,
                  int num_threads);
}

using namespace khronos;

// test stand-in for the instance-forwarding object detector + id tracker (the real
// ConnectedSemantics / MaxIoUTracker are host plugins that are not part of the device path): pixels carrying
// `label` become semantic cluster 1 of the frame (object_image = 1 there), with the bounding box of their
// world-frame vertices, and one track collects them.
struct LabelObjectDetector : ObjectDetector {
  int label;
  const int32_t* current_labels = nullptr;
  explicit LabelObjectDetector(int l) : label(l) {}
  void processInput(const VolumetricMap& map, FrameData& data) override {
    const size_t n = static_cast<size_t>(data.input.sensor.width) * data.input.sensor.height;
    std::vector<int32_t> obj(n, 0);
    const std::vector<float> vm = data.input.vertexMap();
    const std::vector<float> range = data.input.rangeImage();
    MeasurementCluster cl;
    cl.id = 1;
    for (size_t i = 0; i < n; ++i)
      if (current_labels[i] == label && range[i] > 0.f) {
        obj[i] = 1;
        cl.bounding_box.include(&vm[3 * i]);
        ++cl.num_pixels;
      }
    khr_set_frame_image(map.ctx(), data.input.slot, /*object image*/ 1, obj.data(), 0);
    if (cl.num_pixels > 0) data.semantic_clusters.push_back(cl);
  }
};

struct SingleTrackTracker : Tracker {
  void processInput(FrameData& data) override {
    if (data.semantic_clusters.empty()) return;
    if (tracks_.empty()) {
      Track t;
      t.id = 0;
      t.first_seen = data.input.timestamp_ns;
      t.semantics = SemanticClusterInfo(1);
      tracks_.push_back(t);
    }
    Track& t = tracks_[0];
    t.last_seen = data.input.timestamp_ns;
    t.observations.push_back({data.input.timestamp_ns, 1, -1});
    t.confidence = std::min(1.f, static_cast<float>(t.observations.size()) / 4.f);
  }
};

static void circlePose(double t, double* T) {
  const double th = 2.0 * M_PI * t / 10.0, yaw = th + M_PI / 2;
  const double f[3] = {std::cos(yaw), std::sin(yaw), 0}, r[3] = {f[1], -f[0], 0}, d[3] = {0, 0, -1};
  const double p[3] = {1.5 * std::cos(th), 1.5 * std::sin(th), 1.5};
  for (int i = 0; i < 3; ++i) {
    T[4 * i + 0] = r[i];
    T[4 * i + 1] = d[i];
    T[4 * i + 2] = f[i];
    T[4 * i + 3] = p[i];
  }
  T[12] = T[13] = T[14] = 0;
  T[15] = 1;
}

// aw_demo --rayver <policy>: drives khronos::RayVerificator (host mirror over the device index) with a scenario from
// stdin and prints the check results as JSON lines (tests/test_gpu_host.py compares them with the oracle).
//   P <n> then n lines "<stamp> x y z"          agent poses (sorted by time)
//   V <n> then n lines "<first> <last> x y z"   mesh vertices; every P / V block is followed by an update
//   Q <n> then n lines "<earliest> <latest> x y z"
static int rayverDemo(const char* policy) {
  RayVerificator::Config cfg;
  const char* names[] = {"First", "Last", "FirstAndLast", "Middle", "All"};
  for (int i = 0; i < 5; ++i)
    if (std::string(policy) == names[i]) cfg.ray_policy = static_cast<RayVerificator::Config::RayPolicy>(i);
  std::cin >> cfg.block_size >> cfg.radial_tolerance >> cfg.depth_tolerance >> cfg.active_window_duration;
  RayVerificator rv(cfg);
  std::vector<uint64_t> pose_stamps, first_seen, last_seen;
  std::vector<float> pose_pos, vertices;
  std::string tok;
  while (std::cin >> tok) {
    size_t n = 0;
    std::cin >> n;
    if (tok == "P") {
      for (size_t i = 0; i < n; ++i) {
        uint64_t t; float x, y, z;
        std::cin >> t >> x >> y >> z;
        pose_stamps.push_back(t);
        pose_pos.insert(pose_pos.end(), {x, y, z});
      }
    } else if (tok == "V") {
      for (size_t i = 0; i < n; ++i) {
        uint64_t a, b; float x, y, z;
        std::cin >> a >> b >> x >> y >> z;
        first_seen.push_back(a);
        last_seen.push_back(b);
        vertices.insert(vertices.end(), {x, y, z});
      }
      rv.updateData(pose_stamps, pose_pos, vertices, first_seen, last_seen);
    } else if (tok == "Q") {
      std::vector<float> pts;
      std::vector<uint64_t> t0, t1;
      for (size_t i = 0; i < n; ++i) {
        uint64_t a, b; float x, y, z;
        std::cin >> a >> b >> x >> y >> z;
        t0.push_back(a);
        t1.push_back(b);
        pts.insert(pts.end(), {x, y, z});
      }
      const auto res = rv.checkMany(pts, t0, t1);
      std::printf("{\"rays\": %zu, \"results\": [", rv.numRays());
      for (size_t i = 0; i < res.size(); ++i) {
        std::printf("%s{\"present\": [", i ? ", " : "");
        for (size_t k = 0; k < res[i].present.size(); ++k) std::printf("%s%" PRIu64, k ? ", " : "", res[i].present[k]);
        std::printf("], \"absent\": [");
        for (size_t k = 0; k < res[i].absent.size(); ++k) std::printf("%s%" PRIu64, k ? ", " : "", res[i].absent[k]);
        std::printf("]}");
      }
      std::printf("]}\n");
    } else if (tok == "C") {  // C <n> <resolution> <window> <relative> <absence> <presence>, then n lines "<forward> <earliest> <latest> x y z"
      RayChangeDetector::Config dc;
      int rel = 1;
      std::cin >> dc.temporal_resolution >> dc.window_size >> rel >> dc.absence_confidence >> dc.presence_confidence;
      dc.use_relative_confidence = rel != 0;
      const RayChangeDetector det(dc);
      std::vector<float> pts;
      std::vector<uint64_t> t0, t1;
      std::vector<uint8_t> fwd;
      for (size_t i = 0; i < n; ++i) {
        int f; uint64_t a, b; float x, y, z;
        std::cin >> f >> a >> b >> x >> y >> z;
        fwd.push_back(static_cast<uint8_t>(f));
        t0.push_back(a);
        t1.push_back(b);
        pts.insert(pts.end(), {x, y, z});
      }
      const auto res = det.detectChangesMany(rv, pts, t0, t1, fwd);
      std::printf("{\"changes\": [");
      for (size_t i = 0; i < res.size(); ++i)
        std::printf("%s[%lld, %lld]", i ? ", " : "", res[i].closest_absent ? static_cast<long long>(*res[i].closest_absent) : -1ll,
                    res[i].furthest_persistent ? static_cast<long long>(*res[i].furthest_persistent) : -1ll);
      std::printf("]}\n");
    }
  }
  return 0;
}

// aw_demo --bench <config.yaml> <width> <height> <preroll> <warmup> <steps>: khronos::ActiveWindow::spinOnce itself, timed the way
// bench.py times the step-wise C ABI: every frame of the stream is rendered and put into HBM BEFORE the timed
// region (InputPacket::on_device), preroll + warmup untimed steps, then `steps` calls of step() back to back; the clock stops when
// the device has drained and the extraction workers have been joined.  One JSON line.
static int benchDemo(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: aw_demo --bench <config.yaml> <width> <height> <preroll> <warmup> <steps>\n");
    return 2;
  }
  std::ifstream in(argv[2]);
  std::stringstream ss;
  ss << in.rdbuf();
  const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), pre = std::atoi(argv[5]), warm = std::atoi(argv[6]), steps = std::atoi(argv[7]);
  const int N = pre + warm + steps;
  ActiveWindow::Config cfg = ActiveWindow::Config::fromYamlString(ss.str());
  cfg.max_frame_pixels = static_cast<uint32_t>(W) * H;
  auto out_queue = std::make_shared<ActiveWindow::OutputQueue>();
  ActiveWindow aw(cfg, out_queue);
  void* scene = synth_create(1234, 12, 1);
  const size_t n = static_cast<size_t>(W) * H;
  std::vector<float> depth(n);
  std::vector<uint8_t> rgb(n * 3);
  std::vector<int32_t> label(n);
  std::vector<hydra::InputPacket> pkts(static_cast<size_t>(N));
  std::vector<void*> dev;
  for (int i = 0; i < N; ++i) {
    hydra::InputPacket& pkt = pkts[static_cast<size_t>(i)];
    pkt.timestamp_ns = static_cast<uint64_t>(std::llround((1.0 + 0.1 * i) * 1e9));
    circlePose(0.1 * i, pkt.world_T_body);
    pkt.sensor = {W, H, W / 2.f, W / 2.f, W / 2.f, H / 2.f, 0.1f, 5.f};
    synth_render(scene, W, H, pkt.sensor.fx, pkt.sensor.fy, pkt.sensor.cx, pkt.sensor.cy, pkt.world_T_body, 0.1 * i, 5.f, 0.f,
                 1234u + 7919u * i, depth.data(), rgb.data(), label.data(), 0);
    void *d = nullptr, *c = nullptr, *l = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess || hipMalloc(&c, n * 3) != hipSuccess || hipMalloc(&l, n * 4) != hipSuccess ||
        hipMemcpy(d, depth.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(c, rgb.data(), n * 3, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(l, label.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) {
      std::fprintf(stderr, "aw_demo --bench: device frames\n");
      return 1;
    }
    dev.insert(dev.end(), {d, c, l});
    pkt.depth = static_cast<const float*>(d);
    pkt.color = static_cast<const uint8_t*>(c);
    pkt.labels = static_cast<const int32_t*>(l);
    pkt.on_device = true;
    pkt.buffers_complete = true;  // (rendered and synchronised before the timed region)
  }
  int n_out = 0;
  size_t dyn_frames = 0;
  auto run = [&](int a, int b, bool count) {
    for (int i = a; i < b; ++i) {
      khr_host_trace("step_begin");
      auto out = aw.step(pkts[static_cast<size_t>(i)]);
      hydra::ActiveWindowOutput::Ptr popped;
      while (out_queue->pop(&popped)) {}
      if (count) {
        n_out += out ? 1 : 0;
        dyn_frames += aw.numDynamicClustersOfLastFrame() > 0 ? 1 : 0;  // (not getLatestFrameData(): that completes the deferred tracker step)
      }
    }
  };
  run(0, pre + warm, false);
  khr_sync(aw.getMap().ctx());
  khr_host_trace("timed_begin");
  const auto t0 = std::chrono::steady_clock::now();
  run(pre + warm, N, true);
  khr_host_trace("join_begin");
  const auto t1 = std::chrono::steady_clock::now();
  aw.joinExtractions();
  khr_sync(aw.getMap().ctx());
  const auto t2 = std::chrono::steady_clock::now();
  const double ms_steps = std::chrono::duration<double, std::milli>(t1 - t0).count(), ms_all = std::chrono::duration<double, std::milli>(t2 - t0).count();
  std::printf("{\"what\": \"khronos::ActiveWindow::spinOnce (libkhronos_amd_host.so), frames resident in HBM, %dx%d\", \"preroll\": %d, \"warmup\": %d, "
              "\"steps\": %d, \"frames_per_s\": %.2f, \"ms_per_step\": %.5f, \"steps_ms\": %.4f, \"drain_and_join_ms\": %.4f, \"outputs\": %d, "
              "\"frames_with_dynamic_clusters\": %zu, \"tracks\": %zu}\n",
              W, H, pre, warm, steps, 1e3 * steps / ms_all, ms_all / steps, ms_steps, ms_all - ms_steps, n_out, dyn_frames, aw.getTracks().size());
  if (std::getenv("AW_BENCH_TIMERS"))  // (host view of the reference's timer scopes over the whole run: where spinOnce spends its time)
    for (const auto& kv : hydra::timing::ElapsedTimeRecorder::instance().stats())
      std::fprintf(stderr, "%-44s count %6llu  mean %8.4f ms\n", kv.first.c_str(), static_cast<unsigned long long>(kv.second.count),
                   1e3 * kv.second.sum / static_cast<double>(kv.second.count));
  aw.finishMapping();
  for (void* p : dev) hipFree(p);
  synth_destroy(scene);
  return 0;
}

static void onSegv(int sig) {  // a crash must not look like an empty result: print where it happened
  void* bt[48];
  const int n = backtrace(bt, 48);
  const char msg[] = "aw_demo: fatal signal, backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(bt, n, 2);
  _exit(128 + sig);
}

int main(int argc, char** argv) {
  std::signal(SIGSEGV, onSegv);
  std::signal(SIGABRT, onSegv);
  if (argc >= 3 && std::string(argv[1]) == "--rayver") {
    try {
      return rayverDemo(argv[2]);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "aw_demo: %s\n", e.what());
      return 1;
    }
  }
  if (argc >= 2 && std::string(argv[1]) == "--bench") {
    try {
      return benchDemo(argc, argv);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "aw_demo: %s\n", e.what());
      return 1;
    }
  }
  if (argc < 5) {
    std::fprintf(stderr, "usage: aw_demo <config.yaml> <width> <height> <frames> [object_label] [timing_stats.csv]\n");
    return 2;
  }
  std::ifstream in(argv[1]);
  std::stringstream ss;
  ss << in.rdbuf();
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
  const int object_label = argc > 5 ? std::atoi(argv[5]) : -1;
  try {
    ActiveWindow::Config cfg = ActiveWindow::Config::fromYamlString(ss.str());
    cfg.max_frame_pixels = static_cast<uint32_t>(W) * H;
    auto out_queue = std::make_shared<ActiveWindow::OutputQueue>();
    ActiveWindow aw(cfg, out_queue);
    LabelObjectDetector* det = nullptr;
    if (object_label >= 0) {
      auto d = std::make_unique<LabelObjectDetector>(object_label);
      det = d.get();
      aw.setObjectDetector(std::move(d));
      aw.setTracker(std::make_unique<SingleTrackTracker>());
    }
    int sink_calls = 0;
    // (a window without sinks queues the detector's kernels and an output's device stages with the frame's fused call and defers the
    //  tracker's association, ActiveWindow::Config::fuse_device_stages: tests run both forms and compare)
    if (!std::getenv("AW_DEMO_NO_SINK")) aw.addKhronosSink([&](const FrameData&, const VolumetricMap&, const Tracks&) { ++sink_calls; });

    void* scene = synth_create(1234, 12, 1);
    std::vector<float> depth(static_cast<size_t>(W) * H);
    std::vector<uint8_t> rgb(static_cast<size_t>(W) * H * 3);
    std::vector<int32_t> label(static_cast<size_t>(W) * H);
    std::printf("{\"info\": \"%s\", \"outputs\": [", aw.printInfo().c_str());
    int n_out = 0;
    size_t n_popped = 0;
    hydra::ActiveWindowOutput::Ptr first_out;  // kept like the frontend's queue keeps it: read after the map has moved on
    std::vector<float> first_depth;
    std::vector<uint8_t> first_rgb;
    std::vector<int32_t> first_label;
    size_t total_dyn_clusters = 0, total_sem_clusters = 0;
    for (int i = 0; i < N; ++i) {
      hydra::InputPacket pkt;
      pkt.timestamp_ns = static_cast<uint64_t>(std::llround((1.0 + 0.1 * i) * 1e9));
      circlePose(0.1 * i, pkt.world_T_body);
      pkt.sensor = {W, H, W / 2.f, W / 2.f, W / 2.f, H / 2.f, 0.1f, 5.f};
      synth_render(scene, W, H, pkt.sensor.fx, pkt.sensor.fy, pkt.sensor.cx, pkt.sensor.cy, pkt.world_T_body, 0.1 * i, 5.f, 0.f,
                   1234u + 7919u * i, depth.data(), rgb.data(), label.data(), 0);
      pkt.depth = depth.data();
      pkt.color = rgb.data();
      pkt.labels = label.data();
      if (det) det->current_labels = label.data();
      auto out = aw.step(pkt);  // the module thread's iteration: spinOnce (protected) + output queue
      {  // the consumer (hydra frontend role): takes what the module queued
        hydra::ActiveWindowOutput::Ptr popped;
        while (out_queue->pop(&popped)) n_popped += popped == out ? 1 : 0;
      }
      total_dyn_clusters += aw.getLatestFrameData().num_dynamic_clusters;
      total_sem_clusters += aw.getLatestFrameData().semantic_clusters.size();
      if (out) {
        std::printf("%s{\"stamp\": %" PRIu64 ", \"updated\": %zu, \"archived\": %zu, \"objects\": %zu}", n_out ? ", " : "", out->timestamp_ns,
                    out->updatedBlocks().size(), out->archived_mesh_indices.size(), out->graph_update.size());
        if (!first_out) {
          first_out = out;
          first_depth = depth;
          first_rgb = rgb;
          first_label = label;
        }
        ++n_out;
      }
    }
    // map checksum in sorted block order
    double checksum = 0;
    size_t n_blocks = aw.getMap().numBlocks();
    for (const auto& idx : aw.getMap().allocatedBlockIndices()) {
      const hydra::BlockCopy b = aw.getMap().cloneBlock(idx);
      for (size_t k = 0; k < b.distance.size(); ++k) checksum += static_cast<double>(b.distance[k]) * b.weight[k];
    }
    const size_t n_tracks_before = aw.getTracks().size();
    std::string track_json = "[";
    for (const Track& t : aw.getTracks()) {
      char buf[256];
      std::snprintf(buf, sizeof(buf), "%s{\"id\": %d, \"dyn\": %d, \"active\": %d, \"conf\": %.9g, \"cat\": %d, \"n_obs\": %zu, \"first\": %" PRIu64
                    ", \"last\": %" PRIu64 "}", track_json.size() > 1 ? ", " : "", t.id, int(t.is_dynamic), int(t.is_active), t.confidence,
                    t.semantics ? t.semantics->category_id : -1, t.observations.size(), t.first_seen, t.last_seen);
      track_json += buf;
    }
    track_json += "]";
    auto objects = aw.extractObjects();
    std::printf("], \"n_outputs\": %d, \"sink_calls\": %d, \"dynamic_clusters\": %zu, \"n_blocks\": %zu, \"checksum\": %.17g, "
                "\"tracks\": %zu, \"track_list\": %s, \"semantic_clusters\": %zu, \"objects\": [",
                n_out, sink_calls, total_dyn_clusters, n_blocks, checksum, n_tracks_before, track_json.c_str(), total_sem_clusters);
    for (size_t k = 0; k < objects.size(); ++k) {
      const auto& o = *objects[k];
      std::printf("%s{\"label\": %d, \"dynamic\": %d, \"vertices\": %zu, \"bbox_min\": [%.6f, %.6f, %.6f], \"bbox_max\": [%.6f, %.6f, %.6f]}", k ? ", " : "",
                  o.semantic_label, int(!o.trajectory_positions.empty()), o.mesh.numVertices(), o.bounding_box.min[0], o.bounding_box.min[1], o.bounding_box.min[2], o.bounding_box.max[0],
                  o.bounding_box.max[1], o.bounding_box.max[2]);
    }
    aw.finishMapping();
    std::printf("], \"blocks_after_finish\": %zu, \"ring_waits\": %zu, \"queued_outputs\": %zu", aw.getMap().numBlocks(), aw.numRingWaits(),
                n_popped);
    if (first_out) {  // the first output's map clone, read only now (every block has been archived by finishMapping)
      double sum = 0;
      const auto blocks = first_out->cloneUpdatedTsdf();
      for (const auto& b : blocks)
        for (size_t k = 0; k < b.distance.size(); ++k) sum += static_cast<double>(b.distance[k]) * b.weight[k];
      std::printf(", \"first_output_clone\": {\"blocks\": %zu, \"checksum\": %.17g}", blocks.size(), sum);
      // ... and its sensor_data (active_window.cpp:165): the images of ITS frame, long after the ring slot has been reused
      const auto& sd = *first_out->sensor_data;
      const std::vector<float> d = sd.depthImage(), r = sd.rangeImage(), vm = sd.vertexMap();
      const std::vector<uint8_t> c = sd.colorImage();
      const std::vector<int32_t> l = sd.labelImage();
      size_t range_valid = 0, vertex_set = 0;
      for (size_t k = 0; k < r.size(); ++k) {
        range_valid += r[k] > 0.f ? 1 : 0;
        vertex_set += (k < vm.size() / 3 && (vm[3 * k] != 0.f || vm[3 * k + 1] != 0.f || vm[3 * k + 2] != 0.f)) ? 1 : 0;
      }
      std::printf(", \"first_output_images\": {\"depth_equal\": %s, \"color_equal\": %s, \"labels_equal\": %s, \"range_valid\": %zu, \"vertices\": %zu, "
                  "\"pixels\": %zu}", d == first_depth ? "true" : "false", c == first_rgb ? "true" : "false", l == first_label ? "true" : "false",
                  range_valid, vertex_set, r.size());
      first_out.reset();
    }
    // timing/stats.csv of the reference's experiment manager (experiment_manager.cpp:251-258), same scope names
    if (argc > 6) hydra::timing::ElapsedTimeRecorder::instance().logStats(argv[6]);
    std::printf(", \"timing\": {");
    bool first_t = true;
    for (const auto& kv : hydra::timing::ElapsedTimeRecorder::instance().stats()) {
      std::printf("%s\"%s\": {\"count\": %llu, \"mean_ms\": %.4f}", first_t ? "" : ", ", kv.first.c_str(),
                  static_cast<unsigned long long>(kv.second.count), 1e3 * kv.second.sum / static_cast<double>(kv.second.count));
      first_t = false;
    }
    std::printf("}}\n");
    synth_destroy(scene);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "aw_demo: %s\n", e.what());
    return 1;
  }
  return 0;
}
