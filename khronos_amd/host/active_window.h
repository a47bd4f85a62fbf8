// active_window.h — host-side mirror of the reference's plugin surface for the fusion path:
//   khronos::ActiveWindow                     khronos/include/khronos/active_window/active_window.h:67-193
//   khronos::TrackingIntegrator::Config       .../integration/tracking_integrator.h:59-83
//   khronos::MotionDetector / FreeSpaceMotionDetector   .../motion_detection/*.h
//   khronos::ObjectDetector, Tracker (no-op bases)      .../object_detection/object_detector.h, tracking/tracker.h
//   khronos::MeshObjectExtractor (static objects)       .../object_extraction/mesh_object_extractor.h:59-165
//   khronos::FrameData / FrameDataBuffer / Track        .../data/*.h
// Same class names, config keys, method names and call order; the volumetric work goes through the C ABI in
// include/khronos_amd.h to the gfx950 kernels, the map lives in HBM.  Hydra types are the stand-ins of
// hydra_compat.h.  Config errors throw std::invalid_argument (the reference aborts in config::checkValid).
#pragma once
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "hydra_compat.h"
#include "mini_yaml.h"

namespace khronos {

using hydra::BlockIndex;
using hydra::BlockIndices;
using hydra::BoundingBox;
using hydra::fromSeconds;
using hydra::InputData;
using hydra::KhronosObjectAttributes;
using hydra::TimeStamp;
using hydra::toSeconds;
using hydra::VolumetricMap;

// ---- data (khronos/include/khronos/active_window/data/) -------------------------------------------------
using GlobalIndex = std::array<int64_t, 3>;  // spatial_hash::GlobalIndex role

struct SemanticClusterInfo {  // measurement_clusters.h:48-58
  int category_id = -1;
  std::vector<float> feature = {0.f};  // FeatureVector::Zero(1, 1) = "no open-set feature"
  explicit SemanticClusterInfo(int category) : category_id(category) {}
  SemanticClusterInfo(int category, std::vector<float> f) : category_id(category), feature(std::move(f)) {}
};

struct MeasurementCluster {  // measurement_clusters.h:63-80
  int id = 0;
  size_t num_pixels = 0;            // pixels.size(); the pixel list itself stays in the device id image
  BoundingBox bounding_box;
  float centroid[3] = {0, 0, 0};    // utils::computeCentroid of the cluster's vertices
  std::vector<GlobalIndex> voxels;  // GlobalIndexSet role, kept sorted by (x, y, z)
  std::optional<SemanticClusterInfo> semantics;
};

struct FrameData {  // frame_data.h:59-83
  using Ptr = std::shared_ptr<FrameData>;
  InputData input;
  std::vector<MeasurementCluster> dynamic_clusters;
  std::vector<MeasurementCluster> semantic_clusters;
  // dynamic_image / object_image live in the device frame slot `input.slot`; host copies on demand
  std::vector<int32_t> dynamicImage() const;
  int num_dynamic_clusters = 0;
};

struct Observation {  // track.h
  TimeStamp stamp = 0;
  int semantic_cluster_id = -1;
  int dynamic_cluster_id = -1;
};

struct Track {  // track.h:72-111
  int id = 0;
  bool is_active = true;
  bool is_dynamic = false;
  float confidence = 0.f;
  TimeStamp first_seen = 0, last_seen = 0;
  BoundingBox last_bounding_box;
  std::vector<GlobalIndex> last_voxels;  // sorted
  float last_voxel_size = 0.f;
  float last_centroid[3] = {0, 0, 0};
  std::optional<SemanticClusterInfo> semantics;
  size_t num_features = 0;
  std::vector<Observation> observations;
  // track_by = pixels: Track::last_points (track.h) are the vertices of the last observation's pixels.  They are not
  // copied out of HBM: the track names the id image they sit in and keeps that frame slot alive.
  struct PixelRef {
    khr_ctx* ctx = nullptr;
    int slot = -1, which = 0, id = 0;  // which: 0 dynamic image, 1 object image
    size_t num_points = 0;             // last_points.size()
    std::shared_ptr<void> lease;       // khr_retain_slot for as long as any copy of the track refers to the frame
  } last_pixels;
  void updateSemantics(const std::optional<SemanticClusterInfo>& other);  // track.cpp:42-70
};
using Tracks = std::vector<Track>;

class FrameDataBuffer {  // frame_data_buffer.h:52-100, frame_data_buffer.cpp:57-123
 public:
  struct Config {
    size_t max_buffer_size = 300;
    int store_every_n_frames = 1;
  } const config;
  explicit FrameDataBuffer(const Config& config);
  void trimBuffer(const Tracks& tracks);
  void storeData(const FrameData::Ptr& data);
  FrameData::Ptr getData(TimeStamp stamp) const;
  const FrameData& getLatestData() const { return *buffer_.back(); }
  size_t size() const { return buffer_.size(); }
  bool empty() const { return buffer_.empty(); }
  void clear() { buffer_.clear(); }

 private:
  std::deque<FrameData::Ptr> buffer_;
  int input_counter_ = 0;
  TimeStamp oldest_time_stamp_ = 0;
};

// ---- processors ---------------------------------------------------------------------------------------------
struct TrackingIntegrator {
  struct Config {  // tracking_integrator.h:59-83, checks tracking_integrator.cpp:61-65
    int verbosity = 0;
    float temporal_buffer = 1.f;
    float burn_in_period = 1.f;
    float tsdf_occupancy_threshold = -1.5f;
    int neighbor_connectivity = 18;
    float temporal_window = 3.f;
    int num_threads = -1;  // accepted for compatibility; the update is one kernel launch
  };
};

class MotionDetector {  // motion_detector.h:49-69: the base class is a no-op
 public:
  virtual ~MotionDetector() = default;
  virtual void processInput(const VolumetricMap& /*map*/, FrameData& /*data*/) {}
  virtual bool isDeviceBacked() const { return false; }
};

class FreeSpaceMotionDetector : public MotionDetector {  // free_space_motion_detector.h:72-197
 public:
  struct Config {
    int verbosity = 0;
    int neighbor_connectivity = 26;
    int min_cluster_size = 0;
    int max_cluster_size = 1000000;
    float min_separation_distance = 1.f;
    float max_range = 10000.f;
    float min_z_coordinate = -10000.f;
    int num_threads = -1;
  } const config;
  explicit FreeSpaceMotionDetector(const Config& config);
  void processInput(const VolumetricMap& map, FrameData& data) override;
  bool isDeviceBacked() const override { return true; }
  // FrameData::dynamic_clusters from the device (writeClustersToData, free_space_motion_detector.cpp:381-399)
  static void fetchClusters(const VolumetricMap& map, FrameData& data);
};

class ObjectDetector {  // object_detector.h: no-op base
 public:
  virtual ~ObjectDetector() = default;
  virtual void processInput(const VolumetricMap& /*map*/, FrameData& /*data*/) {}
};

class Tracker {  // tracker.h:49-69: no-op base that only owns the track list
 public:
  virtual ~Tracker() = default;
  virtual void processInput(FrameData& /*data*/) {}
  Tracks& getTracks() { return tracks_; }
  const Tracks& getTracks() const { return tracks_; }

 protected:
  Tracks tracks_;
};

// khronos::ConnectedSemantics (connected_semantics.h:59-164): the clustering runs on the device
// (khr_detect_objects), the class carries the config and fills FrameData::semantic_clusters
class ConnectedSemantics : public ObjectDetector {
 public:
  struct Config {  // connected_semantics.h:62-84, declare_config connected_semantics.cpp:44-54
    int verbosity = 0;
    bool use_full_connectivity = true;
    int min_cluster_size = 0;
    int max_cluster_size = -1;
    bool use_3d = true;
    float grid_size = 0.1f;
    float max_range = 0.f;
    // hydra's label space (GlobalInfo::getLabelSpaceConfig().isObject, connected_semantics.cpp:134,157)
    std::vector<int> object_labels;
    static Config fromYaml(const khronos_amd::YamlNode& node);
  } const config;
  ConnectedSemantics(const Config& config, const VolumetricMap& map);
  void processInput(const VolumetricMap& map, FrameData& data) override;
};

// khronos::InstanceForwarding (instance_forwarding.h:57-100, instance_forwarding.cpp:73-149): the instance ids of the label
// image ARE the clusters.  Pixel grouping and the per-cluster summaries are one device pass (khr_forward_instances); the
// background filter scores each id's open-set feature against the background prompts on the host (a handful of ids).
class InstanceForwarding : public ObjectDetector {
 public:
  struct Config {
    int verbosity = 0;
    float max_range = 0.f;
    int min_cluster_size = 0;
    int max_cluster_size = -1;
    double min_object_volume = 0.0;
    double max_object_volume = -1.0;
    double max_background_score = 0.2;
    // hydra::EmbeddingGroup `background` (prompt embeddings) + hydra::CosineDistance metric: un-vendored, given here as
    // plain vectors; empty = no background filter (instance_forwarding.cpp:66-71)
    std::vector<std::vector<float>> background_embeddings;
    int max_instance_id = 4095;  // size of the device's per-id table (extension)
    static Config fromYaml(const khronos_amd::YamlNode& node);
  } const config;
  explicit InstanceForwarding(const Config& config);
  void processInput(const VolumetricMap& map, FrameData& data) override;
  static float bestBackgroundScore(const std::vector<std::vector<float>>& background, const std::vector<float>& feature);

 private:
  bool filter_by_volume_;
};

// khronos::MaxIoUTracker (max_iou_tracker.h:60-214, max_iou_tracker.cpp).  The per-cluster voxel sets come from
// the device (khr_cluster_voxels); the association logic is host code as in the reference.
class MaxIoUTracker : public Tracker {
 public:
  struct Config {  // max_iou_tracker.h:63-101, checks max_iou_tracker.cpp:143-147
    int verbosity = 0;
    enum class SemanticAssociation { kAssignCluster, kAssignTrack } semantic_association = SemanticAssociation::kAssignCluster;
    float min_semantic_iou = 0.5f;
    float min_cosine_sim = 0.0f;
    float min_cross_iou = 0.5f;
    float max_dynamic_distance = 1.f;
    float temporal_window = 3.f;
    int min_num_observations = 20;
    enum class TrackBy { kPixels, kVoxels, kBouningBox } track_by = TrackBy::kPixels;
    float voxel_size = 0.1f;
    static Config fromYaml(const khronos_amd::YamlNode& node);
  } const config;
  explicit MaxIoUTracker(const Config& config);
  void processInput(FrameData& data) override;

  // processInput in two halves for callers that queue the next frame's device work in between (object_pipeline.cpp):
  // beginInput enqueues the voxel-set passes, completeInput waits for them and runs the association
  void beginInput(FrameData& data);
  void completeInput(FrameData& data);

  // the steps of processInput (public as in the reference)
  void setupTrackMeasurements(FrameData& data) const;
  void launchTrackMeasurements(FrameData& data) const;
  void finishTrackMeasurements(FrameData& data) const;
  void associateTracks(const FrameData& data);
  void associateDynamicTracks(const FrameData& data);
  void associateSemanticTracks(const FrameData& data);
  void assignClustersToStaticTrack(const FrameData& data, std::unordered_set<int>& associated_objects);
  void assignStaticTracksToCluster(const FrameData& data, std::unordered_set<int>& associated_objects);
  void updateTrackingDuration();
  Track& addNewTrack(const MeasurementCluster& observation, bool is_dynamic);
  void updateTrack(const MeasurementCluster& observation, Track& track, bool is_observation_dynamic) const;
  void computeCentroid(const MeasurementCluster& cluster, float* centroid) const;
  float computeIoU(const MeasurementCluster& cluster, const Track& track) const;
  static float computeIoUVoxels(const std::vector<GlobalIndex>& cluster_voxels, const std::vector<GlobalIndex>& track_voxels);
  static float computeIoUBoundingBox(const BoundingBox& a, const BoundingBox& b);

 private:
  TimeStamp processing_stamp_ = 0;
  int current_track_id_ = 0;
  mutable std::vector<int32_t> scratch_ids_;
  mutable std::vector<int64_t> scratch_voxels_;
  // track_by = pixels: the frame being associated and, per track (index), the intersections of its re-projected points
  // with every object-image cluster of that frame (khr_pixel_iou)
  const FrameData* current_ = nullptr;
  mutable std::vector<std::vector<uint32_t>> pix_inter_;
  mutable std::vector<uint64_t> pix_key_;  // (track id, last_seen) the row was computed for
  mutable int pix_max_id_ = 0;
  void ensurePixelIntersections(size_t track_index) const;
};

// khronos::ExternalTracker (external_tracker.cpp:59-143): tracks follow externally provided cluster ids
class ExternalTracker : public Tracker {
 public:
  struct Config {
    int verbosity = 0;
    float temporal_window = 3.f;
    int min_num_observations = 20;
    static Config fromYaml(const khronos_amd::YamlNode& node);
  } const config;
  explicit ExternalTracker(const Config& config);
  void processInput(FrameData& data) override;

 private:
  TimeStamp processing_stamp_ = 0;
};

class ObjectExtractor {  // object_extractor.h
 public:
  virtual ~ObjectExtractor() = default;
  virtual std::shared_ptr<KhronosObjectAttributes> extractObject(const Track& track, const FrameDataBuffer& frames) = 0;
  // called once on the thread that is going to call extractObject, before its first request (device runtimes set up per-thread
  // state at a thread's first call: not beside the window's frames)
  virtual void prepareThread() {}
};

class MeshObjectExtractor : public ObjectExtractor {  // mesh_object_extractor.h:59-165
 public:
  struct Config {
    int verbosity = 0;
    float min_object_allocation_confidence = 0.5f;
    float min_object_volume = 0.1f;
    float max_object_volume = 4.0f;
    bool only_extract_reconstructed_objects = false;
    float min_dynamic_displacement = 0.2f;
    float min_object_reconstruction_confidence = 0.5f;
    float min_object_reconstruction_observations = 10.f;
    float object_reconstruction_resolution = -0.02f;
    float min_reconstruction_resolution = 0.f;
    bool visualize_classification = false;
    // the extractor's OWN integrators (mesh_object_extractor.h:87-91, used at mesh_object_extractor.cpp:239,267; uHumans2.yaml:99-100
    // sets their thread counts): the object maps are integrated and meshed with THESE settings, not with the window's
    struct ProjectiveIntegrator {
      int verbosity = 0;
      bool use_weight_dropoff = true;
      float weight_dropoff_epsilon = -1.f;
      bool use_constant_weight = false;
      float max_weight = 1e5f;
      std::string interpolation_method = "adaptive";
      int num_threads = -1;
      std::string color_blend_weight = "post";  // [A] switch (khr_config.color_blend_weight): "post" | "pre"
    } projective_integrator;
    struct MeshIntegrator {
      float min_weight = 1e-4f;
      std::string attr_source = "nearest";  // [A] switch (khr_config.mesh_attr_source): "nearest" | "containing"
      float degenerate_eps = 1e-6f;         // [A] switch (khr_config.mesh_degenerate_eps)
    } mesh_integrator;
    uint32_t max_object_blocks = 32768;  // device pool of the object mini-map
  } const config;
  MeshObjectExtractor(const Config& config, const khr_config& aw_device_config);
  ~MeshObjectExtractor() override;
  MeshObjectExtractor(const MeshObjectExtractor&) = delete;
  MeshObjectExtractor& operator=(const MeshObjectExtractor&) = delete;
  std::shared_ptr<KhronosObjectAttributes> extractObject(const Track& track, const FrameDataBuffer& frames) override;
  void prepareThread() override {
    if (object_ctx_) khr_sync(object_ctx_);
  }
  // MeshObjectExtractor::extractDynamicObject (mesh_object_extractor.cpp:120-172): trajectory summary
  std::shared_ptr<KhronosObjectAttributes> extractDynamicObject(const Track& track, const FrameDataBuffer& frames) const;
  // a13: MeshObjectExtractor::extractStaticObject (mesh_object_extractor.cpp:174-304)
  std::shared_ptr<KhronosObjectAttributes> extractStaticObject(const Track& track, const FrameDataBuffer& frames) const;
  // sizing of the private map (mesh_object_extractor.cpp:201-228): voxel size and block range
  static float objectVoxelSize(const Config& config, const BoundingBox& extent);
  static void objectBlockRange(const BoundingBox& extent, float block_size, int32_t* min_idx, int32_t* max_idx);

 private:
  khr_config device_config_;
  // the object mini-map is one device context that is emptied and re-scaled per object (khr_reset_map) instead of a
  // new VolumetricMap per object: creating / destroying an HBM pool costs milliseconds, the reset one small kernel
  khr_config objectMapConfig(float voxel_size) const;
  mutable khr_ctx* object_ctx_ = nullptr;
  mutable uint32_t object_ctx_blocks_ = 0;
};

// ---- the module -------------------------------------------------------------------------------------------------
// khronos::ObjectWorkerPool (object_worker_pool.h / .cpp:56-146): tracks that left the window are submitted together with
// a copy of the frame buffer (the shared frames stay alive, active_window.cpp:261-263) and extracted on worker threads;
// finished objects are collected with fill().  The reference starts one detached thread per request, at most
// `num_workers` at a time, all sharing one extractor; here every worker owns its extractor (= its own device mini-map
// context and stream), because an extraction re-integrates into that context.
class ObjectWorkerPool {
 public:
  struct Config {
    int num_workers = 2;      // 0 = as many as the reference would start: capped at kMaxWorkers here
    int poll_time_us = 1000;  // (the reference polls its queue; this pool blocks on a condition variable)
    int verbosity = 0;
  };
  static constexpr int kMaxWorkers = 4;
  using ExtractorFactory = std::function<std::unique_ptr<ObjectExtractor>()>;

  ObjectWorkerPool(const Config& config, const ExtractorFactory& make_extractor);
  ~ObjectWorkerPool();
  void stop();
  void join();                 // wait until every submitted request has been worked off
  size_t numRunning() const;   // requests queued or in work
  void submit(TimeStamp stamp, Track&& track, const FrameDataBuffer& frame_data);
  std::shared_ptr<KhronosObjectAttributes> runBlocking(const Track& track, const FrameDataBuffer& data);
  // moves the finished objects (completion order) to `out`; returns how many
  size_t fill(std::vector<std::shared_ptr<KhronosObjectAttributes>>& out);
  bool hasExtractor() const { return !extractors_.empty(); }
  const Config config;

 private:
  struct Request {
    TimeStamp stamp;
    Track track;
    FrameDataBuffer frame_data;
  };
  void workerLoop(size_t worker);
  std::vector<std::unique_ptr<ObjectExtractor>> extractors_;  // [0] also serves runBlocking (under blocking_mutex_)
  std::vector<std::thread> workers_;
  mutable std::mutex mutex_;
  std::mutex blocking_mutex_;
  std::condition_variable cv_work_, cv_idle_;
  std::deque<std::unique_ptr<Request>> queue_;
  std::vector<std::shared_ptr<KhronosObjectAttributes>> output_;
  std::string error_;
  size_t in_work_ = 0;
  std::atomic<size_t> outstanding_{0};  // queued + running requests (join() polls it before it sleeps on cv_idle_)
  bool should_shutdown_ = false;
  // env KHR_TEST_EXTRACT_DELAY_MS: every extraction starts this much later (tests of the frame-ring back-pressure)
  int test_delay_ms_ = std::getenv("KHR_TEST_EXTRACT_DELAY_MS") ? std::atoi(std::getenv("KHR_TEST_EXTRACT_DELAY_MS")) : 0;
};

class ActiveWindow : public hydra::ActiveWindowModule {  // active_window.h:67
 public:
  using OutputQueue = hydra::ActiveWindowModule::OutputQueue;
  using KhronosSink = std::function<void(const FrameData&, const VolumetricMap&, const Tracks&)>;

  struct Config {  // active_window.h:72-96 + declare_config active_window.cpp:50-71
    int verbosity = 0;
    float min_output_separation = 0.0f;
    bool detach_object_extraction = true;
    VolumetricMap::Config volumetric_map;  // base hydra::ActiveWindowModule::Config(false, true)
    struct ProjectiveIntegrator {
      int verbosity = 0;
      bool use_weight_dropoff = true;
      float weight_dropoff_epsilon = -1.f;
      bool use_constant_weight = false;
      float max_weight = 1e5f;
      std::string interpolation_method = "adaptive";
      int num_threads = -1;
      float label_confidence = 0.9f;
      // ASSUMPTIONS.md [A] choices of the absent upstream integrator as switches (khr_config.alloc_candidate / color_blend_weight;
      // not keys of the reference: INTEGRATION.md 3a)
      std::string alloc_candidate = "block_centre";  // "block_centre" | "camera_offset"
      std::string color_blend_weight = "post";       // "post" | "pre"
    } projective_integrator;
    TrackingIntegrator::Config tracking_integrator;
    std::string motion_detector_type;    // "" = none, "FreeSpaceMotionDetector"
    FreeSpaceMotionDetector::Config motion_detector;
    std::string object_detector_type;    // "" = none, "ConnectedSemantics", "InstanceForwarding"
    ConnectedSemantics::Config object_detector;
    InstanceForwarding::Config instance_forwarding;  // read from the same `object_detector` node
    std::string tracker_type;            // "" = none, "MaxIouTracker", "ExternalTracker"
    MaxIoUTracker::Config tracker;
    std::string object_extractor_type;   // "" = none, "MeshObjectExtractor"
    MeshObjectExtractor::Config object_extractor;
    ObjectWorkerPool::Config extraction_worker;
    struct MeshIntegrator {
      float min_weight = 1e-4f;
      std::string attr_source = "nearest";  // [A] switch (khr_config.mesh_attr_source): "nearest" | "containing"
      float degenerate_eps = 1e-6f;         // [A] switch (khr_config.mesh_degenerate_eps)
    } mesh_integrator;
    FrameDataBuffer::Config frame_data_buffer;
    // `khronos_sinks:` (active_window.cpp:70): a list of {type: <registered sink type>, ...} mappings; each is instantiated by
    // the factory registered under its type (registerKhronosSink; config_utilities' RegistrationWithConfig role) at
    // construction (active_window.cpp:80).  Sinks can still be added later with addKhronosSink (khronos_pipeline.cpp:94-104).
    std::vector<khronos_amd::YamlNode> khronos_sinks;
    // device-side sizing (no reference equivalent)
    int num_labels = 20;
    uint32_t max_blocks = 16384;
    int frame_slot_headroom = -1;  // device frame-ring slots beyond max_buffer_size + 1 (-1 = 16 per extraction worker): frames that
                                   // pending extraction requests still hold after the window dropped them
    uint32_t max_snapshot_blocks = 8192;  // capacity of an output's map snapshot (cloneUpdated): ~100 KB of HBM per block
    uint32_t max_frame_pixels = 1280 * 720;
    uint64_t max_mesh_vertices = 8u << 20;
    int device = 0, rank = 0, world_size = 1;
    int exact_arithmetic = 1;  // khr_config.exact_arithmetic: 1 = voxel values bit-identical to the CPU restatement (default), 0 = relaxed values
    // hydra::timing scopes (hydra_compat.h): wait for the device before a scope that launched device work stops, so that
    // "active_window/all" is the per-frame latency the reference's timer measures (off: enqueue time only)
    bool timing_sync_device = false;
    // one khr_process_frame call per frame carries the object detector's kernels and, at output frames without sinks, the
    // output's device stages (false: every stage where the reference's spinOnce has it, one call each)
    bool fuse_device_stages = true;

    // parse the `active_window:` mapping of a Khronos mapper YAML (same keys as uHumans2.yaml:35-100)
    static Config fromYaml(const khronos_amd::YamlNode& active_window_node);
    static Config fromYamlString(const std::string& yaml_text);
    void checkValid() const;  // throws std::invalid_argument
  } const config;

  // active_window.h:99: ActiveWindow(const Config&, const OutputQueue::Ptr&); registered with the string factory under
  // "ActiveWindow" (active_window.h:190-192; active_window.cpp below) -- `type: "ActiveWindow"` in the mapper YAML
  ActiveWindow(const Config& config, const OutputQueue::Ptr& output_queue);
  explicit ActiveWindow(const Config& config) : ActiveWindow(config, nullptr) {}  // (drivers that read spinOnce's return value via step())
  ~ActiveWindow() override;

  std::string printInfo() const override;
  // access (not thread-safe, as in the reference)
  VolumetricMap& getMap() { return map_; }
  const VolumetricMap& getMap() const { return map_; }
  const FrameData& getLatestFrameData() const {
    completePendingFrame();
    return frame_data_buffer_.getLatestData();
  }
  const Tracks& getTracks() const {
    completePendingFrame();
    return tracker_->getTracks();
  }
  void addKhronosSink(const KhronosSink& sink);
  // factory registry behind the `khronos_sinks` config key: type name -> factory(config mapping of the list entry).  An entry
  // whose type nobody registered is reported and skipped (config_utilities logs "cannot create" and hands back nullptr,
  // which KhronosSink::instantiate drops).  Returns false when the name is already taken.
  using KhronosSinkFactory = std::function<KhronosSink(const khronos_amd::YamlNode&)>;
  static bool registerKhronosSink(const std::string& type, KhronosSinkFactory factory);
  size_t numKhronosSinks() const { return sinks_.size(); }
  void setObjectDetector(std::unique_ptr<ObjectDetector> d) { object_detector_ = std::move(d); }
  void setTracker(std::unique_ptr<Tracker> t) {
    completePendingFrame();
    tracker_ = std::move(t);
  }

  // times spinOnce had to wait for the extraction worker because every device frame slot was leased (diagnostics)
  size_t numRingWaits() const { return num_ring_waits_; }
  // FrameData::dynamic_clusters.size() of the frame spinOnce processed last (without touching the frame buffer)
  int numDynamicClustersOfLastFrame() const { return last_num_dynamic_; }
  // wait for the detached object extractions queued so far (object_worker_pool.cpp:115-146); a benchmark's clock stops behind this
  void joinExtractions() {
    if (extraction_worker_) extraction_worker_->join();
  }
  void finishMapping();
  std::vector<std::shared_ptr<KhronosObjectAttributes>> extractObjects();


 protected:
  // called by the module thread (hydra::ActiveWindowModule::step here); active_window.h:134
  hydra::ActiveWindowOutput::Ptr spinOnce(const hydra::InputPacket& input) override;

  std::shared_ptr<FrameData> createData(const hydra::InputPacket& input) const;
  void updateMap(const FrameData& data);
  // device_stages_queued: mesh, snapshot, archival and flag clearing were queued by this frame's khr_process_frame call
  hydra::ActiveWindowOutput::Ptr extractOutputData(const FrameData& data, bool threaded, bool device_stages_queued = false);
  void extractInactiveObjects();
  // second half of the previous frame's tracker step (association, trimBuffer, storeData) when spinOnce deferred it
  void completePendingFrame() const;

  khr_ctx* ctx_ = nullptr;
  khr_config device_config_{};
  VolumetricMap map_;
  std::unique_ptr<MotionDetector> motion_detector_;
  std::unique_ptr<ObjectDetector> object_detector_;
  std::unique_ptr<Tracker> tracker_;
  std::unique_ptr<ObjectWorkerPool> extraction_worker_;  // owns the extractor(s) (active_window.h:189)
  std::mutex mutex_;
  std::vector<KhronosSink> sinks_;
  mutable FrameDataBuffer frame_data_buffer_;
  mutable std::shared_ptr<FrameData> pending_frame_;  // queued voxel-set passes, association outstanding (completePendingFrame)
  TimeStamp latest_stamp_ = 0;
  TimeStamp last_full_upated_ = 0;
  size_t num_frames_processed_ = 0;
  mutable size_t num_ring_waits_ = 0;
  int last_num_dynamic_ = 0;
  std::vector<int32_t> removed_scratch_;  // extractOutputData: the archived blocks' indices on their way into the output

  // active_window.h:190-192: `active_window: {type: "ActiveWindow", ...}` in the mapper YAML creates this class
  inline static const hydra::ActiveWindowRegistration<ActiveWindow> registration_{"ActiveWindow"};
};

}  // namespace khronos
