// hydra_compat.h — minimal stand-ins for the Hydra / spark_dsg types that appear in the signature of
// khronos::ActiveWindow (khronos/include/khronos/active_window/active_window.h:67-193).  Hydra is an
// un-vendored dependency of the reference (install/https.rosinstall:5-8) and is not available offline; in
// a real integration these are the genuine Hydra types and this header disappears.
#pragma once
#include <chrono>
#include <cmath>
#include <fstream>
#include <functional>
#include <map>
#include <mutex>
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/khronos_amd.h"

namespace hydra {

using TimeStamp = uint64_t;
inline double toSeconds(TimeStamp t) { return static_cast<double>(t) / 1e9; }
inline TimeStamp fromSeconds(double s) { return static_cast<TimeStamp>(s * 1e9); }

using BlockIndex = std::array<int32_t, 3>;
using BlockIndices = std::vector<BlockIndex>;

struct Sensor {
  int width = 0, height = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  float min_range = 0.1f, max_range = 5.f;
};

// hydra::InputPacket role: raw sensor frame + pose.  Row-major 4x4 doubles (Eigen::Isometry3d role).
struct InputPacket {
  TimeStamp timestamp_ns = 0;
  double world_T_body[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double body_T_sensor[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  Sensor sensor;
  const float* depth = nullptr;     // H*W metres
  const uint8_t* color = nullptr;   // H*W*3
  const int32_t* labels = nullptr;  // H*W
  bool on_device = false;           // buffers are HBM-resident on the context's device
  bool buffers_complete = false;    // on_device: no stream is still writing the buffers when the packet is handed over (KHR_PF_INPUT_READY:
                                    // the conversion may then run on the context's second stream, beside the previous frame's tail)
  // open-set features of the instance ids of `labels` (InputData::label_features, used at instance_forwarding.cpp:96,141)
  std::map<int, std::vector<float>> label_features;
};

inline void mul4(const double* a, const double* b, double* o) {
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[4 * r + k] * b[4 * k + c];
      o[4 * r + c] = s;
    }
}

// hydra::InputData role (fields SURVEY.md A.2).  The normalised images live in a device frame slot of the
// fusion context; host copies are fetched on demand.
struct InputData {
  TimeStamp timestamp_ns = 0;
  double world_T_body[16];
  double world_T_sensor[16];
  Sensor sensor;
  khr_ctx* ctx = nullptr;
  int slot = -1;  // device frame slot
  std::map<int, std::vector<float>> label_features;  // InputData::label_features
  // keeps the slot out of the ring for as long as any copy of this InputData lives (the shared_ptr<FrameData> ownership of
  // the reference: buffer entries and extraction workers keep frames alive, active_window.cpp:261-263)
  std::shared_ptr<void> slot_lease;
  void retainSlot() {
    if (!ctx || slot < 0 || khr_retain_slot(ctx, slot) < 0) return;
    khr_ctx* c = ctx;
    const int s = slot;
    slot_lease = std::shared_ptr<void>(nullptr, [c, s](void*) { khr_release_slot(c, s); });
  }
  // ActiveWindowOutput::sensor_data (active_window.cpp:165): the output's copy does not lease the ring slot -- outputs wait in a
  // consumer's queue for an unbounded time -- it owns a device-side copy of the frame's images instead (khr_frame_copy, taken in
  // stream order when the output is built) and fetches from that
  std::shared_ptr<khr_frame_copy> images;
  void detachFromRing() {
    if (ctx && slot >= 0) {
      khr_frame_copy* fc = nullptr;
      if (khr_frame_copy_create(ctx, slot, &fc) == 0 && fc) images = std::shared_ptr<khr_frame_copy>(fc, [](khr_frame_copy* p) { khr_frame_copy_release(p); });
    }
    slot_lease.reset();
    slot = -1;
    ctx = nullptr;
  }
  const Sensor& getSensor() const { return sensor; }
  const double* getSensorPose() const { return world_T_sensor; }
  size_t numPixels() const { return static_cast<size_t>(sensor.width) * sensor.height; }
  // host copies of the normalised images (InputData::range_image, ::vertex_map, ::depth_image, ::color_image, ::label_image); empty
  // when neither a slot nor a copy is held
  std::vector<float> rangeImage() const {
    std::vector<float> r(numPixels());
    if (ctx && slot >= 0) khr_download_frame(ctx, slot, r.data(), nullptr, nullptr);
    else if (!images || khr_frame_copy_download(images.get(), nullptr, r.data(), nullptr, nullptr, nullptr) != 0) r.clear();
    return r;
  }
  std::vector<float> vertexMap() const {
    std::vector<float> v(numPixels() * 3);
    if (ctx && slot >= 0) khr_download_frame(ctx, slot, nullptr, v.data(), nullptr);
    else if (!images || khr_frame_copy_download(images.get(), nullptr, nullptr, nullptr, nullptr, v.data()) != 0) v.clear();
    return v;
  }
  std::vector<float> depthImage() const {
    std::vector<float> d(numPixels());
    if (!images || khr_frame_copy_download(images.get(), d.data(), nullptr, nullptr, nullptr, nullptr) != 0) d.clear();
    return d;
  }
  std::vector<uint8_t> colorImage() const {  // rgb8
    std::vector<uint8_t> c(numPixels() * 3);
    if (!images || khr_frame_copy_download(images.get(), nullptr, nullptr, c.data(), nullptr, nullptr) != 0) c.clear();
    return c;
  }
  std::vector<int32_t> labelImage() const {
    std::vector<int32_t> l(numPixels());
    if (!images || khr_frame_copy_download(images.get(), nullptr, nullptr, nullptr, l.data(), nullptr) != 0) l.clear();
    return l;
  }
};

// a voxel block copied to the host (VolumetricMap::cloneUpdated role, active_window.cpp:229)
struct BlockCopy {
  BlockIndex index;
  std::vector<float> distance, weight;
  std::vector<uint8_t> color;  // rgba
  std::vector<uint64_t> last_observed, last_occupied;
  std::vector<uint8_t> flags;
  std::vector<uint32_t> semantic_label;
  uint8_t block_flags = 0;
};

// hydra::VolumetricMap role: here a handle on the HBM-resident map of a fusion context.
class VolumetricMap {
 public:
  struct Config {
    float voxel_size = 0.1f;
    int voxels_per_side = 16;
    float truncation_distance = 0.3f;
    bool with_semantics = false;
    bool with_tracking = true;
  } config;

  VolumetricMap() = default;
  VolumetricMap(const Config& cfg, khr_ctx* ctx) : config(cfg), ctx_(ctx) {}
  bool hasSemantics() const { return config.with_semantics; }
  float blockSize() const { return config.voxel_size * static_cast<float>(config.voxels_per_side); }
  khr_ctx* ctx() const { return ctx_; }
  size_t numBlocks() const { return static_cast<size_t>(khr_num_blocks(ctx_)); }
  // TsdfLayer::allocatedBlockIndices / blockIndicesWithCondition(updated) role (sorted)
  BlockIndices allocatedBlockIndices(bool only_updated = false) const {
    const int64_t n = khr_block_indices(ctx_, nullptr, 0, only_updated);
    BlockIndices out(static_cast<size_t>(n > 0 ? n : 0));
    if (n > 0) khr_block_indices(ctx_, out[0].data(), n, only_updated);
    return out;
  }
  // deep copy of one block of the LIVE map (visualiser / tests; the output's snapshot is ActiveWindowOutput::cloneUpdated)
  BlockCopy cloneBlock(const BlockIndex& idx) const {
    BlockCopy b;
    b.index = idx;
    const size_t n = static_cast<size_t>(config.voxels_per_side) * config.voxels_per_side * config.voxels_per_side;
    b.distance.resize(n); b.weight.resize(n); b.color.resize(4 * n); b.last_observed.resize(n);
    b.last_occupied.resize(n); b.flags.resize(n); b.semantic_label.resize(n);
    khr_download_block(ctx_, idx[0], idx[1], idx[2], b.distance.data(), b.weight.data(), b.color.data(),
                       b.last_observed.data(), b.last_occupied.data(), b.flags.data(), b.semantic_label.data(), nullptr,
                       &b.block_flags);
    return b;
  }

 private:
  khr_ctx* ctx_ = nullptr;
};

struct Mesh {
  std::vector<float> points;    // 3 per vertex
  std::vector<uint8_t> colors;  // rgba per vertex
  std::vector<uint32_t> labels;
  std::vector<uint64_t> first_seen_stamps, stamps;
  size_t numVertices() const { return labels.size(); }
};

struct BoundingBox {
  float min[3] = {0, 0, 0}, max[3] = {0, 0, 0};
  bool valid = false;
  void merge(const BoundingBox& o) {
    if (!o.valid) return;
    if (!valid) { *this = o; return; }
    for (int i = 0; i < 3; ++i) { min[i] = o.min[i] < min[i] ? o.min[i] : min[i]; max[i] = o.max[i] > max[i] ? o.max[i] : max[i]; }
  }
  void include(const float* p) {
    if (!valid) { for (int i = 0; i < 3; ++i) min[i] = max[i] = p[i]; valid = true; return; }
    for (int i = 0; i < 3; ++i) { if (p[i] < min[i]) min[i] = p[i]; if (p[i] > max[i]) max[i] = p[i]; }
  }
  float dimension(int i) const { return max[i] - min[i]; }
  float center(int i) const { return 0.5f * (min[i] + max[i]); }
  float volume() const { return valid ? dimension(0) * dimension(1) * dimension(2) : 0.f; }
  float maxDimension() const { float m = dimension(0); for (int i = 1; i < 3; ++i) if (dimension(i) > m) m = dimension(i); return m; }
};

// spark_dsg::KhronosObjectAttributes role (fields set at mesh_object_extractor.cpp:81-118,268-302)
struct KhronosObjectAttributes {
  Mesh mesh;
  BoundingBox bounding_box;
  int semantic_label = -1;
  std::vector<float> semantic_feature;
  std::vector<TimeStamp> first_observed_ns, last_observed_ns;
  double position[3] = {0, 0, 0};
  // dynamic objects (mesh_object_extractor.cpp:120-172)
  std::vector<std::array<float, 3>> trajectory_positions;
  std::vector<TimeStamp> trajectory_timestamps;
};

// hydra::ActiveWindowOutput role (fields set at active_window.cpp:225-247)
struct ActiveWindowOutput {
  using Ptr = std::shared_ptr<ActiveWindowOutput>;
  TimeStamp timestamp_ns = 0;
  double world_t_body[3] = {0, 0, 0};
  double world_R_body[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  BlockIndices archived_mesh_indices;
  // setMap(map.cloneUpdated()) (active_window.cpp:229): a device-side SNAPSHOT of the blocks flagged updated, taken between
  // meshing and archival (khr_snapshot_updated).  It keeps its contents whatever later frames do to the map -- the hydra
  // frontend takes outputs from a queue -- and costs no host round trip at output time; indices and voxels come to the
  // host when a consumer asks.  Released with the last copy of the output.
  std::shared_ptr<khr_snapshot> map;
  khr_ctx* map_ctx = nullptr;  // the live map the snapshot was taken from (voxels_per_side etc. via khr_get_config)
  std::shared_ptr<InputData> sensor_data;
  std::vector<std::shared_ptr<KhronosObjectAttributes>> graph_update;  // LayerUpdate(2) role

  void setMap(khr_snapshot* snap) { map = std::shared_ptr<khr_snapshot>(snap, [](khr_snapshot* s) { khr_snapshot_release(s); }); }
  // indices of the snapshot's blocks, sorted (TsdfLayer::allocatedBlockIndices of the cloned map)
  const BlockIndices& updatedBlocks() const {
    if (!indices_valid_) {
      const int64_t n = map ? khr_snapshot_num_blocks(map.get()) : 0;
      if (n < 0) throw std::runtime_error(std::string("ActiveWindowOutput: the map snapshot was never produced: ") + khr_last_error());
      updated_blocks_.assign(static_cast<size_t>(n > 0 ? n : 0), BlockIndex{0, 0, 0});
      // (fails when more blocks were updated than the snapshot holds -- max_snapshot_blocks: the clone is incomplete and says so)
      if (n > 0 && khr_snapshot_download(map.get(), updated_blocks_[0].data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n) < 0) {
        updated_blocks_.clear();
        throw std::runtime_error(std::string("ActiveWindowOutput::updatedBlocks: ") + khr_last_error());
      }
      std::sort(updated_blocks_.begin(), updated_blocks_.end());
      indices_valid_ = true;
    }
    return updated_blocks_;
  }
  // the cloned blocks (all snapshotted layers, or distance / weight only) in ONE packed transfer
  std::vector<BlockCopy> cloneUpdated(bool tsdf_only = false) const {
    std::vector<BlockCopy> out;
    const int64_t n = map ? khr_snapshot_num_blocks(map.get()) : 0;
    if (n <= 0) return out;
    khr_config cfg{};
    khr_get_config(map_ctx, &cfg);
    const size_t nv = static_cast<size_t>(cfg.voxels_per_side) * cfg.voxels_per_side * cfg.voxels_per_side, N = static_cast<size_t>(n);
    std::vector<int32_t> idx(3 * N);
    std::vector<float> d(N * nv), w(N * nv);
    std::vector<uint8_t> col, fl;
    std::vector<uint64_t> lo;
    std::vector<uint32_t> lab;
    if (!tsdf_only) { col.resize(4 * N * nv); fl.resize(N * nv); lo.resize(N * nv); lab.resize(N * nv); }
    const int64_t k = khr_snapshot_download(map.get(), idx.data(), d.data(), w.data(), tsdf_only ? nullptr : col.data(),
                                            tsdf_only ? nullptr : lo.data(), tsdf_only ? nullptr : fl.data(),
                                            tsdf_only ? nullptr : lab.data(), n);
    if (k < 0) throw std::runtime_error(std::string("ActiveWindowOutput::cloneUpdated: ") + khr_last_error());
    for (int64_t i = 0; i < k; ++i) {
      BlockCopy b;
      b.index = {idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]};
      b.distance.assign(d.begin() + i * nv, d.begin() + (i + 1) * nv);
      b.weight.assign(w.begin() + i * nv, w.begin() + (i + 1) * nv);
      if (!tsdf_only) {
        b.color.assign(col.begin() + 4 * i * nv, col.begin() + 4 * (i + 1) * nv);
        b.last_observed.assign(lo.begin() + i * nv, lo.begin() + (i + 1) * nv);
        b.flags.assign(fl.begin() + i * nv, fl.begin() + (i + 1) * nv);
        b.semantic_label.assign(lab.begin() + i * nv, lab.begin() + (i + 1) * nv);
      }
      b.block_flags = 1;  // KHR_BLK_UPDATED: what made it part of the clone
      out.push_back(std::move(b));
    }
    std::sort(out.begin(), out.end(), [](const BlockCopy& a, const BlockCopy& b) { return a.index < b.index; });
    return out;
  }
  std::vector<BlockCopy> cloneUpdatedTsdf() const { return cloneUpdated(true); }
  // true when the output updated more blocks than its snapshot could hold (config max_snapshot_blocks): updatedBlocks() /
  // cloneUpdated() of such an output throw instead of handing out a partial clone
  bool mapOverflowed() const {
    if (!map) return false;
    khr_config cfg{};
    if (khr_get_config(map_ctx, &cfg) < 0) return false;
    const int64_t n = khr_snapshot_num_blocks(map.get());
    const int64_t cap = snapshot_capacity > 0 ? snapshot_capacity : (cfg.max_snapshot_blocks ? cfg.max_snapshot_blocks : 8192);
    return n > cap;
  }
  int64_t snapshot_capacity = 0;  // set by the producer (ActiveWindow::extractOutputData)

 private:
  mutable BlockIndices updated_blocks_;
  mutable bool indices_valid_ = false;
};


// ---- hydra::ActiveWindowModule role (the base class of khronos::ActiveWindow, active_window.h:67) ---------------------------
// The Hydra module owns the output queue and the module thread: the thread takes an InputPacket, calls the protected
// virtual spinOnce and pushes a non-null result to the queue (the hydra frontend pops it later: hence the snapshot in
// ActiveWindowOutput::map).  The stand-in keeps exactly that surface: constructor (config, output queue), virtual
// printInfo, protected pure-virtual spinOnce, and `step` as the body of the module thread's loop.
template <typename T>
struct InputQueue {  // hydra::InputQueue role (a mutex-guarded deque)
  using Ptr = std::shared_ptr<InputQueue<T>>;
  void push(const T& v) {
    std::lock_guard<std::mutex> lock(mutex);
    queue.push_back(v);
  }
  bool pop(T* out) {
    std::lock_guard<std::mutex> lock(mutex);
    if (queue.empty()) return false;
    *out = queue.front();
    queue.erase(queue.begin());
    return true;
  }
  size_t size() const {
    std::lock_guard<std::mutex> lock(mutex);
    return queue.size();
  }
  mutable std::mutex mutex;
  std::vector<T> queue;
};

class ActiveWindowModule {
 public:
  using OutputQueue = InputQueue<ActiveWindowOutput::Ptr>;
  ActiveWindowModule(const OutputQueue::Ptr& output_queue) : output_queue_(output_queue) {}
  virtual ~ActiveWindowModule() = default;
  virtual std::string printInfo() const { return ""; }
  // one iteration of the module thread (hydra: ActiveWindowModule::spin): process a packet, queue the output if there is one
  ActiveWindowOutput::Ptr step(const InputPacket& input) {
    ActiveWindowOutput::Ptr out = spinOnce(input);
    if (out && output_queue_) output_queue_->push(out);
    return out;
  }
  const OutputQueue::Ptr& outputQueue() const { return output_queue_; }

 protected:
  virtual ActiveWindowOutput::Ptr spinOnce(const InputPacket& input) = 0;
  OutputQueue::Ptr output_queue_;
};

// config_utilities string factory role (config::RegistrationWithConfig<Base, Derived, Config, Args...>(name) /
// config::createFromYaml): the pipeline selects the active window by `active_window: {type: "<name>", ...}`
// (uHumans2.yaml:35-36; registration at active_window.h:190-192).  The creator receives the text of the `active_window:`
// mapping's document and the output queue.
class ActiveWindowFactory {
 public:
  using Creator = std::function<std::unique_ptr<ActiveWindowModule>(const std::string& yaml_text, const ActiveWindowModule::OutputQueue::Ptr&)>;
  static bool add(const std::string& type, Creator c) {
    registry()[type] = std::move(c);
    return true;
  }
  static bool has(const std::string& type) { return registry().count(type) != 0; }
  static std::unique_ptr<ActiveWindowModule> create(const std::string& type, const std::string& yaml_text,
                                                    const ActiveWindowModule::OutputQueue::Ptr& queue) {
    auto it = registry().find(type);
    return it == registry().end() ? nullptr : it->second(yaml_text, queue);
  }

 private:
  static std::map<std::string, Creator>& registry() {
    static std::map<std::string, Creator> r;
    return r;
  }
};
// config::RegistrationWithConfig<ActiveWindowModule, Derived, Derived::Config, OutputQueue::Ptr>(name) role: a static
// member of this type inside Derived registers it (the creator is only instantiated once Derived is complete)
template <typename Derived>
struct ActiveWindowRegistration {
  explicit ActiveWindowRegistration(const std::string& name) {
    ActiveWindowFactory::add(name, [](const std::string& yaml_text, const ActiveWindowModule::OutputQueue::Ptr& queue) {
      return std::unique_ptr<ActiveWindowModule>(new Derived(Derived::Config::fromYamlString(yaml_text), queue));
    });
  }
};


// ---- hydra::timing (ElapsedTimeRecorder / ScopedTimer role) ------------------------------------------------------------
// The reference brackets its stages with `Timer timer("<scope>", stamp)` (active_window.cpp:121,152,204,220,269;
// tracking_integrator.cpp:72; free_space_motion_detector.cpp:75; connected_semantics.cpp:61; max_iou_tracker.cpp:200,217)
// and dumps `timing/stats.csv` at shutdown (khronos_ros/src/experiments/experiment_manager.cpp:251-258).  Same scope
// names here, so that a CPU-vs-GPU comparison reads the same rows.  The device work of a scope is asynchronous: with
// `ElapsedTimeRecorder::instance().sync_device = fn` set (the ActiveWindow sets it when config.timing_sync_device is on)
// a scope's end first waits for the device, which makes "active_window/all" the per-frame latency the reference measures.
namespace timing {
struct TimerStats {
  uint64_t count = 0;
  double sum = 0.0, sum_sq = 0.0, min = 0.0, max = 0.0, last = 0.0;
};
class ElapsedTimeRecorder {
 public:
  static ElapsedTimeRecorder& instance() {
    static ElapsedTimeRecorder r;
    return r;
  }
  void record(const std::string& name, double seconds) {
    std::lock_guard<std::mutex> lock(mutex_);
    TimerStats& t = stats_[name];
    if (t.count == 0) t.min = t.max = seconds;
    t.min = std::min(t.min, seconds);
    t.max = std::max(t.max, seconds);
    t.sum += seconds;
    t.sum_sq += seconds * seconds;
    t.last = seconds;
    ++t.count;
  }
  std::map<std::string, TimerStats> stats() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return stats_;
  }
  void reset() {
    std::lock_guard<std::mutex> lock(mutex_);
    stats_.clear();
  }
  // "name,mean[s],min[s],max[s],std-dev[s]" rows (ElapsedTimeRecorder::logStats; column set recalled, ASSUMPTIONS.md D.1),
  // plus the sample count as a last column
  bool logStats(const std::string& path) const {
    std::ofstream out(path);
    if (!out) return false;
    out << "name,mean[s],min[s],max[s],std-dev[s],count\n";
    for (const auto& kv : stats()) {
      const TimerStats& t = kv.second;
      const double mean = t.count ? t.sum / static_cast<double>(t.count) : 0.0;
      const double var = t.count > 1 ? std::max(0.0, (t.sum_sq - t.sum * mean) / static_cast<double>(t.count - 1)) : 0.0;
      out << kv.first << ',' << mean << ',' << t.min << ',' << t.max << ',' << std::sqrt(var) << ',' << t.count << "\n";
    }
    return true;
  }
  std::function<void()> sync_device;  // optional: called when a timer with `sync` set stops
  bool disabled = false;

 private:
  mutable std::mutex mutex_;
  std::map<std::string, TimerStats> stats_;
};

class ScopedTimer {
 public:
  ScopedTimer(std::string name, uint64_t /*timestamp_ns*/, bool sync = false)
      : name_(std::move(name)), sync_(sync), start_(std::chrono::steady_clock::now()) {}
  ~ScopedTimer() { stop(); }
  void stop() {
    if (done_) return;
    done_ = true;
    ElapsedTimeRecorder& r = ElapsedTimeRecorder::instance();
    if (r.disabled) return;
    if (sync_ && r.sync_device) r.sync_device();
    r.record(name_, std::chrono::duration<double>(std::chrono::steady_clock::now() - start_).count());
  }

 private:
  std::string name_;
  bool sync_, done_ = false;
  std::chrono::steady_clock::time_point start_;
};
}  // namespace timing

}  // namespace hydra
