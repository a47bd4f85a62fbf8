// object_pipeline.cpp — the object half of khronos::ActiveWindow::spinOnce as a C API for bindings (bench.py, the
// sharded multi-GPU driver): object detector -> tracker -> frame buffer per frame (active_window.cpp:130-145) and
// extraction of the tracks that left the window at output cadence (extractInactiveObjects, :251-266).  It is built
// from the same `active_window:` YAML as the ActiveWindow class and runs on a fusion context's frame slots; the
// volumetric half of spinOnce is khr_process_frame (single GPU) or the sharded tick (multi GPU, where the halo
// exchanges are RCCL collectives driven from Python).
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "active_window.h"
#include "ray_verificator.h"

using namespace khronos;

struct kop_handle {
  ActiveWindow::Config config;
  khr_ctx* ctx = nullptr;
  VolumetricMap map;
  std::unique_ptr<ObjectDetector> detector;
  std::unique_ptr<Tracker> tracker;
  std::unique_ptr<ObjectWorkerPool> pool;  // owns the extractor(s); ObjectWorkerPool role (object_worker_pool.cpp:56-146)
  std::unique_ptr<FrameDataBuffer> buffer;
  std::vector<std::shared_ptr<KhronosObjectAttributes>> last_objects;
  std::vector<std::shared_ptr<KhronosObjectAttributes>> history;  // every object handed out so far (kop_keep_objects)
  bool keep_history = false;
  std::shared_ptr<FrameData> pending;  // kop_launch_frame done, kop_finish_frame outstanding

  ~kop_handle() {
    if (pool) pool->stop();
    // frames hold leases on slots of `ctx`, which the caller destroys after this handle
    pending.reset();
    buffer.reset();
    pool.reset();
  }
};

namespace {
void setErr(char* err, int n, const std::string& s) {
  if (err && n > 0) {
    std::strncpy(err, s.c_str(), static_cast<size_t>(n) - 1);
    err[n - 1] = 0;
  }
}
}  // namespace

extern "C" {

kop_handle* kop_create(khr_ctx* ctx, const char* yaml_text, char* err, int err_len) {
  if (!ctx || !yaml_text) {
    setErr(err, err_len, "null argument");
    return nullptr;
  }
  try {
    auto h = std::make_unique<kop_handle>();
    h->config = ActiveWindow::Config::fromYamlString(yaml_text);
    h->config.checkValid();
    h->ctx = ctx;
    h->map = VolumetricMap(h->config.volumetric_map, ctx);
    const auto& c = h->config;
    if (c.object_detector_type == "ConnectedSemantics") h->detector = std::make_unique<ConnectedSemantics>(c.object_detector, h->map);
    else if (c.object_detector_type == "InstanceForwarding") h->detector = std::make_unique<InstanceForwarding>(c.instance_forwarding);
    else h->detector = std::make_unique<ObjectDetector>();
    if (c.tracker_type == "MaxIouTracker") {
      h->tracker = std::make_unique<MaxIoUTracker>(c.tracker);
    } else if (c.tracker_type == "ExternalTracker") {
      ExternalTracker::Config ec;
      ec.temporal_window = c.tracker.temporal_window;
      ec.min_num_observations = c.tracker.min_num_observations;
      h->tracker = std::make_unique<ExternalTracker>(ec);
    } else {
      h->tracker = std::make_unique<Tracker>();
    }
    if (c.object_extractor_type == "MeshObjectExtractor") {
      khr_config dc{};
      if (khr_get_config(ctx, &dc) < 0) throw std::runtime_error(khr_last_error());
      const MeshObjectExtractor::Config oec = c.object_extractor;
      ObjectWorkerPool::Config pc = c.extraction_worker;
      if (!c.detach_object_extraction) pc.num_workers = 1;  // blocking extraction needs one extractor
      h->pool = std::make_unique<ObjectWorkerPool>(
          pc, [oec, dc]() -> std::unique_ptr<ObjectExtractor> { return std::make_unique<MeshObjectExtractor>(oec, dc); });
    }
    h->buffer = std::make_unique<FrameDataBuffer>(c.frame_data_buffer);
    return h.release();
  } catch (const std::exception& e) {
    setErr(err, err_len, e.what());
    return nullptr;
  }
}

void kop_destroy(kop_handle* h) { delete h; }

// The per-frame work in two halves so that the caller can queue the next frame's device work in between:
// kop_launch_frame = object_detector_->processInput + the tracker's measurement passes (enqueued, not awaited);
// kop_finish_frame = tracker association, trimBuffer, storeData (active_window.cpp:130-145).  Returns the track count.
int kop_finish_frame(kop_handle* h, char* err, int err_len) {
  if (!h) return KHR_EINVAL;
  khr_host_trace("kop_finish_enter");
  struct Exit { ~Exit() { khr_host_trace("kop_finish_exit"); } } on_exit;
  try {
    if (h->pending) {
      std::shared_ptr<FrameData> data = h->pending;
      h->pending.reset();
      if (auto* iou = dynamic_cast<MaxIoUTracker*>(h->tracker.get())) iou->completeInput(*data);
      else h->tracker->processInput(*data);
      h->buffer->trimBuffer(h->tracker->getTracks());
      h->buffer->storeData(data);
    }
    return static_cast<int>(h->tracker->getTracks().size());
  } catch (const std::exception& e) {
    setErr(err, err_len, e.what());
    return KHR_EDEVICE;
  }
}

int kop_launch_frame(kop_handle* h, int slot, uint64_t stamp_ns, const double* world_T_sensor, const khr_sensor* sensor,
                     int n_dynamic_clusters, char* err, int err_len) {
  if (!h || !world_T_sensor || !sensor) return KHR_EINVAL;
  if (h->pending) {
    const int rc = kop_finish_frame(h, err, err_len);
    if (rc < 0) return rc;
  }
  khr_host_trace("kop_launch_enter");
  struct Exit { ~Exit() { khr_host_trace("kop_launch_exit"); } } on_exit;
  try {
    auto data = std::make_shared<FrameData>();
    InputData& in = data->input;
    in.timestamp_ns = stamp_ns;
    std::memcpy(in.world_T_sensor, world_T_sensor, sizeof(in.world_T_sensor));
    std::memcpy(in.world_T_body, world_T_sensor, sizeof(in.world_T_body));
    in.sensor = {sensor->width, sensor->height, sensor->fx, sensor->fy, sensor->cx, sensor->cy, sensor->min_range, sensor->max_range};
    in.ctx = h->ctx;
    in.slot = slot;
    in.retainSlot();
    data->num_dynamic_clusters = n_dynamic_clusters;
    if (n_dynamic_clusters > 0) FreeSpaceMotionDetector::fetchClusters(h->map, *data);
    h->detector->processInput(h->map, *data);  // cached when khr_process_frame ran with KHR_PF_OBJECTS
    khr_host_trace("kop_launch_detected");
    if (auto* iou = dynamic_cast<MaxIoUTracker*>(h->tracker.get())) iou->beginInput(*data);
    h->pending = data;
    return KHR_OK;
  } catch (const std::exception& e) {
    setErr(err, err_len, e.what());
    return KHR_EDEVICE;
  }
}

int kop_process_frame(kop_handle* h, int slot, uint64_t stamp_ns, const double* world_T_sensor, const khr_sensor* sensor,
                      int n_dynamic_clusters, char* err, int err_len) {
  const int rc = kop_launch_frame(h, slot, stamp_ns, world_T_sensor, sensor, n_dynamic_clusters, err, err_len);
  if (rc < 0) return rc;
  return kop_finish_frame(h, err, err_len);
}

// extractInactiveObjects (active_window.cpp:251-266): inactive tracks leave the tracker and go to the extractor -- on the
// worker thread when `detach_object_extraction` is set (the reference's default), in line otherwise.  Returns the number
// of objects handed out with this call (detached: the extractions that finished since the last call, like
// ObjectWorkerPool::getFinishedExtractions); *n_removed = tracks removed now, *n_vertices = mesh vertices of those objects.
int kop_extract_inactive(kop_handle* h, int* n_removed, uint64_t* n_vertices, char* err, int err_len) {
  if (!h) return KHR_EINVAL;
  if (n_removed) *n_removed = 0;
  if (n_vertices) *n_vertices = 0;
  if (h->pending) {
    const int rc = kop_finish_frame(h, err, err_len);
    if (rc < 0) return rc;
  }
  khr_host_trace("kop_extract_enter");
  struct Exit { ~Exit() { khr_host_trace("kop_extract_exit"); } } on_exit;
  try {
    h->last_objects.clear();
    Tracks& tracks = h->tracker->getTracks();
    const TimeStamp stamp = h->buffer->empty() ? 0 : h->buffer->getLatestData().input.timestamp_ns;
    for (auto it = tracks.begin(); it != tracks.end();) {
      if (it->is_active) {
        ++it;
        continue;
      }
      if (h->pool) h->pool->submit(stamp, std::move(*it), *h->buffer);  // (the copy of the buffer keeps the frames alive)
      if (n_removed) ++*n_removed;
      it = tracks.erase(it);
    }
    if (h->pool) {
      if (!h->config.detach_object_extraction) h->pool->join();
      h->pool->fill(h->last_objects);
      if (n_vertices)
        for (const auto& o : h->last_objects) *n_vertices += o->mesh.numVertices();
      if (h->keep_history) h->history.insert(h->history.end(), h->last_objects.begin(), h->last_objects.end());
    }
    return static_cast<int>(h->last_objects.size());
  } catch (const std::exception& e) {
    setErr(err, err_len, e.what());
    return KHR_EDEVICE;
  }
}

// wait for the detached extractions (ObjectWorkerPool::join role); returns the number of finished objects not handed out yet
int kop_join(kop_handle* h, char* err, int err_len) {
  if (!h) return KHR_EINVAL;
  if (!h->pool) return 0;
  try {
    h->pool->join();
    std::vector<std::shared_ptr<KhronosObjectAttributes>> done;
    h->pool->fill(done);
    if (h->keep_history) h->history.insert(h->history.end(), done.begin(), done.end());
    const int n_done = static_cast<int>(done.size());
    std::move(done.begin(), done.end(), std::back_inserter(h->last_objects));
    return n_done;
  } catch (const std::exception& e) {
    setErr(err, err_len, e.what());
    return KHR_EDEVICE;
  }
}

// The objects handed out so far (what a Khronos sink would have received with the outputs): kept when asked for
// (parity tests compare every extracted object with the oracle's restatement of the extractor).
int kop_keep_objects(kop_handle* h, int on) {
  if (!h) return KHR_EINVAL;
  h->keep_history = on != 0;
  if (!on) h->history.clear();
  return KHR_OK;
}
int kop_num_objects(kop_handle* h) { return h ? static_cast<int>(h->history.size()) : KHR_EINVAL; }
// record of object i: semantic label, vertices, first / last observed stamp, trajectory length, bounding box min / max
int kop_get_object(kop_handle* h, int i, int64_t* meta /* 5 */, float* bbox /* 6 */) {
  if (!h || i < 0 || i >= static_cast<int>(h->history.size()) || !meta || !bbox) return KHR_EINVAL;
  const KhronosObjectAttributes& o = *h->history[i];
  meta[0] = o.semantic_label;
  meta[1] = static_cast<int64_t>(o.mesh.numVertices());
  meta[2] = o.first_observed_ns.empty() ? 0 : static_cast<int64_t>(o.first_observed_ns.front());
  meta[3] = o.last_observed_ns.empty() ? 0 : static_cast<int64_t>(o.last_observed_ns.front());
  meta[4] = static_cast<int64_t>(o.trajectory_positions.size());
  for (int d = 0; d < 3; ++d) {
    bbox[d] = o.bounding_box.min[d];
    bbox[3 + d] = o.bounding_box.max[d];
  }
  return KHR_OK;
}
// mesh of object i (vertices in the bounding-box frame, mesh_object_extractor.cpp:299-302): 3 floats + 1 label per vertex
int64_t kop_get_object_mesh(kop_handle* h, int i, float* points, uint32_t* labels, int64_t cap) {
  if (!h || i < 0 || i >= static_cast<int>(h->history.size())) return KHR_EINVAL;
  const hydra::Mesh& m = h->history[i]->mesh;
  const int64_t n = static_cast<int64_t>(m.numVertices());
  if (n > cap) return KHR_EINVAL;
  if (points && n) std::memcpy(points, m.points.data(), sizeof(float) * 3 * n);
  if (labels && n) std::memcpy(labels, m.labels.data(), sizeof(uint32_t) * n);
  return n;
}

int kop_num_tracks(kop_handle* h) { return h ? static_cast<int>(h->tracker->getTracks().size()) : KHR_EINVAL; }
int kop_num_buffered_frames(kop_handle* h) { return h ? static_cast<int>(h->buffer->size()) : KHR_EINVAL; }

// tracks as flat records: id, is_dynamic, is_active, category, n_observations, first_seen, last_seen, confidence * 1e6
int kop_get_tracks(kop_handle* h, int64_t* out /* 8 per track */, int cap) {
  if (!h) return KHR_EINVAL;
  const Tracks& tracks = h->tracker->getTracks();
  int n = 0;
  for (const Track& t : tracks) {
    if (n < cap && out) {
      int64_t* o = out + 8 * n;
      o[0] = t.id;
      o[1] = t.is_dynamic;
      o[2] = t.is_active;
      o[3] = t.semantics ? t.semantics->category_id : -1;
      o[4] = static_cast<int64_t>(t.observations.size());
      o[5] = static_cast<int64_t>(t.first_seen);
      o[6] = static_cast<int64_t>(t.last_seen);
      o[7] = static_cast<int64_t>(static_cast<double>(t.confidence) * 1e6 + 0.5);
    }
    ++n;
  }
  return n;
}

// RayChangeDetector::detectChanges for bindings that do not go through the C++ class (and for the CPU tests): one point's
// presence / absence observations -> out[0] = closest_absent, out[1] = furthest_persistent; return value bit 0 / bit 1 =
// which of the two exist, < 0 on a bad configuration.
int khr_host_detect_changes(const uint64_t* present, int64_t n_present, const uint64_t* absent, int64_t n_absent,
                            float temporal_resolution, int64_t window_size, int use_relative_confidence, float absence_confidence,
                            float presence_confidence, int forward, uint64_t* out) {
  if ((n_present > 0 && !present) || (n_absent > 0 && !absent) || n_present < 0 || n_absent < 0 || !out || window_size < 0) return KHR_EINVAL;
  try {
    khronos::RayChangeDetector::Config cfg;
    cfg.temporal_resolution = temporal_resolution;
    cfg.window_size = static_cast<size_t>(window_size);
    cfg.use_relative_confidence = use_relative_confidence != 0;
    cfg.absence_confidence = absence_confidence;
    cfg.presence_confidence = presence_confidence;
    const khronos::RayChangeDetector det(cfg);
    const auto r = det.detectChanges(present, static_cast<size_t>(n_present), absent, static_cast<size_t>(n_absent), forward != 0);
    out[0] = r.closest_absent.value_or(0);
    out[1] = r.furthest_persistent.value_or(0);
    return (r.closest_absent ? 1 : 0) | (r.furthest_persistent ? 2 : 0);
  } catch (const std::exception&) {
    return KHR_EINVAL;
  }
}

}  // extern "C"
