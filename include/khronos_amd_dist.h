/* khronos_amd_dist.h -- C ABI of the multi-GPU (sharded) active-window tick, libkhronos_amd_host.so.
 *
 * TEST-INFRA-FREE product code: RCCL collectives issued from C++ on the fusion context's own HIP stream.  One process per GPU,
 * one khr_ctx per process created with khr_config.rank / world_size (hash-range block ownership, owner computes).  The
 * reference has no multi-process mode (khronos/src/active_window/active_window.cpp:118-174 is one thread, one map); what
 * these calls replace, per rank, is the body of ActiveWindow::spinOnce -> updateMap / extractOutputData
 * (active_window.cpp:186-260) with the exchanges DESIGN.md section 5 lists between the same steps.
 *
 * Rendezvous: rank 0 calls kdist_unique_id and hands the 128 bytes to the other ranks by whatever the launcher offers
 * (env var, file, MPI, torch store); every rank then calls kdist_create with the same id.  All calls of one handle are
 * collective: every rank must make the same call sequence.
 *
 * Errors: negative khr_status (include/khronos_amd.h), text via khr_last_error().  Exchange buffers are sized at create;
 * exceeding one on ANY rank is KHR_ENOMEM from kdist_output on EVERY rank (the ranks agree on it inside the output's
 * count all-reduce, so nobody is left waiting in a collective), never a silent truncation.  The agreement covers every local
 * failure of kdist_output in front of that all-reduce (request list, record export, the dropped-record counter -- whose NEW
 * drops since the last report count, the counter itself is sticky); the output that reports a fault has not meshed, and the mesh
 * requests of that output are not retried: the caller raises the capacity and the blocks are meshed when they are next updated.
 * kdist_tick's own failures (a frame that cannot be ingested, a device error) are rank-local and end the run on that rank.
 * KDIST_RCCL_LIB=<path> (environment): load the eight nccl* entry points from that library instead of librccl.so.1 -- a
 * site's own RCCL build, or the shared-memory transport of tests/transport/ that runs N ranks on one GPU for the tests.  The capacities bound what a
 * rank may hold, not what travels: the halo all-gather of a tick ships, per rank, the live-block count of the fullest rank
 * (it rides in the tick's seed-count all-reduce, rounded up to 256 records), the mesh-record all-gather of an output the
 * record count of the fullest rank (one 8-byte max all-reduce).
 */
#ifndef KHRONOS_AMD_DIST_H_
#define KHRONOS_AMD_DIST_H_

#include <stddef.h>
#include <stdint.h>

#include "khronos_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kdist_handle kdist_handle;

enum kdist_flags {
  KDIST_MOTION = 1,          /* run the free-space motion detector (motion_detection/, a8-a11) inside the tick */
  KDIST_SHARD_MOTION = 2,    /* cluster each camera on its home rank (camera % world) and broadcast the dynamic image;
                                off: every rank all-reduces the keys and clusters every camera itself */
  KDIST_ALWAYS_EXCHANGE = 4, /* issue the collectives even when world_size == 1 (single-GPU smoke of the RCCL path) */
  KDIST_EMULATE = 8          /* measurement aid: run rank `rank` of `world_size` alone -- no communicator, every collective
                                skipped, the export / import kernels run on this rank's records only.  The map it builds is
                                the rank's shard WITHOUT its neighbours' halos: timing only, never a result. */
};

/* ncclGetUniqueId: 128 opaque bytes. */
int kdist_unique_id(char id_out[128]);

/* n_cameras: frames per tick.  halo_cap: ever-free halo records (528 B) a rank may export per tick.  mesh_req_cap /
 * mesh_rec_cap: marching-cubes halo plane requests / records per output stage.  Takes the context's stream over
 * (khr_set_stream) for its lifetime.  NULL on failure. */
kdist_handle* kdist_create(khr_ctx* ctx, const khr_sensor* sensor, int rank, int world_size, const char unique_id[128],
                           int n_cameras, int64_t halo_cap, int64_t mesh_req_cap, int64_t mesh_rec_cap, uint32_t flags);
void kdist_destroy(kdist_handle* h);

/* the HIP stream (hipStream_t) every call of this handle is ordered on */
void* kdist_stream(kdist_handle* h);

/* records per rank shipped by the last tick's halo all-gather and by the last output's mesh-record all-gather (0 = no
 * such exchange yet; the capacity when the trimmed form was not available: motion detector off, more than 8 cameras) */
int kdist_last_exchange(kdist_handle* h, int64_t* halo_records_per_rank, int64_t* mesh_records_per_rank);
/* The mesh halo of the last output in bytes: out[0] requests this rank sent (all-gathered: it received world x that), out[1]
 * answers sent, out[2] answers received, out[3] the number of answers received.  Default (up to 16 ranks): the compact form of
 * khronos_amd.h -- per-relation face / line / voxel answers, ncclAllToAllv owner -> requester, counts derived from the
 * all-gathered request headers (no count exchange); in that form mesh_records_per_rank above is the number of answers this
 * rank received.  KDIST_MESH_HALO=records in the environment of kdist_create selects the all-gather of whole-block records.
 * Emulation (one rank of N without the others): what this rank would receive; "sent" is reported equal to it. */
int kdist_last_mesh_exchange(kdist_handle* h, int64_t out[4]);

/* What the tick's / the output's collectives cost (the first multi-GPU record has to explain itself): per kind of collective
 * the calls, the bytes THIS rank sent and -- while profiling is on -- the milliseconds between HIP events recorded around each
 * call on the handle's stream (~5 us of stream time per bracket: keep it off inside a throughput measurement).
 * kdist_profile resets the counters; kdist_profile_get fills min(cap, count) entries and returns the count. */
typedef struct kdist_coll_stat {
  char name[32];       /* frames_allgather, converted_allgather, counts_allreduce, motion_keys_reduce, dynamic_image_broadcast,
                          halo_allgather, mesh_request_allgather, mesh_agree_allreduce, mesh_record_allgather,
                          mesh_answer_alltoallv */
  uint64_t calls;
  uint64_t bytes_sent;
  double ms;
} kdist_coll_stat;
int kdist_profile(kdist_handle* h, int enable);
int kdist_profile_get(kdist_handle* h, kdist_coll_stat* out, int cap);

/* ncclAllGather of one packed camera frame per rank (device pointer, same byte count on every rank); *gathered_out is a
 * device buffer of world_size * bytes owned by the handle, valid until the next call. */
int kdist_gather_frames(kdist_handle* h, const void* packed_local, size_t bytes, void** gathered_out);

/* one tick: every camera's frame (device pointers) into this rank's shard.  slots_out[n]: frame slots; clusters_out[n]:
 * dynamic clusters per camera (identical on every rank). */
int kdist_tick(kdist_handle* h, uint64_t stamp, const khr_frame* frames, int n, int* slots_out, int* clusters_out);

/* kdist_tick with SENDER-SIDE INGEST (one camera per rank, n == world_size): frames[n] carry every camera's pose and stamp,
 * only frames[rank] its (device) images.  The rank converts its own frame (khr_tick_ingest of one frame; *own_slot_out is
 * that slot: the one the object half of the active window works on), packs the converted planes (khr_export_converted:
 * 12 B per pixel), the ranks all-gather them on the context's stream, and all n cameras are adopted where the all-gather
 * put them (khr_tick_adopt: seed test + reset of the dynamic image, 8 B per pixel, no copy) instead of every rank
 * converting every camera's raw frame (31 B per pixel and camera).  Everything behind the ingest is kdist_tick.
 * The rig is homogeneous: which optional planes exist (colour, labels) is decided by frames[rank] for ALL cameras -- a
 * depth-only rig passes color == label == NULL in its own frame on every rank and nothing is blended / fused for any camera,
 * exactly as kdist_tick does with depth-only frames.
 * `emulated_gather`: KDIST_EMULATE only -- the stand-in for the all-gather's receive buffer, n x khr_converted_bytes of
 * planes converted beforehand (entry `rank` is not read); NULL otherwise. */
int kdist_tick_own(kdist_handle* h, uint64_t stamp, const khr_frame* frames, int n, const void* emulated_gather, int* slots_out,
                   int* clusters_out, int* own_slot_out);

/* output stage (extractOutputData): mesh halo exchange, marching cubes over the owned updated blocks, archival of the
 * blocks that left the window, flag clearing.  Results stay in the context (khr_download_mesh, khr_last_removed). */
int kdist_output(kdist_handle* h);

#ifdef __cplusplus
}
#endif
#endif
