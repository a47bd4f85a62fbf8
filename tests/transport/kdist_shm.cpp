// kdist_shm.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A stand-in for librccl that lets N ranks of the product's C++ tick (khronos_amd/host/sharded_fusion.cpp) run as N
// processes on ONE GPU: the nine nccl* entry points sharded_fusion.cpp binds (through KDIST_RCCL_LIB), implemented over
// a shared-memory segment with host-staged copies.  RCCL itself refuses several ranks on one device, and the development
// box has exactly one MI355X; this is how the N > 1 control flow of the product code (who sends what to whom, buffer
// sizing, trimmed exchanges, home-rank clustering) is executed and checked against the oracle before an 8-GPU node runs
// it over real RCCL / xGMI.  It proves nothing about xGMI performance and is never loaded unless KDIST_RCCL_LIB says so.
//
// Semantics: every collective is stream-ordered the blunt way -- hipStreamSynchronize(stream), then blocking copies
// device -> segment, a barrier, blocking copies segment -> device, a barrier.  That is a legal (stronger) ordering of what
// RCCL promises.  A rank that does not show up within KDIST_SHM_TIMEOUT_S (default 120) makes every rank fail with
// ncclSystemError instead of hanging the box.  KDIST_SHM_HOST=1 treats the buffers as host memory (memcpy): the transport's
// own CPU test (tests/test_cpu_shm_transport.py).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x4b445348u;  // "KDSH"

struct Header {
  std::atomic<uint32_t> magic;
  uint32_t nranks;
  uint64_t chunk;
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> failed;
};
constexpr size_t kHeaderBytes = 4096;

double nowSeconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return static_cast<double>(ts.tv_sec) + 1e-9 * static_cast<double>(ts.tv_nsec);
}

double timeoutSeconds() {
  const char* e = std::getenv("KDIST_SHM_TIMEOUT_S");
  return e ? std::max(1.0, std::atof(e)) : 120.0;
}

bool hostMode() {
  const char* e = std::getenv("KDIST_SHM_HOST");
  return e && e[0] == '1';
}

size_t typeSize(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

template <typename T>
void combineT(T* acc, const T* in, size_t n, ncclRedOp_t op) {
  switch (op) {
    case ncclSum: for (size_t i = 0; i < n; ++i) acc[i] = static_cast<T>(acc[i] + in[i]); break;
    case ncclProd: for (size_t i = 0; i < n; ++i) acc[i] = static_cast<T>(acc[i] * in[i]); break;
    case ncclMax: for (size_t i = 0; i < n; ++i) acc[i] = std::max(acc[i], in[i]); break;
    case ncclMin: for (size_t i = 0; i < n; ++i) acc[i] = std::min(acc[i], in[i]); break;
    default: break;
  }
}

bool combine(void* acc, const void* in, size_t bytes, ncclDataType_t t, ncclRedOp_t op) {
  if (op != ncclSum && op != ncclProd && op != ncclMax && op != ncclMin) return false;
  switch (t) {
    case ncclInt8: combineT(static_cast<int8_t*>(acc), static_cast<const int8_t*>(in), bytes, op); return true;
    case ncclUint8: combineT(static_cast<uint8_t*>(acc), static_cast<const uint8_t*>(in), bytes, op); return true;
    case ncclInt32: combineT(static_cast<int32_t*>(acc), static_cast<const int32_t*>(in), bytes / 4, op); return true;
    case ncclUint32: combineT(static_cast<uint32_t*>(acc), static_cast<const uint32_t*>(in), bytes / 4, op); return true;
    case ncclInt64: combineT(static_cast<int64_t*>(acc), static_cast<const int64_t*>(in), bytes / 8, op); return true;
    case ncclUint64: combineT(static_cast<uint64_t*>(acc), static_cast<const uint64_t*>(in), bytes / 8, op); return true;
    case ncclFloat32: combineT(static_cast<float*>(acc), static_cast<const float*>(in), bytes / 4, op); return true;
    case ncclFloat64: combineT(static_cast<double*>(acc), static_cast<const double*>(in), bytes / 8, op); return true;
    default: return false;
  }
}

}  // namespace

struct ncclComm {
  int rank = 0, nranks = 1;
  Header* hdr = nullptr;
  uint8_t* base = nullptr;
  size_t map_bytes = 0, chunk = 0;
  bool host = false;
  double timeout = 120.0;
  std::vector<uint8_t> scratch;
  uint8_t* slot(int r) const { return base + kHeaderBytes + static_cast<size_t>(r) * chunk; }

  ncclResult_t barrier() {
    if (hdr->failed.load()) return ncclSystemError;
    const uint32_t gen = hdr->generation.load();
    if (hdr->arrived.fetch_add(1) + 1 == static_cast<uint32_t>(nranks)) {
      hdr->arrived.store(0);
      hdr->generation.fetch_add(1);
      return ncclSuccess;
    }
    const double t0 = nowSeconds();
    unsigned spins = 0;
    while (hdr->generation.load() == gen) {
      if (hdr->failed.load()) return ncclSystemError;
      if ((++spins & 1023u) == 0) {
        if (nowSeconds() - t0 > timeout) {
          hdr->failed.store(1);
          std::fprintf(stderr, "[kdist_shm] rank %d: barrier timed out after %.0f s\n", rank, timeout);
          return ncclSystemError;
        }
        sched_yield();
      }
    }
    return ncclSuccess;
  }
  bool in(void* dst_host, const void* src, size_t bytes) const {
    if (host) { std::memcpy(dst_host, src, bytes); return true; }
    return hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost) == hipSuccess;
  }
  bool out(void* dst, const void* src_host, size_t bytes) const {
    if (host) { std::memcpy(dst, src_host, bytes); return true; }
    return hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice) == hipSuccess;
  }
  ncclResult_t fail(const char* what) {
    hdr->failed.store(1);
    std::fprintf(stderr, "[kdist_shm] rank %d: %s failed\n", rank, what);
    return ncclUnhandledCudaError;
  }
};

namespace {

// every contributing rank puts bytes [off, off + nb) of its send buffer into its slot; consume(off, nb) then reads the slots
template <typename Consume>
ncclResult_t exchange(ncclComm* c, const void* send, size_t bytes, bool contribute, hipStream_t stream, Consume consume) {
  if (!c) return ncclInvalidArgument;
  if (!c->host && hipStreamSynchronize(stream) != hipSuccess) return c->fail("hipStreamSynchronize");
  for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += c->chunk) {
    const size_t nb = bytes == 0 ? 0 : std::min(c->chunk, bytes - off);
    if (contribute && nb && !c->in(c->slot(c->rank), static_cast<const uint8_t*>(send) + off, nb)) return c->fail("copy in");
    ncclResult_t r = c->barrier();
    if (r != ncclSuccess) return r;
    if (nb && !consume(off, nb)) return c->fail("copy out");
    r = c->barrier();
    if (r != ncclSuccess) return r;
    if (bytes == 0) break;
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  static std::atomic<unsigned> counter{0};
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/kdist-shm-%d-%llu-%u", static_cast<int>(getpid()),
                static_cast<unsigned long long>(nowSeconds() * 1e6), counter.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
  char name[128];
  std::memcpy(name, id.internal, sizeof(name));
  name[127] = 0;
  const char* e = std::getenv("KDIST_SHM_CHUNK_MB");
  const size_t chunk = static_cast<size_t>(std::max(1, e ? std::atoi(e) : 2)) << 20;
  auto* c = new ncclComm();
  c->rank = rank;
  c->nranks = nranks;
  c->host = hostMode();
  c->timeout = timeoutSeconds();
  const double t0 = nowSeconds();
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, static_cast<off_t>(kHeaderBytes + chunk * static_cast<size_t>(nranks))) != 0) {
      std::perror("[kdist_shm] shm_open / ftruncate");
      delete c;
      return ncclSystemError;
    }
  } else {
    while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {
      if (nowSeconds() - t0 > c->timeout) { delete c; return ncclSystemError; }
      usleep(1000);
    }
    struct stat st;
    while (fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < kHeaderBytes) {
      if (nowSeconds() - t0 > c->timeout) { close(fd); delete c; return ncclSystemError; }
      usleep(1000);
    }
  }
  // the header first (rank 0 publishes nranks / chunk in it), then the whole segment
  void* hp = mmap(nullptr, kHeaderBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (hp == MAP_FAILED) { close(fd); delete c; return ncclSystemError; }
  Header* hdr = static_cast<Header*>(hp);
  if (rank == 0) {
    hdr->nranks = static_cast<uint32_t>(nranks);
    hdr->chunk = chunk;
    hdr->arrived.store(0);
    hdr->generation.store(0);
    hdr->failed.store(0);
    hdr->magic.store(kMagic);
  } else {
    while (hdr->magic.load() != kMagic) {
      if (nowSeconds() - t0 > c->timeout) { munmap(hp, kHeaderBytes); close(fd); delete c; return ncclSystemError; }
      usleep(1000);
    }
    if (hdr->nranks != static_cast<uint32_t>(nranks)) { munmap(hp, kHeaderBytes); close(fd); delete c; return ncclInvalidArgument; }
  }
  c->chunk = hdr->chunk;
  c->map_bytes = kHeaderBytes + c->chunk * static_cast<size_t>(nranks);
  munmap(hp, kHeaderBytes);
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->base = static_cast<uint8_t*>(p);
  c->hdr = reinterpret_cast<Header*>(p);
  c->scratch.resize(c->chunk);
  const ncclResult_t r = c->barrier();  // everybody is attached: the name can go (no leak if a rank dies later)
  if (rank == 0) shm_unlink(name);
  if (r != ncclSuccess) { munmap(p, c->map_bytes); delete c; return r; }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  if (comm->base) munmap(comm->base, comm->map_bytes);
  delete comm;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
  const size_t bytes = sendcount * typeSize(datatype);
  if (!typeSize(datatype)) return ncclInvalidArgument;
  return exchange(comm, sendbuff, bytes, true, stream, [&](size_t off, size_t nb) {
    for (int r = 0; r < comm->nranks; ++r)
      if (!comm->out(static_cast<uint8_t*>(recvbuff) + static_cast<size_t>(r) * bytes + off, comm->slot(r), nb)) return false;
    return true;
  });
}

static ncclResult_t reduceImpl(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                               ncclComm_t comm, hipStream_t stream) {
  const size_t bytes = count * typeSize(datatype);
  if (!typeSize(datatype)) return ncclInvalidArgument;
  bool ok_type = true;
  const ncclResult_t r = exchange(comm, sendbuff, bytes, true, stream, [&](size_t off, size_t nb) {
    if (root >= 0 && comm->rank != root) return true;
    std::memcpy(comm->scratch.data(), comm->slot(0), nb);
    for (int k = 1; k < comm->nranks; ++k)
      if (!combine(comm->scratch.data(), comm->slot(k), nb, datatype, op)) ok_type = false;
    return comm->out(static_cast<uint8_t*>(recvbuff) + off, comm->scratch.data(), nb);
  });
  return r != ncclSuccess ? r : (ok_type ? ncclSuccess : ncclInvalidArgument);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  return reduceImpl(sendbuff, recvbuff, count, datatype, op, -1, comm, stream);
}

ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream) {
  if (!comm || root < 0 || root >= comm->nranks) return ncclInvalidArgument;
  return reduceImpl(sendbuff, recvbuff, count, datatype, op, root, comm, stream);
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
  if (!comm || root < 0 || root >= comm->nranks) return ncclInvalidArgument;
  const size_t bytes = count * typeSize(datatype);
  if (!typeSize(datatype)) return ncclInvalidArgument;
  return exchange(comm, sendbuff, bytes, comm->rank == root, stream, [&](size_t off, size_t nb) {
    if (comm->rank == root && recvbuff == sendbuff) return true;
    return comm->out(static_cast<uint8_t*>(recvbuff) + off, comm->slot(root), nb);
  });
}

// every rank puts its whole send buffer (all destinations, chunk by chunk) into its slot; rank r copies out, from each peer's
// slot, the part of the chunk that lies inside [sdispls_of_peer[r], + sendcounts_of_peer[r]).  The peers' counts and
// displacements towards r are not known to r (only its own receive side is): they travel in a first, fixed-size exchange.
ncclResult_t ncclAllToAllv(const void* sendbuff, const size_t sendcounts[], const size_t sdispls[], void* recvbuff, const size_t recvcounts[],
                           const size_t rdispls[], ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  if (!comm || !sendcounts || !sdispls || !recvcounts || !rdispls) return ncclInvalidArgument;
  const size_t ts = typeSize(datatype);
  if (!ts) return ncclInvalidArgument;
  const int n = comm->nranks;
  if (!comm->host && hipStreamSynchronize(stream) != hipSuccess) return comm->fail("hipStreamSynchronize");
  // (1) the send-side tables of every rank, through the slots (host memory on both sides)
  if (2 * sizeof(uint64_t) * static_cast<size_t>(n) > comm->chunk) return ncclInvalidArgument;
  {
    uint64_t* mine = reinterpret_cast<uint64_t*>(comm->slot(comm->rank));
    for (int q = 0; q < n; ++q) {
      mine[2 * q] = static_cast<uint64_t>(sendcounts[q]);
      mine[2 * q + 1] = static_cast<uint64_t>(sdispls[q]);
    }
  }
  ncclResult_t r = comm->barrier();
  if (r != ncclSuccess) return r;
  std::vector<uint64_t> peer_count(static_cast<size_t>(n)), peer_displ(static_cast<size_t>(n));
  size_t span = 0;  // the longest send buffer (in bytes): the chunk loop runs over it on every rank
  for (int q = 0; q < n; ++q) {
    const uint64_t* t = reinterpret_cast<const uint64_t*>(comm->slot(q));
    peer_count[q] = t[2 * comm->rank];
    peer_displ[q] = t[2 * comm->rank + 1];
    if (peer_count[q] != recvcounts[q]) {
      std::fprintf(stderr, "[kdist_shm] rank %d: all-to-all-v expects %zu elements from rank %d, which sends %llu\n", comm->rank, recvcounts[q], q,
                   static_cast<unsigned long long>(peer_count[q]));
      comm->hdr->failed.store(1);
    }
    for (int d = 0; d < n; ++d) span = std::max<size_t>(span, static_cast<size_t>(t[2 * d] + t[2 * d + 1]) * ts);
  }
  r = comm->barrier();
  if (r != ncclSuccess) return r;
  // (2) the payload, chunk by chunk
  size_t my_span = 0;
  for (int q = 0; q < n; ++q) my_span = std::max(my_span, (sendcounts[q] + sdispls[q]) * ts);
  for (size_t off = 0; off < span; off += comm->chunk) {
    const size_t nb = std::min(comm->chunk, span - off);
    if (off < my_span) {
      const size_t mine = std::min(nb, my_span - off);
      if (!comm->in(comm->slot(comm->rank), static_cast<const uint8_t*>(sendbuff) + off, mine)) return comm->fail("copy in");
    }
    r = comm->barrier();
    if (r != ncclSuccess) return r;
    for (int q = 0; q < n; ++q) {
      const size_t a = static_cast<size_t>(peer_displ[q]) * ts, b = a + static_cast<size_t>(peer_count[q]) * ts;  // q's bytes for this rank
      const size_t lo = std::max(a, off), hi = std::min(b, off + nb);
      if (lo >= hi) continue;
      if (!comm->out(static_cast<uint8_t*>(recvbuff) + rdispls[q] * ts + (lo - a), comm->slot(q) + (lo - off), hi - lo)) return comm->fail("copy out");
    }
    r = comm->barrier();
    if (r != ncclSuccess) return r;
  }
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "kdist_shm: a HIP copy failed";
    case ncclSystemError: return "kdist_shm: rendezvous / barrier failed (a rank is missing or timed out)";
    case ncclInvalidArgument: return "kdist_shm: invalid argument";
    default: return "kdist_shm: error";
  }
}

}  // extern "C"
