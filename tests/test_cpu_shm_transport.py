"""-m "not gpu": the shared-memory stand-in for librccl (tests/transport/kdist_shm.cpp, test infrastructure) on host buffers
(KDIST_SHM_HOST=1): the collectives sharded_fusion.cpp issues -- all-gather, all-reduce (sum / max), reduce to a root, broadcast, all-to-all-v --
across 3 processes, with buffers larger than one exchange chunk.  It is the transport the GPU tests run the product's C++ tick
over with N ranks on one device (tests/test_gpu_dist_multiproc.py), so it is checked on its own first."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "transport", "kdist_shm.cpp")
LIB = os.path.join(HERE, "transport", "libkdist_shm.so")


def build_transport():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRC, "-I/opt/rocm/include", "-L/opt/rocm/lib",
                               "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _worker(rank, world, id_path):
    lib = C.CDLL(LIB)
    vp, sz = C.c_void_p, C.c_size_t
    lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, UniqueId, C.c_int]
    lib.ncclAllGather.argtypes = [vp, vp, sz, C.c_int, vp, vp]
    lib.ncclAllReduce.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp]
    lib.ncclReduce.argtypes = [vp, vp, sz, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.ncclBroadcast.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp]
    lib.ncclAllToAllv.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, vp]
    lib.ncclCommDestroy.argtypes = [vp]
    U8, I32, U32, I64, U64 = 1, 2, 3, 4, 5  # ncclDataType_t
    SUM, MAX = 0, 2                 # ncclRedOp_t
    uid = UniqueId()
    if rank == 0:
        assert lib.ncclGetUniqueId(C.byref(uid)) == 0
        with open(id_path + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.rename(id_path + ".tmp", id_path)
    else:
        import time
        t0 = time.time()
        while not os.path.exists(id_path):
            assert time.time() - t0 < 60
            time.sleep(0.01)
        C.memmove(C.byref(uid), open(id_path, "rb").read(), 128)
    comm = vp()
    assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    p = lambda a: a.ctypes.data_as(vp)
    rng = lambda r: np.random.default_rng(100 + r)
    n = (3 << 20) + 12345  # bytes: three chunks and a bit at KDIST_SHM_CHUNK_MB=1
    mine = rng(rank).integers(0, 256, n, dtype=np.uint8)
    # all-gather, receive buffer separate and in place (the send buffer is this rank's part of the receive buffer)
    recv = np.zeros(world * n, np.uint8)
    assert lib.ncclAllGather(p(mine), p(recv), n, U8, comm, None) == 0
    want = np.concatenate([rng(r).integers(0, 256, n, dtype=np.uint8) for r in range(world)])
    assert np.array_equal(recv, want)
    recv2 = np.zeros(world * n, np.uint8)
    recv2[rank * n:(rank + 1) * n] = mine
    assert lib.ncclAllGather(p(recv2[rank * n:]), p(recv2), n, U8, comm, None) == 0
    assert np.array_equal(recv2, want)
    # all-reduce in place: int64 sum, int64 max, uint64 sum (wraps)
    m = 300_000
    vals = [rng(10 + r).integers(-2**40, 2**40, m, dtype=np.int64) for r in range(world)]
    a = vals[rank].copy()
    assert lib.ncclAllReduce(p(a), p(a), m, I64, SUM, comm, None) == 0
    assert np.array_equal(a, sum(vals))
    a = vals[rank].copy()
    assert lib.ncclAllReduce(p(a), p(a), m, I64, MAX, comm, None) == 0
    assert np.array_equal(a, np.maximum.reduce(vals))
    uvals = [rng(20 + r).integers(0, 2**64 - 1, m, dtype=np.uint64) for r in range(world)]
    u = uvals[rank].copy()
    assert lib.ncclAllReduce(p(u), p(u), m, U64, SUM, comm, None) == 0
    assert np.array_equal(u, np.add.reduce(uvals))
    # reduce to root 1 (in place): only the root's buffer changes
    u = uvals[rank].copy()
    assert lib.ncclReduce(p(u), p(u), m, U64, SUM, 1 % world, comm, None) == 0
    assert np.array_equal(u, np.add.reduce(uvals) if rank == 1 % world else uvals[rank])
    # broadcast from the last rank, in place on the root
    img = rng(30 + rank).integers(-5, 5, m, dtype=np.int32)
    assert lib.ncclBroadcast(p(img), p(img), m, I32, world - 1, comm, None) == 0
    assert np.array_equal(img, rng(30 + world - 1).integers(-5, 5, m, dtype=np.int32))
    # a one-element exchange (the 8-byte agreement all-reduce of kdist_output)
    one = np.array([rank + 7], np.int64)
    assert lib.ncclAllReduce(p(one), p(one), 1, I64, MAX, comm, None) == 0
    assert one[0] == world + 6
    # all-to-all-v (the compact mesh halo's answers): uneven counts incl. zero, gaps between the segments, more than one chunk
    def count(src, dst):
        return 0 if (src + dst) % 3 == 1 else 150_000 + 70_001 * src + 13 * dst
    def seg(src, dst):
        return np.random.default_rng(1000 + 10 * src + dst).integers(0, 2**32, count(src, dst), dtype=np.uint32)
    sc = np.array([count(rank, d) for d in range(world)], np.uint64)
    sd = np.concatenate([[5], 5 + np.cumsum(sc[:-1] + 9)]).astype(np.uint64)   # (9 unused words between segments)
    sbuf = np.zeros(int(sd[-1] + sc[-1]) + 3, np.uint32)
    for d in range(world):
        sbuf[int(sd[d]):int(sd[d] + sc[d])] = seg(rank, d)
    rcnt = np.array([count(q, rank) for q in range(world)], np.uint64)
    rd = np.concatenate([[0], np.cumsum(rcnt[:-1])]).astype(np.uint64)
    rbuf = np.zeros(int(rcnt.sum()) + 1, np.uint32)
    assert lib.ncclAllToAllv(p(sbuf), p(sc), p(sd), p(rbuf), p(rcnt), p(rd), U32, comm, None) == 0
    for q in range(world):
        assert np.array_equal(rbuf[int(rd[q]):int(rd[q] + rcnt[q])], seg(q, rank)), q
    lib.ncclCommDestroy(comm)
    print("SHM_OK %d" % rank)


def test_shm_transport_collectives(tmp_path):
    build_transport()
    world = 3
    env = dict(os.environ, KDIST_SHM_HOST="1", KDIST_SHM_CHUNK_MB="1", KDIST_SHM_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(tmp_path / "id")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "SHM_OK %d" % r in o, o[-3000:]


def test_shm_transport_missing_rank_times_out(tmp_path):
    """a rank that never shows up is an error on the others, not a hang"""
    build_transport()
    env = dict(os.environ, KDIST_SHM_HOST="1", KDIST_SHM_TIMEOUT_S="2")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "0", "2", str(tmp_path / "id")], env=env, capture_output=True, text=True,
                       timeout=60)
    assert p.returncode != 0 and "SHM_OK" not in p.stdout


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
