// Test infrastructure: the handful of RCCL entry points libsophtmi calls, carried over Unix domain sockets between
// rank processes that SHARE one GPU.  RCCL itself refuses two ranks on one device, so on a one-GPU box the library's
// RCCL branches (peer ranks, byte counts, offsets, the order of sends and receives inside a group, all-gather and
// all-reduce) would never run; linked against this stand-in instead of librccl they do, under the same tests as the
// host-staged transport.
//
// What it checks beyond moving the bytes: every message carries its length, and a receive whose length differs from
// what the sender sent fails with ncclInvalidArgument (real RCCL would hang or corrupt memory); a receive that no
// send answers within the time limit fails with ncclSystemError instead of hanging.
//
// What it does not model: asynchrony.  Every call completes on the host before it returns (the stream an operation
// names is synchronised first), so a missing event between two streams cannot show here.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

struct ncclComm {
  int rank = 0, n = 0;
  std::string dir;
  int listen_fd = -1;
  std::vector<int> fd;  // one stream socket per peer (fd[rank] = -1)
};

namespace {

constexpr uint64_t kMagic = 0x534d495f52434c31ull;  // "SMI_RCL1"
constexpr int kTimeoutMs = 60000;

struct Op {
  bool send;
  void* ptr;
  size_t bytes;
  int peer;
  ncclComm* comm;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local char g_detail[512] = "";

ncclResult_t fail(ncclResult_t code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
  snprintf(g_detail, sizeof(g_detail), fmt, a, b, c);
  fprintf(stderr, "[rccl stand-in] %s\n", g_detail);
  return code;
}

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

bool write_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t k = write(fd, c, n);
    if (k < 0 && (errno == EINTR)) continue;
    if (k <= 0) return false;
    c += k;
    n -= (size_t)k;
  }
  return true;
}
bool read_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    ssize_t k = read(fd, c, n);
    if (k < 0 && (errno == EINTR)) continue;
    if (k <= 0) return false;
    c += k;
    n -= (size_t)k;
  }
  return true;
}

struct Wire {  // one message on one socket: 16-byte header, then the payload
  std::vector<char> buf;
  size_t done = 0;
  size_t expect = 0;     // receives: payload bytes the caller asked for
  void* device = nullptr;
};

// Moves every queued message; per peer the sends go out and the receives come in in call order.
ncclResult_t progress(ncclComm* comm, std::vector<std::vector<Wire>>& sendq, std::vector<std::vector<Wire>>& recvq) {
  const int n = comm->n;
  std::vector<size_t> si(n, 0), ri(n, 0);
  for (int p = 0; p < n; ++p)
    if (comm->fd[p] >= 0) fcntl(comm->fd[p], F_SETFL, fcntl(comm->fd[p], F_GETFL) | O_NONBLOCK);
  ncclResult_t rc = ncclSuccess;
  for (;;) {
    std::vector<pollfd> pf;
    std::vector<int> who;
    for (int p = 0; p < n; ++p) {
      short ev = 0;
      if (si[p] < sendq[p].size()) ev |= POLLOUT;
      if (ri[p] < recvq[p].size()) ev |= POLLIN;
      if (ev) {
        pf.push_back({comm->fd[p], ev, 0});
        who.push_back(p);
      }
    }
    if (pf.empty()) break;
    const int ready = poll(pf.data(), pf.size(), kTimeoutMs);
    if (ready == 0) {
      rc = fail(ncclSystemError, "rank %s%lld: no progress for 60 s (a receive without a matching send, or the reverse; first stalled peer %lld)",
                "", comm->rank, who[0]);
      break;
    }
    if (ready < 0) {
      if (errno == EINTR) continue;
      rc = fail(ncclSystemError, "poll failed%s", "");
      break;
    }
    for (size_t i = 0; i < pf.size() && rc == ncclSuccess; ++i) {
      const int p = who[i];
      if (pf[i].revents & (POLLERR | POLLNVAL)) {
        rc = fail(ncclSystemError, "socket error towards rank %s%lld", "", p);
        break;
      }
      if ((pf[i].revents & POLLOUT) && si[p] < sendq[p].size()) {
        Wire& w = sendq[p][si[p]];
        ssize_t k = write(pf[i].fd, w.buf.data() + w.done, w.buf.size() - w.done);
        if (k > 0) w.done += (size_t)k;
        else if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)
          rc = fail(ncclSystemError, "write to rank %s%lld failed", "", p);
        if (w.done == w.buf.size()) ++si[p];
      }
      if ((pf[i].revents & (POLLIN | POLLHUP)) && ri[p] < recvq[p].size() && rc == ncclSuccess) {
        Wire& w = recvq[p][ri[p]];
        // header first: the payload buffer is sized for what the caller expects
        const size_t want = w.done < 16 ? 16 - w.done : w.buf.size() - w.done;
        ssize_t k = read(pf[i].fd, w.buf.data() + w.done, want);
        if (k > 0) {
          w.done += (size_t)k;
          if (w.done == 16) {
            uint64_t head[2];
            memcpy(head, w.buf.data(), 16);
            if (head[0] != kMagic) rc = fail(ncclInternalError, "bad message header from rank %s%lld", "", p);
            else if (head[1] != w.expect)
              rc = fail(ncclInvalidArgument, "size mismatch%s: rank %lld sent %lld bytes where this receive expects another count",
                        "", p, (long long)head[1]);
          }
        } else if (k == 0) {
          rc = fail(ncclSystemError, "rank %s%lld closed its socket", "", p);
        } else if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) {
          rc = fail(ncclSystemError, "read from rank %s%lld failed", "", p);
        }
        if (w.done == w.buf.size()) ++ri[p];
      }
    }
    if (rc != ncclSuccess) break;
  }
  for (int p = 0; p < n; ++p)
    if (comm->fd[p] >= 0) fcntl(comm->fd[p], F_SETFL, fcntl(comm->fd[p], F_GETFL) & ~O_NONBLOCK);
  return rc;
}

ncclResult_t run(std::vector<Op>& ops) {
  if (ops.empty()) return ncclSuccess;
  ncclComm* comm = ops[0].comm;
  std::vector<hipStream_t> streams;
  for (const Op& o : ops) {
    if (o.comm != comm) return fail(ncclInvalidUsage, "one group, two communicators%s", "");
    if (o.peer < 0 || o.peer >= comm->n || o.peer == comm->rank)
      return fail(ncclInvalidArgument, "rank %s%lld: peer %lld is not another rank of this communicator", "", comm->rank, o.peer);
    if (std::find(streams.begin(), streams.end(), o.stream) == streams.end()) streams.push_back(o.stream);
  }
  for (hipStream_t s : streams)
    if (hipStreamSynchronize(s) != hipSuccess) return fail(ncclUnhandledCudaError, "stream synchronise failed%s", "");
  std::vector<std::vector<Wire>> sendq(comm->n), recvq(comm->n);
  for (const Op& o : ops) {
    Wire w;
    w.buf.resize(16 + o.bytes);
    w.device = o.ptr;
    w.expect = o.bytes;
    if (o.send) {
      const uint64_t head[2] = {kMagic, (uint64_t)o.bytes};
      memcpy(w.buf.data(), head, 16);
      if (o.bytes && hipMemcpy(w.buf.data() + 16, o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(ncclUnhandledCudaError, "device to host copy of a send buffer failed%s", "");
      sendq[o.peer].push_back(std::move(w));
    } else {
      recvq[o.peer].push_back(std::move(w));
    }
  }
  ncclResult_t rc = progress(comm, sendq, recvq);
  if (rc != ncclSuccess) return rc;
  for (auto& q : recvq)
    for (Wire& w : q)
      if (w.expect && hipMemcpy(w.device, w.buf.data() + 16, w.expect, hipMemcpyHostToDevice) != hipSuccess)
        return fail(ncclUnhandledCudaError, "host to device copy into a receive buffer failed%s", "");
  return ncclSuccess;
}

ncclResult_t enqueue(const Op& op) {
  g_ops.push_back(op);
  if (g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run(ops);
}

template <typename T>
void reduce_into(T* acc, const T* other, size_t count, ncclRedOp_t op) {
  for (size_t i = 0; i < count; ++i) {
    if (op == ncclSum) acc[i] += other[i];
    else if (op == ncclMin) acc[i] = std::min(acc[i], other[i]);
    else if (op == ncclMax) acc[i] = std::max(acc[i], other[i]);
    else if (op == ncclProd) acc[i] *= other[i];
  }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  FILE* f = fopen("/dev/urandom", "rb");
  unsigned char raw[8] = {0};
  if (!f || fread(raw, 1, sizeof(raw), f) != sizeof(raw)) {
    if (f) fclose(f);
    return fail(ncclSystemError, "cannot read /dev/urandom%s", "");
  }
  fclose(f);
  char* out = id->internal;
  for (int i = 0; i < 8; ++i) snprintf(out + 2 * i, 3, "%02x", raw[i]);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "bad rank / size%s", "");
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->n = nranks;
  c->fd.assign(nranks, -1);
  char tag[17];
  memcpy(tag, id.internal, 16);
  tag[16] = 0;
  c->dir = std::string("/tmp/smi_rccl_standin_") + tag;
  mkdir(c->dir.c_str(), 0700);
  auto address = [&](int r) {
    sockaddr_un a;
    memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    snprintf(a.sun_path, sizeof(a.sun_path), "%s/r%d", c->dir.c_str(), r);
    return a;
  };
  c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
  sockaddr_un mine = address(rank);
  unlink(mine.sun_path);
  if (c->listen_fd < 0 || bind(c->listen_fd, (sockaddr*)&mine, sizeof(mine)) != 0 || listen(c->listen_fd, nranks) != 0)
    return fail(ncclSystemError, "cannot listen on %s", mine.sun_path);
  // lower ranks are called, higher ranks call in and name themselves
  for (int p = 0; p < rank; ++p) {
    sockaddr_un a = address(p);
    int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    bool ok = false;
    for (int attempt = 0; attempt < 600 && !ok; ++attempt) {
      ok = connect(fd, (sockaddr*)&a, sizeof(a)) == 0;
      if (!ok) {
        timespec ts = {0, 100 * 1000 * 1000};
        nanosleep(&ts, nullptr);
      }
    }
    int32_t me = rank;
    if (!ok || !write_all(fd, &me, sizeof(me))) return fail(ncclSystemError, "cannot reach %s", a.sun_path);
    c->fd[p] = fd;
  }
  for (int k = rank + 1; k < nranks; ++k) {
    pollfd pf = {c->listen_fd, POLLIN, 0};
    if (poll(&pf, 1, kTimeoutMs) <= 0) return fail(ncclSystemError, "rank %s%lld: a higher rank never connected", "", rank);
    int fd = accept(c->listen_fd, nullptr, nullptr);
    int32_t who = -1;
    if (fd < 0 || !read_all(fd, &who, sizeof(who)) || who <= rank || who >= nranks || c->fd[who] >= 0)
      return fail(ncclSystemError, "bad connection while building the mesh%s", "");
    c->fd[who] = fd;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  for (int fd : c->fd)
    if (fd >= 0) close(fd);
  if (c->listen_fd >= 0) close(c->listen_fd);
  char path[200];
  snprintf(path, sizeof(path), "%s/r%d", c->dir.c_str(), c->rank);
  unlink(path);
  rmdir(c->dir.c_str());  // succeeds for the last rank to leave
  delete c;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  if (r == ncclSuccess) return "no error";
  return g_detail[0] ? g_detail : "rccl stand-in error";
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart%s", "");
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s) {
  const size_t es = type_size(type);
  if (!es) return fail(ncclInvalidArgument, "unsupported data type%s", "");
  return enqueue(Op{true, const_cast<void*>(buf), count * es, peer, comm, s});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s) {
  const size_t es = type_size(type);
  if (!es) return fail(ncclInvalidArgument, "unsupported data type%s", "");
  return enqueue(Op{false, buf, count * es, peer, comm, s});
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm,
                           hipStream_t s) {
  if (g_depth > 0) return fail(ncclInvalidUsage, "collectives inside a group are not modelled%s", "");
  const size_t bytes = count * type_size(type);
  if (hipStreamSynchronize(s) != hipSuccess) return fail(ncclUnhandledCudaError, "stream synchronise failed%s", "");
  char* own = (char*)recv + (size_t)comm->rank * bytes;
  if (own != send && bytes && hipMemcpy(own, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess)
    return fail(ncclUnhandledCudaError, "copy of the own block failed%s", "");
  std::vector<Op> ops;
  for (int p = 0; p < comm->n; ++p) {
    if (p == comm->rank) continue;
    ops.push_back(Op{true, const_cast<void*>(send), bytes, p, comm, s});
    ops.push_back(Op{false, (char*)recv + (size_t)p * bytes, bytes, p, comm, s});
  }
  return run(ops);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t s) {
  if (g_depth > 0) return fail(ncclInvalidUsage, "collectives inside a group are not modelled%s", "");
  if (type != ncclFloat64 && type != ncclFloat32) return fail(ncclInvalidArgument, "all-reduce of floats and doubles only%s", "");
  const size_t bytes = count * type_size(type);
  if (hipStreamSynchronize(s) != hipSuccess) return fail(ncclUnhandledCudaError, "stream synchronise failed%s", "");
  // every rank's contribution, staged on the device behind one another, then combined in rank order on the host
  void* stage = nullptr;
  if (hipMalloc(&stage, bytes * comm->n) != hipSuccess) return fail(ncclUnhandledCudaError, "staging allocation failed%s", "");
  ncclResult_t rc = ncclAllGather(send, stage, count, type, comm, s);
  std::vector<char> host(bytes * comm->n);
  if (rc == ncclSuccess && hipMemcpy(host.data(), stage, host.size(), hipMemcpyDeviceToHost) != hipSuccess)
    rc = fail(ncclUnhandledCudaError, "copy of the gathered contributions failed%s", "");
  (void)hipFree(stage);
  if (rc != ncclSuccess) return rc;
  for (int p = 1; p < comm->n; ++p) {
    if (type == ncclFloat64) reduce_into((double*)host.data(), (const double*)(host.data() + p * bytes), count, op);
    else reduce_into((float*)host.data(), (const float*)(host.data() + p * bytes), count, op);
  }
  if (hipMemcpy(recv, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
    return fail(ncclUnhandledCudaError, "copy of the result failed%s", "");
  return ncclSuccess;
}

}  // extern "C"
