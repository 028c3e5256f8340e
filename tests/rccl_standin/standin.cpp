// Test infrastructure: the handful of RCCL entry points libsophtmi calls, carried over Unix domain sockets between
// rank processes that SHARE one GPU.  RCCL itself refuses two ranks on one device, so on a one-GPU box the library's
// RCCL branches (peer ranks, byte counts, offsets, the order of sends and receives inside a group, all-gather and
// all-reduce, and the ordering of all that against the kernels on two streams) would never run; linked against this
// stand-in instead of librccl they do, under the same tests as the host-staged transport.
//
// Like RCCL, every call only QUEUES work on the stream it names and returns: a device-to-host copy of the send
// buffers into pinned memory, a host function that moves the bytes over the sockets, a host-to-device copy into the
// receive buffers.  Operations of one communicator run in the order they were issued, whatever their streams (RCCL
// serialises them the same way).  So a kernel that reads a receive buffer without being ordered behind the
// operation reads stale bytes here as it would there - and SMI_STANDIN_DELAY_MS=<n> holds every transfer back by n
// milliseconds to make such a race lose reliably.
//
// What it checks beyond moving the bytes: every message carries its length, and a receive whose length differs from
// what the sender sent, or that no send answers within the time limit, ends the process with a message (real RCCL
// would hang or corrupt memory).
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
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Job;
}

struct ncclComm {
  int rank = 0, n = 0;
  std::string dir;
  int listen_fd = -1;
  std::vector<int> fd;  // one stream socket per peer (fd[rank] = -1)
  // issue order = execution order
  std::mutex m;
  std::condition_variable cv;
  uint64_t next_issue = 0, next_run = 0;
  std::vector<Job*> jobs;  // queued or finished, recycled by the issuing thread
  // pinned staging buffers are reused and only freed with the communicator: hipHostFree synchronises the device,
  // which would order the very streams whose ordering the tests are about
  std::vector<std::pair<size_t, char*>> pool;
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

struct Wire {  // one message on one socket: 16-byte header, then the payload; lives in pinned host memory
  char* buf = nullptr;
  size_t cap = 0;  // capacity of buf (it may come from the pool, larger than this message)
  size_t size = 0, done = 0, expect = 0;
  int peer = -1;
  bool send = false;
  void* device = nullptr;
};

struct Job {
  ncclComm* comm = nullptr;
  uint64_t ticket = 0;
  std::vector<Wire> wires;  // in call order
  // all-reduce: every rank's contribution in rank order, and their combination
  char* gathered = nullptr;
  char* result = nullptr;
  size_t gathered_cap = 0, result_cap = 0;
  size_t count = 0, elem = 0;
  ncclRedOp_t op = ncclSum;
  std::atomic<int> finished{0};
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local char g_detail[512] = "";

int delay_ms() {
  static const int d = getenv("SMI_STANDIN_DELAY_MS") ? atoi(getenv("SMI_STANDIN_DELAY_MS")) : 0;
  return d;
}

ncclResult_t fail(ncclResult_t code, const char* what) {
  snprintf(g_detail, sizeof(g_detail), "%s", what);
  fprintf(stderr, "[rccl stand-in] %s\n", g_detail);
  return code;
}

[[noreturn]] void die(const ncclComm* c, const char* what, long long a = 0, long long b = 0) {
  fprintf(stderr, "[rccl stand-in] rank %d: ", c->rank);
  fprintf(stderr, what, a, b);
  fprintf(stderr, "\n");
  fflush(stderr);
  _exit(86);
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
    if (k < 0 && errno == EINTR) continue;
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
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return false;
    c += k;
    n -= (size_t)k;
  }
  return true;
}

// Moves every message of a job; per peer the sends go out and the receives come in in call order.  Host only.
void progress(Job* job) {
  ncclComm* comm = job->comm;
  const int n = comm->n;
  std::vector<std::vector<Wire*>> sendq(n), recvq(n);
  for (Wire& w : job->wires) (w.send ? sendq : recvq)[w.peer].push_back(&w);
  std::vector<size_t> si(n, 0), ri(n, 0);
  for (int p = 0; p < n; ++p)
    if (comm->fd[p] >= 0) fcntl(comm->fd[p], F_SETFL, fcntl(comm->fd[p], F_GETFL) | O_NONBLOCK);
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
    if (ready == 0)
      die(comm, "operation %lld: no progress for 60 s towards rank %lld (a receive without a matching send, or the reverse)",
          (long long)job->ticket, who[0]);
    if (ready < 0) {
      if (errno == EINTR) continue;
      die(comm, "poll failed");
    }
    for (size_t i = 0; i < pf.size(); ++i) {
      const int p = who[i];
      if (pf[i].revents & (POLLERR | POLLNVAL)) die(comm, "socket error towards rank %lld", p);
      if ((pf[i].revents & POLLOUT) && si[p] < sendq[p].size()) {
        Wire& w = *sendq[p][si[p]];
        ssize_t k = write(pf[i].fd, w.buf + w.done, w.size - w.done);
        if (k > 0) w.done += (size_t)k;
        else if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) die(comm, "write to rank %lld failed", p);
        if (w.done == w.size) ++si[p];
      }
      if ((pf[i].revents & (POLLIN | POLLHUP)) && ri[p] < recvq[p].size()) {
        Wire& w = *recvq[p][ri[p]];
        // header first: the payload buffer is sized for what the caller expects
        const size_t want = w.done < 16 ? 16 - w.done : w.size - w.done;
        ssize_t k = read(pf[i].fd, w.buf + w.done, want);
        if (k > 0) {
          w.done += (size_t)k;
          if (w.done == 16) {
            uint64_t head[2];
            memcpy(head, w.buf, 16);
            if (head[0] != kMagic) die(comm, "bad message header from rank %lld", p);
            if (head[1] != w.expect)
              die(comm, "size mismatch: rank %lld sent %lld bytes where this receive expects another count", p,
                  (long long)head[1]);
          }
        } else if (k == 0) {
          die(comm, "rank %lld closed its socket in the middle of operation %lld", p, (long long)job->ticket);
        } else if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) {
          die(comm, "read from rank %lld failed", p);
        }
        if (w.done == w.size) ++ri[p];
      }
    }
  }
  for (int p = 0; p < n; ++p)
    if (comm->fd[p] >= 0) fcntl(comm->fd[p], F_SETFL, fcntl(comm->fd[p], F_GETFL) & ~O_NONBLOCK);
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

// host function on the operation's stream: waits for its turn on the communicator, then moves the bytes
void exchange_cb(void* arg) {
  Job* job = (Job*)arg;
  ncclComm* comm = job->comm;
  {
    std::unique_lock<std::mutex> lock(comm->m);
    comm->cv.wait(lock, [&] { return comm->next_run == job->ticket; });
  }
  if (delay_ms() > 0) {
    timespec ts = {delay_ms() / 1000, (long)(delay_ms() % 1000) * 1000000L};
    nanosleep(&ts, nullptr);
  }
  progress(job);
  {
    std::lock_guard<std::mutex> lock(comm->m);
    ++comm->next_run;
  }
  comm->cv.notify_all();
}

// all-reduce, first host function: the own contribution (just copied to the host) into the outgoing messages
void reduce_fill_cb(void* arg) {
  Job* j = (Job*)arg;
  const size_t b = j->count * j->elem;
  for (Wire& w : j->wires)
    if (w.send) {
      const uint64_t head[2] = {kMagic, (uint64_t)b};
      memcpy(w.buf, head, 16);
      memcpy(w.buf + 16, j->gathered + (size_t)j->comm->rank * b, b);
    }
}

// all-reduce, second host function: exchange, then combine the contributions in rank order
void reduce_cb(void* arg) {
  Job* j = (Job*)arg;
  exchange_cb(j);
  const size_t b = j->count * j->elem;
  for (Wire& w : j->wires)
    if (!w.send) memcpy(j->gathered + (size_t)w.peer * b, w.buf + 16, b);
  memcpy(j->result, j->gathered, b);
  for (int p = 1; p < j->comm->n; ++p) {
    if (j->elem == 8) reduce_into((double*)j->result, (const double*)(j->gathered + p * b), j->count, j->op);
    else reduce_into((float*)j->result, (const float*)(j->gathered + p * b), j->count, j->op);
  }
}

void finished_cb(void* arg) { ((Job*)arg)->finished.store(1); }

hipError_t pinned(ncclComm* comm, char** out, size_t* cap, size_t bytes) {
  size_t best = comm->pool.size();
  for (size_t i = 0; i < comm->pool.size(); ++i)
    if (comm->pool[i].first >= bytes && (best == comm->pool.size() || comm->pool[i].first < comm->pool[best].first)) best = i;
  if (best < comm->pool.size()) {
    *out = comm->pool[best].second;
    *cap = comm->pool[best].first;
    comm->pool[best] = comm->pool.back();
    comm->pool.pop_back();
    return hipSuccess;
  }
  *cap = bytes;
  return hipHostMalloc((void**)out, bytes, hipHostMallocDefault);
}

void release(Job* job) {
  ncclComm* comm = job->comm;
  for (Wire& w : job->wires)
    if (w.buf) comm->pool.push_back({w.cap, w.buf});
  if (job->gathered) comm->pool.push_back({job->gathered_cap, job->gathered});
  if (job->result) comm->pool.push_back({job->result_cap, job->result});
  delete job;
}

void collect_finished(ncclComm* comm) {
  auto& v = comm->jobs;
  for (size_t i = 0; i < v.size();) {
    if (v[i]->finished.load()) {
      release(v[i]);
      v[i] = v.back();
      v.pop_back();
    } else {
      ++i;
    }
  }
}

#define HIP_OK(expr)                                                       \
  do {                                                                     \
    if ((expr) != hipSuccess) return fail(ncclUnhandledCudaError, #expr);  \
  } while (0)

Job* new_job(ncclComm* comm) {
  collect_finished(comm);
  Job* job = new Job();
  job->comm = comm;
  job->ticket = comm->next_issue++;
  comm->jobs.push_back(job);
  return job;
}

// SMI_STANDIN_ALONE=1: one process plays ONE rank of a larger job and nothing travels - every send and receive is
// accepted and dropped (received buffers keep what they held).  The library then runs its RCCL branches for that
// rank (streams, events, groups, launch order) at the rank's share of the grid, which is how
// tools/rank_local_step.py times the per-rank kernels of a multi-GPU step on one GPU.  Results are meaningless.
bool alone() {
  static const bool on = getenv("SMI_STANDIN_ALONE") && atoi(getenv("SMI_STANDIN_ALONE")) != 0;
  return on;
}

// queues one group of sends and receives on their stream
ncclResult_t submit(std::vector<Op>& ops) {
  if (ops.empty() || alone()) return ncclSuccess;
  ncclComm* comm = ops[0].comm;
  hipStream_t s = ops[0].stream;
  for (const Op& o : ops) {
    if (o.comm != comm) return fail(ncclInvalidUsage, "one group, two communicators");
    if (o.stream != s) return fail(ncclInvalidUsage, "one group, two streams: not modelled");
    if (o.peer < 0 || o.peer >= comm->n || o.peer == comm->rank)
      return fail(ncclInvalidArgument, "a peer that is not another rank of this communicator");
  }
  Job* job = new_job(comm);
  job->wires.resize(ops.size());
  for (size_t i = 0; i < ops.size(); ++i) {
    Wire& w = job->wires[i];
    const Op& o = ops[i];
    w.size = 16 + o.bytes;
    w.expect = o.bytes;
    w.peer = o.peer;
    w.send = o.send;
    w.device = o.ptr;
    HIP_OK(pinned(comm, &w.buf, &w.cap, w.size));
    if (o.send) {
      const uint64_t head[2] = {kMagic, (uint64_t)o.bytes};
      memcpy(w.buf, head, 16);
      if (o.bytes) HIP_OK(hipMemcpyAsync(w.buf + 16, o.ptr, o.bytes, hipMemcpyDeviceToHost, s));
    }
  }
  HIP_OK(hipLaunchHostFunc(s, exchange_cb, job));
  for (Wire& w : job->wires)
    if (!w.send && w.expect) HIP_OK(hipMemcpyAsync(w.device, w.buf + 16, w.expect, hipMemcpyHostToDevice, s));
  HIP_OK(hipLaunchHostFunc(s, finished_cb, job));
  return ncclSuccess;
}

ncclResult_t enqueue(const Op& op) {
  g_ops.push_back(op);
  if (g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return submit(ops);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  FILE* f = fopen("/dev/urandom", "rb");
  unsigned char raw[8] = {0};
  if (!f || fread(raw, 1, sizeof(raw), f) != sizeof(raw)) {
    if (f) fclose(f);
    return fail(ncclSystemError, "cannot read /dev/urandom");
  }
  fclose(f);
  char* out = id->internal;
  for (int i = 0; i < 8; ++i) snprintf(out + 2 * i, 3, "%02x", raw[i]);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "bad rank / size");
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->n = nranks;
  c->fd.assign(nranks, -1);
  char tag[17];
  memcpy(tag, id.internal, 16);
  tag[16] = 0;
  c->dir = std::string("/tmp/smi_rccl_standin_") + tag;
  c->listen_fd = -1;
  if (alone()) {
    *out = c;
    return ncclSuccess;
  }
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
    return fail(ncclSystemError, "cannot listen on the rank's socket");
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
    if (!ok || !write_all(fd, &me, sizeof(me))) return fail(ncclSystemError, "cannot reach a lower rank's socket");
    c->fd[p] = fd;
  }
  for (int k = rank + 1; k < nranks; ++k) {
    pollfd pf = {c->listen_fd, POLLIN, 0};
    if (poll(&pf, 1, kTimeoutMs) <= 0) return fail(ncclSystemError, "a higher rank never connected");
    int fd = accept(c->listen_fd, nullptr, nullptr);
    int32_t who = -1;
    if (fd < 0 || !read_all(fd, &who, sizeof(who)) || who <= rank || who >= nranks || c->fd[who] >= 0)
      return fail(ncclSystemError, "bad connection while building the mesh");
    c->fd[who] = fd;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  (void)hipDeviceSynchronize();  // every queued operation has run
  for (Job* job : c->jobs) release(job);
  c->jobs.clear();
  for (auto& entry : c->pool) (void)hipHostFree(entry.second);
  c->pool.clear();
  for (int fd : c->fd)
    if (fd >= 0) close(fd);
  if (c->listen_fd >= 0) close(c->listen_fd);
  if (!alone()) {
    char path[200];
    snprintf(path, sizeof(path), "%s/r%d", c->dir.c_str(), c->rank);
    unlink(path);
    rmdir(c->dir.c_str());  // succeeds for the last rank to leave
  }
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
  if (g_depth <= 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return submit(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s) {
  const size_t es = type_size(type);
  if (!es) return fail(ncclInvalidArgument, "unsupported data type");
  return enqueue(Op{true, const_cast<void*>(buf), count * es, peer, comm, s});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s) {
  const size_t es = type_size(type);
  if (!es) return fail(ncclInvalidArgument, "unsupported data type");
  return enqueue(Op{false, buf, count * es, peer, comm, s});
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm,
                           hipStream_t s) {
  if (g_depth > 0) return fail(ncclInvalidUsage, "collectives inside a group are not modelled");
  const size_t bytes = count * type_size(type);
  char* own = (char*)recv + (size_t)comm->rank * bytes;
  if (own != send && bytes) HIP_OK(hipMemcpyAsync(own, send, bytes, hipMemcpyDeviceToDevice, s));
  std::vector<Op> ops;
  for (int p = 0; p < comm->n; ++p) {
    if (p == comm->rank) continue;
    ops.push_back(Op{true, const_cast<void*>(send), bytes, p, comm, s});
    ops.push_back(Op{false, (char*)recv + (size_t)p * bytes, bytes, p, comm, s});
  }
  return submit(ops);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t s) {
  if (g_depth > 0) return fail(ncclInvalidUsage, "collectives inside a group are not modelled");
  if (type != ncclFloat64 && type != ncclFloat32) return fail(ncclInvalidArgument, "all-reduce of floats and doubles only");
  const size_t elem = type_size(type), bytes = count * elem;
  if (alone()) {
    if (recv != send && bytes) HIP_OK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
    return ncclSuccess;
  }
  Job* job = new_job(comm);
  job->count = count;
  job->elem = elem;
  job->op = op;
  HIP_OK(pinned(comm, &job->gathered, &job->gathered_cap, bytes * comm->n));
  HIP_OK(pinned(comm, &job->result, &job->result_cap, bytes));
  job->wires.resize(2 * (size_t)(comm->n - 1));
  size_t i = 0;
  for (int p = 0; p < comm->n; ++p) {
    if (p == comm->rank) continue;
    for (int dir = 0; dir < 2; ++dir) {
      Wire& w = job->wires[i++];
      w.size = 16 + bytes;
      w.expect = bytes;
      w.peer = p;
      w.send = dir == 0;
      HIP_OK(pinned(comm, &w.buf, &w.cap, w.size));
    }
  }
  // own contribution to the host, into the outgoing messages, exchange and combine, result to the device
  HIP_OK(hipMemcpyAsync(job->gathered + (size_t)comm->rank * bytes, send, bytes, hipMemcpyDeviceToHost, s));
  HIP_OK(hipLaunchHostFunc(s, reduce_fill_cb, job));
  HIP_OK(hipLaunchHostFunc(s, reduce_cb, job));
  HIP_OK(hipMemcpyAsync(recv, job->result, bytes, hipMemcpyHostToDevice, s));
  HIP_OK(hipLaunchHostFunc(s, finished_cb, job));
  return ncclSuccess;
}

}  // extern "C"
