// TEST INFRASTRUCTURE -- a stand-in for librccl.so that moves the data through POSIX shared memory.
//
// One MI355X is all the development box has, and RCCL refuses two ranks on one device, so the RCCL branch
// of cfd_hemodynamic_amd/csrc/cfdh_comm.cpp (dlopen + ncclCommInitRank, the grouped ncclSend/ncclRecv halo,
// the in-stream ncclAllReduce with its counts, offsets and enum values) could otherwise never run with more
// than one rank before the multi-GPU node does.  This library exports the ten NCCL entry points libcfdh
// binds and implements them for processes of ONE host: buffers are staged device -> shared memory -> device
// with the semantics of the real calls (stream-ordered, p2p matched per (src, dst) in issue order,
// all-reduce in rank order).  Selected with CFDH_RCCL_LIB=<this .so>; never loaded by the product otherwise.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr size_t RED_DOUBLES = 4u << 20;    // per-rank all-reduce staging (32 MiB)
constexpr size_t BOX_DOUBLES = 128u << 10;  // per (src,dst) mailbox (1 MiB)
struct Header {
  std::atomic<int> arrived, generation;
};
struct Box {
  std::atomic<long> seq, ack;
  long count;
};
struct Comm {
  int rank, nranks;
  char name[64];
  size_t bytes;
  unsigned char *base;
  Header *hdr;
  Box *boxes;        // [src][dst]
  double *boxdata;   // [src][dst][BOX_DOUBLES]
  double *red;       // [rank][RED_DOUBLES]
  std::vector<long> sent, recvd;  // per peer message counters
};
struct Op { int kind; void *ptr; size_t count; int peer; Comm *comm; hipStream_t stream; };
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;

void barrier(Comm *c) {
  const int gen = c->hdr->generation.load(std::memory_order_acquire);
  if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
    c->hdr->arrived.store(0, std::memory_order_relaxed);
    c->hdr->generation.store(gen + 1, std::memory_order_release);
  } else {
    while (c->hdr->generation.load(std::memory_order_acquire) == gen) usleep(20);
  }
}
Box &box(Comm *c, int src, int dst) { return c->boxes[(size_t)src * c->nranks + dst]; }
double *boxbuf(Comm *c, int src, int dst) { return c->boxdata + ((size_t)src * c->nranks + dst) * BOX_DOUBLES; }

int run_ops(std::vector<Op> &ops) {
  // sends first (they never block on the receiver beyond the previous message), then receives
  for (Op &o : ops) {
    if (o.kind != 0) continue;
    Comm *c = o.comm;
    if (o.count > BOX_DOUBLES) return 5;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    Box &b = box(c, c->rank, o.peer);
    const long k = ++c->sent[o.peer];
    while (b.ack.load(std::memory_order_acquire) != k - 1) usleep(20);
    if (hipMemcpy(boxbuf(c, c->rank, o.peer), o.ptr, o.count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    b.count = (long)o.count;
    b.seq.store(k, std::memory_order_release);
  }
  for (Op &o : ops) {
    if (o.kind != 1) continue;
    Comm *c = o.comm;
    Box &b = box(c, o.peer, c->rank);
    const long k = ++c->recvd[o.peer];
    while (b.seq.load(std::memory_order_acquire) != k) usleep(20);
    if (b.count != (long)o.count) { fprintf(stderr, "[fake rccl] rank %d: recv of %zu from %d matched a send of %ld\n", c->rank, o.count, o.peer, b.count); return 5; }
    if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    if (hipMemcpy(o.ptr, boxbuf(c, o.peer, c->rank), o.count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
    b.ack.store(k, std::memory_order_release);
  }
  ops.clear();
  return 0;
}
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/cfdh_fake_rccl_%d_%ld", (int)getpid(), (long)random());
  return 0;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return 4;
  Comm *c = new Comm();
  c->rank = rank; c->nranks = nranks;
  snprintf(c->name, sizeof c->name, "%s", id.internal);
  const size_t nb = (size_t)nranks * nranks;
  c->bytes = 4096 + nb * sizeof(Box) + nb * BOX_DOUBLES * sizeof(double) + (size_t)nranks * RED_DOUBLES * sizeof(double);
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return 2; }
  c->base = (unsigned char *)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) { delete c; return 2; }
  c->hdr = (Header *)c->base;                       // fresh shm pages are zero: counters start at 0
  c->boxes = (Box *)(c->base + 4096);
  c->boxdata = (double *)(c->base + 4096 + nb * sizeof(Box));
  c->red = c->boxdata + nb * BOX_DOUBLES;
  c->sent.assign(nranks, 0); c->recvd.assign(nranks, 0);
  barrier(c);
  *comm = c;
  return 0;
}

int ncclCommDestroy(void *comm) {
  Comm *c = (Comm *)comm;
  if (!c) return 0;
  munmap(c->base, c->bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream) {
  Comm *c = (Comm *)comm;
  if (dtype != 8 /* ncclFloat64 */ || (op != 0 /* sum */ && op != 2 /* max */)) return 4;
  if (count > RED_DOUBLES) return 5;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  double *mine = c->red + (size_t)c->rank * RED_DOUBLES;
  if (hipMemcpy(mine, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  barrier(c);
  std::vector<double> acc(count);
  for (size_t i = 0; i < count; i++) acc[i] = c->red[i];
  for (int r = 1; r < c->nranks; r++) {
    const double *o = c->red + (size_t)r * RED_DOUBLES;
    for (size_t i = 0; i < count; i++) acc[i] = (op == 0) ? acc[i] + o[i] : (acc[i] > o[i] ? acc[i] : o[i]);
  }
  barrier(c);  // everybody has read before anybody stages the next call
  if (hipMemcpy(recv, acc.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
  Comm *c = (Comm *)comm;
  if (dtype != 8) return 4;
  if (count > RED_DOUBLES) return 5;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (hipMemcpy(c->red + (size_t)c->rank * RED_DOUBLES, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  barrier(c);
  for (int r = 0; r < c->nranks; r++)
    if (hipMemcpy((double *)recv + (size_t)r * count, c->red + (size_t)r * RED_DOUBLES, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
  barrier(c);
  return 0;
}

int ncclGroupStart() { g_group++; return 0; }
int ncclGroupEnd() {
  if (--g_group > 0) return 0;
  return run_ops(g_ops);
}
static int p2p(int kind, void *ptr, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
  Comm *c = (Comm *)comm;
  if (dtype != 8 || peer < 0 || peer >= c->nranks || peer == c->rank) return 4;
  g_ops.push_back(Op{kind, ptr, count, peer, c, stream});
  return g_group > 0 ? 0 : run_ops(g_ops);
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
  return p2p(0, (void *)buf, count, dtype, peer, comm, stream);
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
  return p2p(1, buf, count, dtype, peer, comm, stream);
}
const char *ncclGetErrorString(int r) {
  switch (r) {
    case 0: return "success";
    case 1: return "unhandled hip error (fake rccl)";
    case 2: return "system error (fake rccl: shared memory)";
    case 4: return "invalid argument (fake rccl)";
    case 5: return "invalid usage (fake rccl: message too large or mismatched)";
    default: return "error (fake rccl)";
  }
}
}
