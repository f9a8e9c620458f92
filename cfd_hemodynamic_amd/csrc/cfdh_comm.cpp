// Inter-GPU exchange of libcfdh.so: forward halo of the iterate (owner -> ghost,
// the reference's ghostUpdate(INSERT, FORWARD), stabilized_schur.py:137-142,168)
// and the Krylov scalar reductions (MPI_Allreduce inside PETSc's KSP), on RCCL
// over xGMI.  One process per GPU; RCCL is bound at run time with dlopen so that
// the single-GPU library has no link-time dependency on it.  A host-callback
// path (cfdh_comm_set_callbacks) carries the same traffic through the launcher
// (torch.distributed/gloo) for tests.
#include <dlfcn.h>

#include "cfdh_internal.hpp"

namespace {
struct nccl_uid { char internal[128]; };
typedef void *nccl_comm_t;
enum { NCCL_FLOAT64 = 8 };
enum { NCCL_SUM = 0, NCCL_MAX = 2 };
struct NcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(nccl_uid *) = nullptr;
  int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
NcclApi g_nccl;

bool load_nccl(std::string &why) {
  if (g_nccl.lib) return true;
  // reuse an RCCL the process already holds (PyTorch ships one, built against the HIP
  // runtime that is then also ours) before loading the ROCm one
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  // CFDH_RCCL_LIB selects a specific build of the library (the tests point it at a shared-memory stand-in
  // so that this file's RCCL branch runs with several ranks on the one-GPU development box)
  if (const char *e = getenv("CFDH_RCCL_LIB")) {
    h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
    if (!h) { why = std::string("CFDH_RCCL_LIB: cannot dlopen ") + e + ": " + dlerror(); return false; }
  }
  if (!h)
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (h) break; }
  if (!h)
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) { why = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
#define SYM(field, name)                                          \
  *(void **)(&g_nccl.field) = dlsym(h, name);                     \
  if (!g_nccl.field) { why = std::string("missing symbol ") + name; return false; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  g_nccl.lib = h;
  return true;
}
}  // namespace

#define NCCLCHK(c, call)                                                                                      \
  do {                                                                                                        \
    int r_ = (call);                                                                                          \
    if (r_ != 0) return cfdh_fail((c), CFDH_E_COMM, "%s: %s", #call, g_nccl.GetErrorString ? g_nccl.GetErrorString(r_) : "?"); \
  } while (0)

extern "C" int cfdh_comm_unique_id(void *id128) {
  std::string why;
  if (!id128) return cfdh_fail(nullptr, CFDH_E_ARG, "null id buffer");
  if (!load_nccl(why)) return cfdh_fail(nullptr, CFDH_E_COMM, "%s", why.c_str());
  nccl_uid id;
  int r = g_nccl.GetUniqueId(&id);
  if (r != 0) return cfdh_fail(nullptr, CFDH_E_COMM, "ncclGetUniqueId: %s", g_nccl.GetErrorString(r));
  memcpy(id128, &id, sizeof id);
  return 0;
}

static int global_counts(cfdh_ctx *c) {
  double cnt = (double)c->nvo;
  HIPCHK(c, hipMemcpyAsync(c->red_out.p + 16, &cnt, sizeof(double), hipMemcpyHostToDevice, c->stream));
  CHK(comm_allreduce_dev(c, c->red_out.p + 16, 1, 0));
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->red_out.p + 16, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->nvo_global = c->h_pinned[0];
  return 0;
}

extern "C" int cfdh_comm_init(cfdh_ctx *c, const void *id128, int rank, int nranks) {
  if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return cfdh_fail(c, CFDH_E_ARG, "bad comm arguments");
  std::string why;
  if (!load_nccl(why)) return cfdh_fail(c, CFDH_E_COMM, "%s", why.c_str());
  HIPCHK(c, hipSetDevice(c->device));
  nccl_uid id;
  memcpy(&id, id128, sizeof id);
  nccl_comm_t comm = nullptr;
  NCCLCHK(c, g_nccl.CommInitRank(&comm, nranks, id, rank));
  c->nccl_comm = comm;
  c->rank = rank; c->nranks = nranks;
  c->cb_ar = nullptr; c->cb_ex = nullptr;
  return global_counts(c);
}

extern "C" int cfdh_comm_set_callbacks(cfdh_ctx *c, cfdh_allreduce_fn ar, cfdh_exchange_fn ex, void *user, int rank, int nranks) {
  if (!c || !ar || !ex || nranks < 1) return cfdh_fail(c, CFDH_E_ARG, "bad comm callbacks");
  if (c->nccl_comm) comm_finalize(c);  // the callbacks replace an RCCL communicator
  if (c->gp_allgather) { c->gp_allgather = false; c->pc_graph_valid = false; }  // back to the all-reduce of the padded vector
  c->cb_ar = ar; c->cb_ex = ex; c->cb_user = user;
  c->rank = rank; c->nranks = nranks;
  return global_counts(c);
}

int comm_finalize(cfdh_ctx *c) {
  if (c->nccl_comm && g_nccl.CommDestroy) g_nccl.CommDestroy((nccl_comm_t)c->nccl_comm);
  c->nccl_comm = nullptr;
  return 0;
}

// in-stream reduction of n doubles at `dev` over all ranks; op 0 sum, 1 max
int comm_allreduce_dev(cfdh_ctx *c, double *dev, int n, int op) {
  if (c->nranks <= 1) return 0;
  c->n_allreduce++;
  if (c->nccl_comm) {
    NCCLCHK(c, g_nccl.AllReduce(dev, dev, (size_t)n, NCCL_FLOAT64, op == 0 ? NCCL_SUM : NCCL_MAX, (nccl_comm_t)c->nccl_comm, c->stream));
    return 0;
  }
  if (!c->cb_ar) return cfdh_fail(c, CFDH_E_COMM, "multi-rank context without communicator");
  double *h = c->h_pinned + 512;
  if (n > 512) {
    if (c->h_big_n < (size_t)n) {
      if (c->h_big) (void)hipHostFree(c->h_big);
      HIPCHK(c, hipHostMalloc((void **)&c->h_big, sizeof(double) * (size_t)n));
      c->h_big_n = (size_t)n;
    }
    h = c->h_big;
  }
  HIPCHK(c, hipMemcpyAsync(h, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->cb_ar(c->cb_user, h, n, op) != 0) return cfdh_fail(c, CFDH_E_COMM, "allreduce callback failed");
  HIPCHK(c, hipMemcpyAsync(dev, h, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // h is reused by the next call
  return 0;
}

// recv[r * count + i] = send_r[i] for every rank r (in-stream)
int comm_allgather_dev(cfdh_ctx *c, const double *send, double *recv, int count) {
  if (!c->nccl_comm) return cfdh_fail(c, CFDH_E_COMM, "all-gather needs an RCCL communicator");
  c->n_allgather++;
  NCCLCHK(c, g_nccl.AllGather(send, recv, (size_t)count, NCCL_FLOAT64, (nccl_comm_t)c->nccl_comm, c->stream));
  return 0;
}

// fill the ghost tail of `vec` (layout [u owned | p owned | (ux,uy,p) per ghost]) from the owners
int comm_halo(cfdh_ctx *c, double *vec) {
  if (c->nranks <= 1 || c->ng == 0) return 0;
  if (c->nnbr == 0) return cfdh_fail(c, CFDH_E_STATE, "ghost vertices present but cfdh_set_halo was not called");
  CHK(k_halo_pack(c, vec));
  c->n_halo++;
  const size_t W = (size_t)c->dim + 1;  // doubles per vertex record: (ux, uy[, uz], p)
  double *tail = vec + W * (size_t)c->nvo;
  if (c->nccl_comm) {
    NCCLCHK(c, g_nccl.GroupStart());
    // an error inside the group must not leave it open: remember the first one, always close, then report
    int first = 0;
    const char *what = "";
    for (int k = 0; k < c->nnbr && !first; k++) {
      const size_t ns = (size_t)(c->send_ptr[k + 1] - c->send_ptr[k]), nr = (size_t)(c->recv_ptr[k + 1] - c->recv_ptr[k]);
      if (ns && (first = g_nccl.Send(c->send_buf.p + W * c->send_ptr[k], W * ns, NCCL_FLOAT64, c->nbr_rank[k], (nccl_comm_t)c->nccl_comm, c->stream))) { what = "ncclSend"; break; }
      if (nr && (first = g_nccl.Recv(tail + W * c->recv_ptr[k], W * nr, NCCL_FLOAT64, c->nbr_rank[k], (nccl_comm_t)c->nccl_comm, c->stream))) { what = "ncclRecv"; break; }
    }
    const int endrc = g_nccl.GroupEnd();
    if (first) return cfdh_fail(c, CFDH_E_COMM, "%s (halo exchange): %s", what, g_nccl.GetErrorString(first));
    if (endrc) return cfdh_fail(c, CFDH_E_COMM, "ncclGroupEnd (halo exchange): %s", g_nccl.GetErrorString(endrc));
    return 0;
  }
  if (!c->cb_ex) return cfdh_fail(c, CFDH_E_COMM, "multi-rank context without communicator");
  const size_t nsend = W * (size_t)c->send_ptr[c->nnbr], nrecv = W * (size_t)c->recv_ptr[c->nnbr];
  c->h_send.resize(nsend); c->h_recv.resize(nrecv);
  if (nsend) HIPCHK(c, hipMemcpyAsync(c->h_send.data(), c->send_buf.p, sizeof(double) * nsend, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->cb_ex(c->cb_user, c->h_send.data(), c->h_recv.data()) != 0) return cfdh_fail(c, CFDH_E_COMM, "exchange callback failed");
  if (nrecv) HIPCHK(c, hipMemcpyAsync(tail, c->h_recv.data(), sizeof(double) * nrecv, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int cfdh_set_halo(cfdh_ctx *c, int nnbr, const int32_t *nbr_rank, const int64_t *send_ptr, const int32_t *send_idx,
                             const int64_t *recv_ptr, const int32_t *recv_idx) {
  if (!c || nnbr < 0) return cfdh_fail(c, CFDH_E_ARG, "bad halo arguments");
  c->nnbr = nnbr;
  c->rasp.ready = false;  // the ghost-row pattern follows the halo plan
  c->nbr_rank.assign(nbr_rank, nbr_rank + nnbr);
  c->send_ptr.assign(send_ptr, send_ptr + nnbr + 1);
  c->recv_ptr.assign(recv_ptr, recv_ptr + nnbr + 1);
  if (c->recv_ptr[nnbr] != c->ng) return cfdh_fail(c, CFDH_E_ARG, "halo plan covers %lld ghosts, mesh has %d", c->recv_ptr[nnbr], c->ng);
  for (long long i = 0; i < c->recv_ptr[nnbr]; i++)
    if (recv_idx[i] != c->nvo + (int)i) return cfdh_fail(c, CFDH_E_ARG, "ghosts must be numbered contiguously per neighbour from nv_owned");
  std::vector<int> sidx((size_t)c->send_ptr[nnbr]);
  for (size_t i = 0; i < sidx.size(); i++) {
    if (send_idx[i] < 0 || send_idx[i] >= c->nvo) return cfdh_fail(c, CFDH_E_ARG, "send index is not an owned vertex");
    sidx[i] = c->perm[send_idx[i]];
  }
  HIPCHK(c, c->send_idx.upload(sidx, c->stream));
  HIPCHK(c, c->send_buf.alloc(((size_t)c->dim + 1) * sidx.size() + 4));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
