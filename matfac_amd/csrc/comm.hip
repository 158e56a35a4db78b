// comm.hip -- multi-GPU item-factor exchange over RCCL (xGMI inside a node).
//
// The reference is single-process OpenMP: nothing to cite.  Design:
// rank g owns a user-row block (its CSR rows and its U shard, never communicated)
// and a full replica of V.  After a local (sub-)epoch every rank holds
// V_g = V_sync + D_g.  MFX_REDUCE_DELTA_SUM forms V <- V_sync + sum_g D_g with ONE
// all-reduce of nItems*ld floats; MFX_REDUCE_AVERAGE forms V <- mean_g V_g.
//
// librccl is loaded lazily with dlopen so that single-GPU use has no RCCL
// dependency; inside a process that already loaded RCCL (e.g. through
// torch.distributed) the same library instance is reused (same SONAME).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>

#include "mfx_internal.h"

namespace {
struct UniqueId { char internal[128]; };
typedef int (*fn_get_uid)(UniqueId*);
typedef int (*fn_init_rank)(void**, int, UniqueId, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_sendrecv)(void*, size_t, int, int, void*, hipStream_t);          // ncclSend (const void*) / ncclRecv
typedef int (*fn_group)();
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_reducescatter)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_sendrecv send = nullptr, recv = nullptr;
  fn_group group_start = nullptr, group_end = nullptr;
  fn_allgather allgather = nullptr;
  fn_reducescatter reducescatter = nullptr;
  fn_errstr errstr = nullptr;
  std::string err;
};
Rccl g_rccl;
constexpr int kNcclFloat = 7, kNcclDouble = 8, kNcclSum = 0;

bool load_rccl() {
  if (g_rccl.h) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.h) break;
  }
  if (!g_rccl.h) {
    const char* e = dlerror();   // one call: dlerror() clears the message it returns
    g_rccl.err = e ? e : "dlopen(librccl) failed";
    return false;
  }
  g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.h, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.h, "ncclCommInitRank");
  g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
  g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.h, "ncclAllReduce");
  g_rccl.errstr = (fn_errstr)dlsym(g_rccl.h, "ncclGetErrorString");
  g_rccl.send = (fn_sendrecv)dlsym(g_rccl.h, "ncclSend");
  g_rccl.recv = (fn_sendrecv)dlsym(g_rccl.h, "ncclRecv");
  g_rccl.group_start = (fn_group)dlsym(g_rccl.h, "ncclGroupStart");
  g_rccl.group_end = (fn_group)dlsym(g_rccl.h, "ncclGroupEnd");
  g_rccl.allgather = (fn_allgather)dlsym(g_rccl.h, "ncclAllGather");
  g_rccl.reducescatter = (fn_reducescatter)dlsym(g_rccl.h, "ncclReduceScatter");
  if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allreduce || !g_rccl.send || !g_rccl.recv || !g_rccl.group_start ||
      !g_rccl.group_end || !g_rccl.allgather || !g_rccl.reducescatter) {
    g_rccl.err = "librccl lacks a required symbol";
    dlclose(g_rccl.h);
    g_rccl.h = nullptr;
    return false;
  }
  return true;
}
const char* rccl_err(int rc) { return g_rccl.errstr ? g_rccl.errstr(rc) : "rccl error"; }
}  // namespace

__global__ void delta_kernel(const float* __restrict__ v, const float* __restrict__ vsync,
                             float* __restrict__ d, int64_t n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride)
    ((f4*)d)[t] = ((const f4*)v)[t] - ((const f4*)vsync)[t];
}
__global__ void apply_delta_kernel(float* __restrict__ v, float* __restrict__ vsync,
                                   const float* __restrict__ dsum, int64_t n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
    const f4 x = ((const f4*)vsync)[t] + ((const f4*)dsum)[t];
    ((f4*)v)[t] = x;
    ((f4*)vsync)[t] = x;
  }
}
__global__ void scale_copy_kernel(float* __restrict__ v, float* __restrict__ vsync, float s, int64_t n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
    const f4 x = ((const f4*)v)[t] * s;
    ((f4*)v)[t] = x;
    ((f4*)vsync)[t] = x;
  }
}

int mfx_comm_free_internal(mfx_ctx* ctx) {
  if (ctx->comm && g_rccl.destroy) g_rccl.destroy(ctx->comm);
  ctx->comm = nullptr;
  ctx->ext_reduce = nullptr;
  ctx->ext_user = nullptr;
  mfx_comm_drop_col_counts(ctx);
  ctx->nranks = 1;
  ctx->rank = 0;
  dev_free(ctx->comm_tmp);
  ctx->comm_tmp_cap = 0;
  if (ctx->ext_stage) (void)hipHostFree(ctx->ext_stage);
  ctx->ext_stage = nullptr;
  ctx->ext_stage_bytes = 0;
  return MFX_OK;
}

// Sum `count` elements (dtype 0 = float, 1 = double) of a device buffer over all ranks, in place, ordered
// on ctx->stream.  RCCL when mfx_comm_init made a communicator; the caller's own all-reduce on a host copy
// when mfx_comm_init_external registered one (MPI / gloo callers, and the two-process tests).
int mfx_comm_allreduce(mfx_ctx* ctx, void* dev, size_t count, int dtype) {
  if (ctx->nranks <= 1 && !ctx->comm && !ctx->ext_reduce) return MFX_OK;
  if (count == 0) return MFX_OK;
  if (ctx->comm) {
    int r = g_rccl.allreduce(dev, dev, count, dtype ? kNcclDouble : kNcclFloat, kNcclSum, ctx->comm, ctx->stream);
    NEED(r == 0, MFX_E_COMM, "ncclAllReduce: %s", rccl_err(r));
    return MFX_OK;
  }
  NEED(ctx->ext_reduce, MFX_E_STATE, "all-reduce requested without a communicator");
  const size_t bytes = count * (dtype ? 8 : 4);
  if (bytes > ctx->ext_stage_bytes) {
    if (ctx->ext_stage) (void)hipHostFree(ctx->ext_stage);
    ctx->ext_stage = nullptr;
    ctx->ext_stage_bytes = 0;
    HIPCHK(hipHostMalloc(&ctx->ext_stage, bytes, hipHostMallocDefault));
    ctx->ext_stage_bytes = bytes;
  }
  HIPCHK(hipMemcpyAsync(ctx->ext_stage, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int r = ctx->ext_reduce(ctx->ext_user, ctx->ext_stage, (int64_t)count, dtype);
  NEED(r == 0, MFX_E_COMM, "external all-reduce returned %d", r);
  HIPCHK(hipMemcpyAsync(dev, ctx->ext_stage, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MFX_OK;
}

// Reduce-scatter (sum, f32) and all-gather of equal slices, RCCL only: the callers fall back to mfx_comm_allreduce on the
// caller-supplied reducer.  In place: rank r's slice of `buf` (count floats at r * count) receives the sum over the ranks /
// is what rank r contributes.
bool mfx_comm_has_rccl(const mfx_ctx* ctx) { return ctx->comm != nullptr; }
int mfx_comm_reduce_scatter(mfx_ctx* ctx, float* buf, size_t count) {
  NEED(ctx->comm, MFX_E_STATE, "reduce-scatter without an RCCL communicator");
  const int r = g_rccl.reducescatter(buf, buf + (size_t)ctx->rank * count, count, kNcclFloat, kNcclSum, ctx->comm, ctx->stream);
  NEED(r == 0, MFX_E_COMM, "ncclReduceScatter: %s", rccl_err(r));
  return MFX_OK;
}
int mfx_comm_allgather(mfx_ctx* ctx, const float* mine, float* all, size_t count) {
  NEED(ctx->comm, MFX_E_STATE, "all-gather without an RCCL communicator");
  const int r = g_rccl.allgather(mine, all, count, kNcclFloat, ctx->comm, ctx->stream);
  NEED(r == 0, MFX_E_COMM, "ncclAllGather: %s", rccl_err(r));
  return MFX_OK;
}

__global__ void bits_checksum_kernel(const uint32_t* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  unsigned s = 0;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x)
    s += x[t] * (unsigned)(2 * (t % 65521) + 1);                    // position-weighted: rows that changed places do not cancel
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
int mfx_comm_check_replicas(mfx_ctx* ctx, const float* buf, size_t n, const char* what) {
  if (ctx->nranks <= 1 || (!ctx->comm && !ctx->ext_reduce) || n == 0) return MFX_OK;
  double v[2] = {0.0, 1.0};
  {
    unsigned* d = nullptr;
    int rc = dev_alloc(ctx, &d, (size_t)1);
    if (rc) return rc;
    unsigned h = 0;
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned), ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(bits_checksum_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const uint32_t*)buf, n, d);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    dev_free(d);
    if (e != hipSuccess) return mfx_fail(ctx, MFX_E_HIP, "replica check: %s", hipGetErrorString(e));
    v[0] = (double)(h & 0x7fffffffu);                               // exact in a double, and so is the sum over <= 2^20 ranks
  }
  const double own = v[0];
  int rc = mfx_allreduce_f64(ctx, v, 2);
  if (rc) return rc;
  if (v[1] != (double)ctx->nranks || v[0] != own * (double)ctx->nranks)
    return mfx_fail(ctx, MFX_E_COMM, "%s: the ranks do not hold the same bytes afterwards (checksum of this rank %.0f, sum over %d ranks %.0f, "
                    "%.0f ranks answered) -- the exchange is broken on this communicator; MFX_ALS_ALLREDUCE=1 / --exchange allreduce take the "
                    "all-reduce paths", what, own, ctx->nranks, v[0], v[1]);
  return MFX_OK;
}

extern "C" int mfx_comm_unique_id(void* id128) {
  mfx_ctx* ctx = nullptr;
  NEED(id128, MFX_E_ARG, "mfx_comm_unique_id: NULL");
  NEED(load_rccl(), MFX_E_COMM, "mfx_comm_unique_id: %s", g_rccl.err.c_str());
  UniqueId id;
  int rc = g_rccl.get_uid(&id);
  NEED(rc == 0, MFX_E_COMM, "ncclGetUniqueId: %s", rccl_err(rc));
  memcpy(id128, &id, sizeof id);
  return MFX_OK;
}

extern "C" int mfx_comm_init(mfx_ctx* ctx, int nranks, int rank, const void* id128) {
  if (ctx) ctx->comm_checked_parts = ctx->comm_checked_als = false;
  if (!ctx) return MFX_E_ARG;
  NEED(nranks >= 1 && rank >= 0 && rank < nranks && id128, MFX_E_ARG, "mfx_comm_init: nranks=%d rank=%d", nranks, rank);
  NEED(load_rccl(), MFX_E_COMM, "mfx_comm_init: %s", g_rccl.err.c_str());
  HIPCHK(hipSetDevice(ctx->device));
  mfx_comm_free_internal(ctx);
  UniqueId id;
  memcpy(&id, id128, sizeof id);
  void* comm = nullptr;
  int rc = g_rccl.init_rank(&comm, nranks, id, rank);
  NEED(rc == 0, MFX_E_COMM, "ncclCommInitRank: %s", rccl_err(rc));
  ctx->comm = comm;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return MFX_OK;
}

__global__ void col_count_kernel(const int64_t* __restrict__ colptr, int32_t ncols, double* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ncols) out[t] = (double)(colptr[t + 1] - colptr[t]);
}
void mfx_comm_drop_col_counts(mfx_ctx* ctx) { dev_free(ctx->gcol); dev_free(ctx->als_global); ctx->als_global_cap = 0; }
int mfx_comm_global_col_counts(mfx_ctx* ctx, const double** out) {
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present && m.has_col, MFX_E_STATE, "global item counts: train matrix with column view needed");
  if (!ctx->gcol) {
    int rc = dev_alloc(ctx, &ctx->gcol, (size_t)m.ncols);
    if (rc) return rc;
    if (m.ncols > 0) {
      hipLaunchKernelGGL(col_count_kernel, dim3((m.ncols + 255) / 256), dim3(256), 0, ctx->stream, m.colptr, m.ncols, ctx->gcol);
      HIPCHK(hipGetLastError());
      if ((rc = mfx_comm_allreduce(ctx, ctx->gcol, (size_t)m.ncols, 1))) { dev_free(ctx->gcol); return rc; }
    }
  }
  *out = ctx->gcol;
  return MFX_OK;
}

extern "C" int mfx_comm_init_external(mfx_ctx* ctx, int nranks, int rank, mfx_reduce_fn fn, void* user) {
  if (ctx) ctx->comm_checked_parts = ctx->comm_checked_als = false;
  if (!ctx) return MFX_E_ARG;
  NEED(nranks >= 1 && rank >= 0 && rank < nranks && fn, MFX_E_ARG, "mfx_comm_init_external: nranks=%d rank=%d", nranks, rank);
  HIPCHK(hipSetDevice(ctx->device));
  mfx_comm_free_internal(ctx);
  ctx->ext_reduce = fn;
  ctx->ext_user = user;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return MFX_OK;
}

extern "C" int mfx_comm_destroy(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  return mfx_comm_free_internal(ctx);
}

// V_sync tracks the last agreed item factors; it is (re)initialised from V the first
// time and after every exchange.
static int ensure_sync_buffers(mfx_ctx* ctx) {
  const size_t n = (size_t)ctx->nI * ctx->ld;      // (comm_tmp is shared with the item-part rotation, which may have sized it larger)
  int rc;
  if (!ctx->comm_tmp || ctx->comm_tmp_cap < n) {   // mfx_set_model may have grown the item table since the last exchange
    dev_free(ctx->comm_tmp);
    ctx->comm_tmp_cap = 0;
    if ((rc = dev_alloc(ctx, &ctx->comm_tmp, n))) return rc;
    ctx->comm_tmp_cap = n;
  }
  if (!ctx->Vsync) {
    if ((rc = dev_alloc(ctx, &ctx->Vsync, n))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->Vsync, ctx->V, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  }
  return MFX_OK;
}

// Call once after mfx_set_factors on every rank (V identical on all ranks) and
// before the first local epoch: snapshots V into V_sync.
extern "C" int mfx_comm_mark_synced(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->V, MFX_E_STATE, "mfx_comm_mark_synced: no model");
  HIPCHK(hipSetDevice(ctx->device));
  dev_free(ctx->Vsync);
  return ensure_sync_buffers(ctx);
}

extern "C" int mfx_allreduce_item_factors(mfx_ctx* ctx, int op) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->V, MFX_E_STATE, "mfx_allreduce_item_factors: no model");
  NEED(op == MFX_REDUCE_DELTA_SUM || op == MFX_REDUCE_AVERAGE, MFX_E_ARG, "mfx_allreduce_item_factors: op=%d", op);
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->comm && !ctx->ext_reduce) return MFX_OK;  // no communicator: single device, V is already the sum
  int rc;
  if ((rc = ensure_sync_buffers(ctx))) return rc;
  const int64_t n = (int64_t)ctx->nI * ctx->ld, n4 = n / 4;
  const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 4096);
  if (op == MFX_REDUCE_DELTA_SUM) {
    hipLaunchKernelGGL(delta_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->V, ctx->Vsync, ctx->comm_tmp, n4);
    HIPCHK(hipGetLastError());
    if ((rc = mfx_comm_allreduce(ctx, ctx->comm_tmp, (size_t)n, 0))) return rc;
    hipLaunchKernelGGL(apply_delta_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->V, ctx->Vsync, ctx->comm_tmp, n4);
  } else {
    if ((rc = mfx_comm_allreduce(ctx, ctx->V, (size_t)n, 0))) return rc;
    hipLaunchKernelGGL(scale_copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->V, ctx->Vsync,
                       1.0f / (float)ctx->nranks, n4);
  }
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

// ---- item-part rotation (include/mfx.h) ----------------------------------------------------------------------------
// rows p, p + N, p + 2N, ... of V <-> a packed slab of `rows` x ld floats
__global__ void part_pack_kernel(const float* __restrict__ V, int64_t nI, int ld4, int part, int nparts, float* __restrict__ slab) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t rows = (nI - part + nparts - 1) / nparts, total = rows * ld4, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t k = t / ld4, c = t % ld4;
    ((f4*)slab)[t] = ((const f4*)V)[((int64_t)part + k * nparts) * ld4 + c];
  }
}
__global__ void part_unpack_kernel(float* __restrict__ V, int64_t nI, int ld4, int part, int nparts, const float* __restrict__ slab) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t rows = (nI - part + nparts - 1) / nparts, total = rows * ld4, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t k = t / ld4, c = t % ld4;
    ((f4*)V)[((int64_t)part + k * nparts) * ld4 + c] = ((const f4*)slab)[t];
  }
}
// comm_tmp holds N + 1 slabs of `slab` floats each (slab = ceil(nI / N) rows x ld: the same size for every part)
static int ensure_part_buffers(mfx_ctx* ctx, int64_t* slab_out) {
  const int N = ctx->nranks;
  const int64_t slab = (((int64_t)ctx->nI + N - 1) / N) * ctx->ld;
  const size_t need = (size_t)slab * ((size_t)N + 1);
  if (!ctx->comm_tmp || ctx->comm_tmp_cap < need) {
    dev_free(ctx->comm_tmp);
    ctx->comm_tmp_cap = 0;
    int rc = dev_alloc(ctx, &ctx->comm_tmp, need);
    if (rc) return rc;
    ctx->comm_tmp_cap = need;
  }
  *slab_out = slab;
  return MFX_OK;
}
static int part_grid(int64_t slab) { return (int)std::min<int64_t>((slab / 4 + 255) / 256 + 1, 4096); }

extern "C" int mfx_rotate_item_part(mfx_ctx* ctx, int send_part, int recv_part) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->V, MFX_E_STATE, "mfx_rotate_item_part: no model");
  const int N = ctx->nranks;
  // one rank keeps every part (MFX_COMM_SELF_TEST=1: the RCCL calls run anyway, rank 0 sending to itself -- tests/test_comm_gpu.py)
  if ((N <= 1 && !(ctx->comm && getenv("MFX_COMM_SELF_TEST"))) || (!ctx->comm && !ctx->ext_reduce)) return MFX_OK;
  NEED(ctx->item_parts == N || N == 1, MFX_E_STATE, "mfx_rotate_item_part: mfx_sgd_set_item_parts(%d) must name as many parts as there are ranks (%d)",
       ctx->item_parts, N);
  NEED(send_part >= 0 && send_part < N && recv_part >= 0 && recv_part < N, MFX_E_ARG, "mfx_rotate_item_part: parts %d, %d of %d", send_part, recv_part, N);
  HIPCHK(hipSetDevice(ctx->device));
  int64_t slab;
  int rc = ensure_part_buffers(ctx, &slab);
  if (rc) return rc;
  const int ld4 = ctx->ld / 4, grid = part_grid(slab);
  float* out = ctx->comm_tmp;                 // slab 0: what leaves; slab 1 (RCCL) or slabs 1 .. N (external reducer): what arrives
  hipLaunchKernelGGL(part_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->V, (int64_t)ctx->nI, ld4, send_part, N, out);
  HIPCHK(hipGetLastError());
  const float* in;
  if (ctx->comm) {
    const int prev = (ctx->rank + N - 1) % N, next = (ctx->rank + 1) % N;
    int r = g_rccl.group_start();
    NEED(r == 0, MFX_E_COMM, "ncclGroupStart: %s", rccl_err(r));
    r = g_rccl.send(out, (size_t)slab, kNcclFloat, prev, ctx->comm, ctx->stream);
    const int r2 = g_rccl.recv(out + slab, (size_t)slab, kNcclFloat, next, ctx->comm, ctx->stream);
    const int r3 = g_rccl.group_end();
    NEED(r == 0 && r2 == 0 && r3 == 0, MFX_E_COMM, "ncclSend / ncclRecv: %s", rccl_err(r ? r : (r2 ? r2 : r3)));
    in = out + slab;
  } else {
    // a caller-supplied all-reduce only (gloo / MPI callers, the two-process tests): every rank writes its slab into the place of
    // its destination rank in an otherwise zero buffer of N slabs; the sum hands every rank the slab meant for it (N times the
    // bytes of a send / receive: the rehearsal path, not the product path)
    float* big = out + slab;
    HIPCHK(hipMemsetAsync(big, 0, sizeof(float) * (size_t)slab * N, ctx->stream));
    HIPCHK(hipMemcpyAsync(big + (size_t)((ctx->rank + N - 1) % N) * slab, out, sizeof(float) * (size_t)slab, hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = mfx_comm_allreduce(ctx, big, (size_t)slab * N, 0))) return rc;
    in = big + (size_t)ctx->rank * slab;
  }
  hipLaunchKernelGGL(part_unpack_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->V, (int64_t)ctx->nI, ld4, recv_part, N, in);
  HIPCHK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_allgather_item_parts(mfx_ctx* ctx, int my_part) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->V, MFX_E_STATE, "mfx_allgather_item_parts: no model");
  const int N = ctx->nranks;
  if ((N <= 1 && !(ctx->comm && getenv("MFX_COMM_SELF_TEST"))) || (!ctx->comm && !ctx->ext_reduce)) return MFX_OK;
  NEED(ctx->item_parts == N || N == 1, MFX_E_STATE, "mfx_allgather_item_parts: mfx_sgd_set_item_parts(%d) must name as many parts as there are ranks (%d)",
       ctx->item_parts, N);
  NEED(my_part >= 0 && my_part < N, MFX_E_ARG, "mfx_allgather_item_parts: part %d of %d", my_part, N);
  HIPCHK(hipSetDevice(ctx->device));
  int64_t slab;
  int rc = ensure_part_buffers(ctx, &slab);
  if (rc) return rc;
  const int ld4 = ctx->ld / 4, grid = part_grid(slab);
  float* mine = ctx->comm_tmp;
  float* big = ctx->comm_tmp + slab;          // slab r of `big`: what rank r contributed, i.e. part (my_part - rank + r) mod N
  hipLaunchKernelGGL(part_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->V, (int64_t)ctx->nI, ld4, my_part, N, mine);
  HIPCHK(hipGetLastError());
  if (ctx->comm) {
    const int r = g_rccl.allgather(mine, big, (size_t)slab, kNcclFloat, ctx->comm, ctx->stream);
    NEED(r == 0, MFX_E_COMM, "ncclAllGather: %s", rccl_err(r));
  } else {
    HIPCHK(hipMemsetAsync(big, 0, sizeof(float) * (size_t)slab * N, ctx->stream));
    HIPCHK(hipMemcpyAsync(big + (size_t)ctx->rank * slab, mine, sizeof(float) * (size_t)slab, hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = mfx_comm_allreduce(ctx, big, (size_t)slab * N, 0))) return rc;
  }
  for (int r = 0; r < N; r++) {
    if (r == ctx->rank) continue;
    const int part = ((my_part - ctx->rank + r) % N + N) % N;
    hipLaunchKernelGGL(part_unpack_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->V, (int64_t)ctx->nI, ld4, part, N,
                       (const float*)(big + (size_t)r * slab));
  }
  HIPCHK(hipGetLastError());
  if (ctx->Vsync) HIPCHK(hipMemcpyAsync(ctx->Vsync, ctx->V, sizeof(float) * (size_t)ctx->nI * ctx->ld, hipMemcpyDeviceToDevice, ctx->stream));
  if (!ctx->comm_checked_parts) {            // once per communicator: prev / next, the receive offsets and the unpacked parts are rank arithmetic
    ctx->comm_checked_parts = true;
    if ((rc = mfx_comm_check_replicas(ctx, ctx->V, (size_t)ctx->nI * ctx->ld, "mfx_allgather_item_parts (item-part rotation)"))) return rc;
  }
  return MFX_OK;
}

extern "C" int mfx_allreduce_f64(mfx_ctx* ctx, double* vals, int n) {
  if (!ctx) return MFX_E_ARG;
  NEED(vals && n > 0 && n <= 8, MFX_E_ARG, "mfx_allreduce_f64: n must be in [1,8]");
  if (!ctx->comm && !ctx->ext_reduce) return MFX_OK;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->red_blocks < 1) {
    int rc;
    dev_free(ctx->red_d); dev_free(ctx->red_i);
    if ((rc = dev_alloc(ctx, &ctx->red_d, (size_t)16))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->red_i, (size_t)8))) return rc;
    ctx->red_blocks = 8;
  }
  HIPCHK(hipMemcpyAsync(ctx->red_d, vals, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  int rc = mfx_comm_allreduce(ctx, ctx->red_d, (size_t)n, 1);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ctx->red_out, ctx->red_d, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < n; k++) vals[k] = ctx->red_out[k];
  return MFX_OK;
}
