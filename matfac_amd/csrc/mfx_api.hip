// mfx_api.hip -- context lifetime, data/model upload, epoch-list construction.
// Replaces the state handling of Data (datastruct.cpp:3-120) and Model
// (model.cpp:2315-2366, :1492, :1500-1504) on the device.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "mfx_internal.h"

static thread_local std::string g_create_err;

int mfx_fail(mfx_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_create_err = buf;
  return code;
}

ProfScope::ProfScope(mfx_ctx* c, int kernel) : ctx(c), k(kernel) {
  if (!ctx->prof_on) return;
  ProfSlot& s = ctx->prof[k];
  if (!s.pool.empty()) { a = s.pool.back().first; b = s.pool.back().second; s.pool.pop_back(); }
  else {
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
  }
  (void)hipEventRecord(a, ctx->stream);
}
ProfScope::~ProfScope() {
  if (!a) return;
  (void)hipEventRecord(b, ctx->stream);
  ctx->prof[k].pending.emplace_back(a, b);
}

extern "C" {

int mfx_version(void) { return 100; }

int mfx_device_count(int* n) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) c = 0;
  if (n) *n = c;
  return MFX_OK;
}

const char* mfx_last_error(const mfx_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int mfx_create(int device, mfx_ctx** out) {
  mfx_ctx* ctx = nullptr;
  if (!out) return mfx_fail(nullptr, MFX_E_ARG, "mfx_create: out is NULL");
  *out = nullptr;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c <= 0)
    return mfx_fail(nullptr, MFX_E_NODEVICE,
                    "mfx_create: no HIP device (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= c)
    return mfx_fail(nullptr, MFX_E_ARG, "mfx_create: device %d out of range [0,%d)", device, c);
  e = hipSetDevice(device);
  if (e != hipSuccess) return mfx_fail(nullptr, MFX_E_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  ctx = new mfx_ctx;
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete ctx;
    return mfx_fail(nullptr, MFX_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  e = hipHostMalloc((void**)&ctx->red_out, 8 * sizeof(double));
  if (e != hipSuccess) {
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return mfx_fail(nullptr, MFX_E_OOM, "hipHostMalloc: %s", hipGetErrorString(e));
  }
  *out = ctx;
  return MFX_OK;
}

static void free_csr(DevCSR& m) {
  dev_free(m.rowptr); dev_free(m.rowind); dev_free(m.rowval); dev_free(m.rowid);
  dev_free(m.colptr); dev_free(m.colind); dev_free(m.colval);
  m = DevCSR();
}

static void free_model(mfx_ctx* ctx) {
  mfx_bias_free_internal(ctx);
  dev_free(ctx->U); dev_free(ctx->V); dev_free(ctx->Ubest); dev_free(ctx->Vbest);
  dev_free(ctx->Vsync); dev_free(ctx->invU); dev_free(ctx->invI);
  ctx->have_invalid = false;
  ctx->n_invalid = -1;
}

void mfx_destroy(mfx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  mfx_comm_free_internal(ctx);
  dev_free(ctx->dimreg);
  mfx_ifw_free_internal(ctx);
  mfx_tmf_free_internal(ctx);
  mfx_ccd_free_internal(ctx);
  mfx_cd_free_internal(ctx);
  mfx_als_free_internal(ctx);
  mfx_segs_free_internal(ctx);
  for (auto& m : ctx->mat) free_csr(m);
  free_model(ctx);
  dev_free(ctx->eu); dev_free(ctx->ei); dev_free(ctx->er); dev_free(ctx->order); dev_free(ctx->order32); dev_free(ctx->sw_buf); dev_free(ctx->sw_tmp);
  mfx_slots_free_internal(ctx);
  mfx_levels_free_internal(ctx);
  mfx_flow_free_internal(ctx);
  dev_free(ctx->ulist); dev_free(ctx->red_d); dev_free(ctx->red_i);
  if (ctx->red_out) (void)hipHostFree(ctx->red_out);
  for (auto& s : ctx->prof) {
    for (auto& p : s.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto& p : s.pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int mfx_synchronize(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return mfx_slots_check_abort(ctx);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// CSR upload
// ---------------------------------------------------------------------------
__global__ void expand_rowid_kernel(const int64_t* __restrict__ rowptr, int32_t nrows,
                                    int32_t* __restrict__ rowid) {
  // one wave per row, grid-stride; rows are short on average and this runs once
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t u = wave; u < nrows; u += nwaves) {
    const int64_t b = rowptr[u], e = rowptr[u + 1];
    for (int64_t t = b + lane; t < e; t += 64) rowid[t] = (int32_t)u;
  }
}

extern "C" int mfx_set_csr(mfx_ctx* ctx, int which, int32_t nrows, int32_t ncols,
                           const int64_t* rowptr, const int32_t* rowind, const float* rowval,
                           const int64_t* colptr, const int32_t* colind, const float* colval) {
  if (!ctx) return MFX_E_ARG;
  NEED(which >= 0 && which < 3, MFX_E_ARG, "mfx_set_csr: which=%d", which);
  NEED(nrows >= 0 && ncols >= 0 && rowptr, MFX_E_ARG, "mfx_set_csr: bad shape/pointers");
  const int64_t nnz = rowptr[nrows];
  NEED(rowptr[0] == 0 && nnz >= 0, MFX_E_ARG, "mfx_set_csr: rowptr[0] must be 0");
  NEED(nnz == 0 || (rowind && rowval), MFX_E_ARG, "mfx_set_csr: rowind/rowval NULL");
  // the kernels index factor rows with these: validate once on the host
  for (int32_t u = 0; u < nrows; u++)
    NEED(rowptr[u + 1] >= rowptr[u], MFX_E_ARG, "mfx_set_csr: rowptr not monotone at row %d", u);
  for (int64_t e = 0; e < nnz; e++)
    NEED(rowind[e] >= 0 && rowind[e] < ncols, MFX_E_ARG,
         "mfx_set_csr: column index %d out of [0,%d) at entry %lld", rowind[e], ncols, (long long)e);
  HIPCHK(hipSetDevice(ctx->device));
  DevCSR& m = ctx->mat[which];
  free_csr(m);
  if (which == MFX_MAT_TRAIN) { ctx->train_gen++; mfx_comm_drop_col_counts(ctx); mfx_ccd_free_internal(ctx); mfx_cd_free_internal(ctx); mfx_als_free_internal(ctx); mfx_segs_free_internal(ctx); mfx_slots_free_internal(ctx); }
  int rc;
  if ((rc = dev_alloc(ctx, &m.rowptr, (size_t)nrows + 1))) return rc;
  if ((rc = dev_alloc(ctx, &m.rowind, (size_t)nnz))) return rc;
  if ((rc = dev_alloc(ctx, &m.rowval, (size_t)nnz))) return rc;
  if ((rc = dev_alloc(ctx, &m.rowid, (size_t)nnz))) return rc;
  HIPCHK(hipMemcpyAsync(m.rowptr, rowptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice, ctx->stream));
  if (nnz) {
    HIPCHK(hipMemcpyAsync(m.rowind, rowind, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(m.rowval, rowval, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
  }
  m.nrows = nrows; m.ncols = ncols; m.nnz = nnz;
  if (nrows > 0) {
    int blocks = (int)std::min<int64_t>(((int64_t)nrows + 3) / 4, 4096);
    hipLaunchKernelGGL(expand_rowid_kernel, dim3(blocks), dim3(256), 0, ctx->stream, m.rowptr, nrows, m.rowid);
    HIPCHK(hipGetLastError());
  }
  // column view: given, or built on the device (stable sort by column == gk_csr_CreateIndex(COL))
  if (!colptr && which == MFX_MAT_TRAIN) {
    if ((rc = mfx_build_col_index_device(ctx, m))) return rc;
  }
  if (colptr) {
    NEED(colptr[0] == 0 && colptr[ncols] == nnz, MFX_E_ARG, "mfx_set_csr: colptr inconsistent with nnz");
    for (int64_t e = 0; e < nnz; e++)
      NEED(colind[e] >= 0 && colind[e] < nrows, MFX_E_ARG, "mfx_set_csr: row index out of range in column view");
    if ((rc = dev_alloc(ctx, &m.colptr, (size_t)ncols + 1))) return rc;
    if ((rc = dev_alloc(ctx, &m.colind, (size_t)nnz))) return rc;
    if ((rc = dev_alloc(ctx, &m.colval, (size_t)nnz))) return rc;
    HIPCHK(hipMemcpyAsync(m.colptr, colptr, sizeof(int64_t) * ((size_t)ncols + 1), hipMemcpyHostToDevice, ctx->stream));
    if (nnz) {
      HIPCHK(hipMemcpyAsync(m.colind, colind, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(m.colval, colval, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    }
    m.has_col = true;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));  // host buffers are borrowed for this call only
  m.present = true;
  if (which == MFX_MAT_TRAIN) { ctx->have_invalid = false; ctx->n_invalid = -1; }
  return MFX_OK;
}

extern "C" int mfx_debug_col_view(mfx_ctx* ctx, int64_t* colptr, int32_t* colind, float* colval) {
  if (!ctx) return MFX_E_ARG;
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present && m.has_col, MFX_E_STATE, "mfx_debug_col_view: no train matrix with a column view");
  NEED(colptr && colind && colval, MFX_E_ARG, "mfx_debug_col_view: NULL output");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(colptr, m.colptr, sizeof(int64_t) * ((size_t)m.ncols + 1), hipMemcpyDeviceToHost));
  if (m.nnz) {
    HIPCHK(hipMemcpy(colind, m.colind, sizeof(int32_t) * (size_t)m.nnz, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(colval, m.colval, sizeof(float) * (size_t)m.nnz, hipMemcpyDeviceToHost));
  }
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
extern "C" int mfx_set_model(mfx_ctx* ctx, int32_t nUsers, int32_t nItems, int32_t K) {
  if (!ctx) return MFX_E_ARG;
  NEED(nUsers > 0 && nItems > 0 && K > 0 && K <= 512, MFX_E_ARG,
       "mfx_set_model: need nUsers>0, nItems>0, 0<K<=512 (got %d,%d,%d)", nUsers, nItems, K);
  HIPCHK(hipSetDevice(ctx->device));
  free_model(ctx);
  mfx_ccd_free_internal(ctx);
  mfx_cd_free_internal(ctx);
  mfx_als_wide_free_internal(ctx);
  dev_free(ctx->dimreg);
  mfx_ifw_free_internal(ctx);
  mfx_tmf_free_internal(ctx);
  ctx->nU = nUsers; ctx->nI = nItems; ctx->K = K;
  mfx_tree_shape(K, &ctx->L, &ctx->C);
  ctx->ld = 4 * ctx->L * ctx->C;
  // one extra, always-zero row behind each factor table: the ALS accumulation gathers it for ratings it must skip
  // (<= 0, or past the end of a row) instead of masking every step
  const size_t su = ((size_t)nUsers + 1) * ctx->ld, si = ((size_t)nItems + 1) * ctx->ld;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->U, su))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->V, si))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->Ubest, su))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->Vbest, si))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->invU, (size_t)nUsers))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->invI, (size_t)nItems))) return rc;
  HIPCHK(hipMemsetAsync(ctx->U, 0, su * sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->V, 0, si * sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->Ubest, 0, su * sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->Vbest, 0, si * sizeof(float), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->invU, 0, (size_t)nUsers, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->invI, 0, (size_t)nItems, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MFX_OK;
}

// host [n][K] (row- or column-major) <-> device [n][ld] zero padded
static void pack_rows(const float* src, float* dst, int64_t n, int K, int ld, int layout) {
  for (int64_t r = 0; r < n; r++) {
    float* d = dst + r * ld;
    if (layout == MFX_ROWMAJOR) memcpy(d, src + r * K, sizeof(float) * K);
    else for (int k = 0; k < K; k++) d[k] = src[(int64_t)k * n + r];
    for (int k = K; k < ld; k++) d[k] = 0.0f;
  }
}
static void unpack_rows(const float* src, float* dst, int64_t n, int K, int ld, int layout) {
  for (int64_t r = 0; r < n; r++) {
    const float* s = src + r * ld;
    if (layout == MFX_ROWMAJOR) memcpy(dst + r * K, s, sizeof(float) * K);
    else for (int k = 0; k < K; k++) dst[(int64_t)k * n + r] = s[k];
  }
}

static int upload_mat(mfx_ctx* ctx, const float* host, float* dev, int64_t n, int layout) {
  const int K = ctx->K, ld = ctx->ld;
  const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / (ld * 4));
  std::vector<float> stage((size_t)std::min(chunk, n) * ld);
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t m = std::min(chunk, n - r0);
    if (layout == MFX_ROWMAJOR) pack_rows(host + r0 * K, stage.data(), m, K, ld, layout);
    else
      for (int64_t r = 0; r < m; r++) {
        float* d = stage.data() + r * ld;
        for (int k = 0; k < K; k++) d[k] = host[(int64_t)k * n + r0 + r];
        for (int k = K; k < ld; k++) d[k] = 0.0f;
      }
    HIPCHK(hipMemcpy(dev + r0 * ld, stage.data(), sizeof(float) * (size_t)m * ld, hipMemcpyHostToDevice));
  }
  return MFX_OK;
}
static int download_mat(mfx_ctx* ctx, const float* dev, float* host, int64_t n, int layout) {
  const int K = ctx->K, ld = ctx->ld;
  const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / (ld * 4));
  std::vector<float> stage((size_t)std::min(chunk, n) * ld);
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t m = std::min(chunk, n - r0);
    HIPCHK(hipMemcpy(stage.data(), dev + r0 * ld, sizeof(float) * (size_t)m * ld, hipMemcpyDeviceToHost));
    if (layout == MFX_ROWMAJOR) unpack_rows(stage.data(), host + r0 * K, m, K, ld, layout);
    else
      for (int64_t r = 0; r < m; r++)
        for (int k = 0; k < K; k++) host[(int64_t)k * n + r0 + r] = stage[(size_t)r * ld + k];
  }
  return MFX_OK;
}

extern "C" int mfx_set_factors(mfx_ctx* ctx, const float* U, const float* V, int layout) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_set_factors: call mfx_set_model first");
  NEED(layout == MFX_ROWMAJOR || layout == MFX_COLMAJOR, MFX_E_ARG, "mfx_set_factors: layout");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc;
  if (U && (rc = upload_mat(ctx, U, ctx->U, ctx->nU, layout))) return rc;
  if (V && (rc = upload_mat(ctx, V, ctx->V, ctx->nI, layout))) return rc;
  return MFX_OK;
}

extern "C" int mfx_get_factors(mfx_ctx* ctx, int snapshot, float* U, float* V, int layout) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_get_factors: call mfx_set_model first");
  NEED(layout == MFX_ROWMAJOR || layout == MFX_COLMAJOR, MFX_E_ARG, "mfx_get_factors: layout");
  NEED(snapshot == MFX_SNAP_CURRENT || snapshot == MFX_SNAP_BEST, MFX_E_ARG, "mfx_get_factors: snapshot");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc;
  if ((rc = mfx_slots_check_abort(ctx))) return rc;      // factors of an epoch whose drain gave up are not handed out silently
  if (U && (rc = download_mat(ctx, snapshot ? ctx->Ubest : ctx->U, U, ctx->nU, layout))) return rc;
  if (V && (rc = download_mat(ctx, snapshot ? ctx->Vbest : ctx->V, V, ctx->nI, layout))) return rc;
  return MFX_OK;
}

__global__ void invalid_kernel(const int64_t* __restrict__ ptr, int32_t nmat, int32_t n,
                               uint8_t* __restrict__ inv) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) inv[t] = (t >= nmat) || (ptr[t + 1] == ptr[t]);
}

__global__ void invalid_items_global_kernel(const double* __restrict__ gcol, int32_t ncols, int32_t n, uint8_t* __restrict__ inv) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) inv[t] = (t >= ncols) || (gcol[t] == 0.0);
}

// number of set bytes, added to *out (the evaluation skips its mask loads when nothing is invalid)
__global__ void mask_count_kernel(const uint8_t* __restrict__ inv, int32_t n, unsigned long long* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long b = __ballot(t < n && inv[t] != 0);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long)__popcll(b));
}

extern "C" int mfx_compute_invalid(mfx_ctx* ctx, uint8_t* invalidUsers, uint8_t* invalidItems) {
  if (!ctx) return MFX_E_ARG;
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present && m.has_col, MFX_E_STATE, "mfx_compute_invalid: train matrix with column view needed");
  NEED(ctx->U, MFX_E_STATE, "mfx_compute_invalid: call mfx_set_model first");
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(invalid_kernel, dim3((ctx->nU + 255) / 256), dim3(256), 0, ctx->stream,
                     m.rowptr, m.nrows, ctx->nU, ctx->invU);
  if (mfx_sharded(ctx)) {   // an item is invalid when NO rank has a rating for it
    const double* gcol;
    int rc = mfx_comm_global_col_counts(ctx, &gcol);
    if (rc) return rc;
    hipLaunchKernelGGL(invalid_items_global_kernel, dim3((ctx->nI + 255) / 256), dim3(256), 0, ctx->stream, gcol, m.ncols,
                       ctx->nI, ctx->invI);
  } else {
    hipLaunchKernelGGL(invalid_kernel, dim3((ctx->nI + 255) / 256), dim3(256), 0, ctx->stream,
                       m.colptr, m.ncols, ctx->nI, ctx->invI);
  }
  HIPCHK(hipGetLastError());
  ctx->n_invalid = -1;
  unsigned long long* dcount = nullptr;
  int rcnt = dev_alloc(ctx, &dcount, (size_t)1);
  if (rcnt) return rcnt;
  HIPCHK(hipMemsetAsync(dcount, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(mask_count_kernel, dim3((ctx->nU + 255) / 256), dim3(256), 0, ctx->stream, ctx->invU, ctx->nU, dcount);
  hipLaunchKernelGGL(mask_count_kernel, dim3((ctx->nI + 255) / 256), dim3(256), 0, ctx->stream, ctx->invI, ctx->nI, dcount);
  unsigned long long hcount = 0;
  HIPCHK(hipMemcpyAsync(&hcount, dcount, sizeof(hcount), hipMemcpyDeviceToHost, ctx->stream));
  if (invalidUsers) HIPCHK(hipMemcpyAsync(invalidUsers, ctx->invU, (size_t)ctx->nU, hipMemcpyDeviceToHost, ctx->stream));
  if (invalidItems) HIPCHK(hipMemcpyAsync(invalidItems, ctx->invI, (size_t)ctx->nI, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  dev_free(dcount);
  ctx->n_invalid = (int64_t)hcount;
  ctx->have_invalid = true;
  return MFX_OK;
}

extern "C" int mfx_snapshot_best(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_snapshot_best: no model");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(ctx->Ubest, ctx->U, sizeof(float) * (size_t)ctx->nU * ctx->ld, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->Vbest, ctx->V, sizeof(float) * (size_t)ctx->nI * ctx->ld, hipMemcpyDeviceToDevice, ctx->stream));
  if (ctx->ub) {
    HIPCHK(hipMemcpyAsync(ctx->ub_best, ctx->ub, sizeof(float) * (size_t)ctx->nU, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->ib_best, ctx->ib, sizeof(float) * (size_t)ctx->nI, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return MFX_OK;
}
extern "C" int mfx_restore_best(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  NEED(ctx->U, MFX_E_STATE, "mfx_restore_best: no model");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(ctx->U, ctx->Ubest, sizeof(float) * (size_t)ctx->nU * ctx->ld, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->V, ctx->Vbest, sizeof(float) * (size_t)ctx->nI * ctx->ld, hipMemcpyDeviceToDevice, ctx->stream));
  if (ctx->ub) {
    HIPCHK(hipMemcpyAsync(ctx->ub, ctx->ub_best, sizeof(float) * (size_t)ctx->nU, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->ib, ctx->ib_best, sizeof(float) * (size_t)ctx->nI, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// epoch list: replaces std::shuffle(uiRatingInds) (modelMF.cpp:76-81, 1739-1744)
// ---------------------------------------------------------------------------
// Bijection on [0,n): alternating (unbalanced) Feistel network on ceil(log2 n) bits
// keyed by (seed, epoch), cycle-walked into range.  Documented in DESIGN.md; the
// python tests re-implement it to check the device list is that permutation.
template <int MODE>  // 0: feistel, 1: order[] array
__global__ void build_epoch_list_kernel(const int32_t* __restrict__ cu, const int32_t* __restrict__ ci,
                                        const float* __restrict__ cr, int64_t nsrc,
                                        const uint64_t* __restrict__ order, int64_t n, int abits, int bbits,
                                        uint32_t k0, uint32_t k1, int32_t* __restrict__ eu,
                                        int32_t* __restrict__ ei, float* __restrict__ er) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    int64_t s;
    if (MODE == 0) s = mfx_perm_index(t, n, abits, bbits, k0, k1);
    else { s = (int64_t)order[t]; if (s < 0 || s >= nsrc) s = 0; }
    eu[t] = cu[s]; ei[t] = ci[s]; er[t] = cr[s];
  }
}

static int ensure_elist(mfx_ctx* ctx, int64_t n) {
  if (ctx->elist_cap >= n && ctx->eu) return MFX_OK;
  dev_free(ctx->eu); dev_free(ctx->ei); dev_free(ctx->er);
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->eu, (size_t)n))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->ei, (size_t)n))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->er, (size_t)n))) return rc;
  ctx->elist_cap = n;
  return MFX_OK;
}

extern "C" int mfx_sgd_set_order(mfx_ctx* ctx, const uint64_t* perm, int64_t n) {
  if (!ctx) return MFX_E_ARG;
  NEED(perm && n >= 0, MFX_E_ARG, "mfx_sgd_set_order: perm NULL or n<0");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->order_cap < n || !ctx->order) {
    dev_free(ctx->order);
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->order, (size_t)n))) return rc;
    ctx->order_cap = n;
  }
  if (n) HIPCHK(hipMemcpyAsync(ctx->order, perm, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->order32_valid = false;
  ctx->order_n = n;
  return MFX_OK;
}

__global__ void widen_order_kernel(const uint32_t* __restrict__ in, int64_t n, uint64_t* __restrict__ out) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = in[t];
}
extern "C" int mfx_sgd_set_order32(mfx_ctx* ctx, const uint32_t* perm, int64_t n) {
  if (!ctx) return MFX_E_ARG;
  NEED(perm && n >= 0, MFX_E_ARG, "mfx_sgd_set_order32: perm NULL or n<0");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  if (ctx->order_cap < n || !ctx->order) {
    dev_free(ctx->order);
    ctx->order_cap = 0;
    if ((rc = dev_alloc(ctx, &ctx->order, (size_t)n))) return rc;
    ctx->order_cap = n;
  }
  if (ctx->order32_cap < n || !ctx->order32) {
    dev_free(ctx->order32);
    ctx->order32_cap = 0;
    if ((rc = dev_alloc(ctx, &ctx->order32, (size_t)n))) return rc;
    ctx->order32_cap = n;
  }
  if (n) {
    HIPCHK(hipMemcpyAsync(ctx->order32, perm, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(widen_order_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)ctx->order32, n, ctx->order);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));        // (the caller's buffer is free when this returns)
  ctx->order_n = n;
  ctx->order32_valid = true;
  return MFX_OK;
}

__global__ void build_user_list_kernel(const uint64_t* __restrict__ order, int64_t n, int32_t nU,
                                       int32_t* __restrict__ ulist) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) { uint64_t u = order ? order[t] : (uint64_t)t; ulist[t] = u < (uint64_t)nU ? (int32_t)u : 0; }
}

extern "C" int mfx_sgd_epoch(mfx_ctx* ctx, const mfx_sgd_opts* o) {
  if (!ctx) return MFX_E_ARG;
  NEED(o, MFX_E_ARG, "mfx_sgd_epoch: opts NULL");
  const DevCSR& m = ctx->mat[MFX_MAT_TRAIN];
  NEED(m.present, MFX_E_STATE, "mfx_sgd_epoch: no train matrix");
  NEED(ctx->U, MFX_E_STATE, "mfx_sgd_epoch: no model");
  NEED(m.nrows <= ctx->nU && m.ncols <= ctx->nI, MFX_E_ARG,
       "mfx_sgd_epoch: train matrix %dx%d exceeds model %dx%d", m.nrows, m.ncols, ctx->nU, ctx->nI);
  NEED(o->mode >= MFX_SGD_HOGWILD && o->mode <= MFX_SGD_LEVELS, MFX_E_ARG, "mfx_sgd_epoch: mode=%d", o->mode);
  NEED(!ctx->dimreg || o->mode == MFX_SGD_HOGWILD || o->mode == MFX_SGD_SERIAL || o->mode == MFX_SGD_TILED || o->mode == MFX_SGD_LEVELS, MFX_E_ARG,
       "mfx_sgd_epoch: per-dimension regularisation (mfx_sgd_set_dim_reg) runs on MFX_SGD_HOGWILD / MFX_SGD_SERIAL / MFX_SGD_TILED");
  NEED(!ctx->ifw || o->mode == MFX_SGD_HOGWILD || o->mode == MFX_SGD_SERIAL || o->mode == MFX_SGD_TILED || o->mode == MFX_SGD_LEVELS, MFX_E_ARG,
       "mfx_sgd_epoch: rating weights (mfx_sgd_set_ifw) run on MFX_SGD_HOGWILD / MFX_SGD_SERIAL / MFX_SGD_TILED");
  NEED(!(ctx->ifw && ctx->dimreg), MFX_E_STATE, "mfx_sgd_epoch: rating weights and per-dimension regularisation are exclusive");
  NEED(!ctx->tmf_u || o->mode == MFX_SGD_HOGWILD || o->mode == MFX_SGD_SERIAL || o->mode == MFX_SGD_TILED || o->mode == MFX_SGD_LEVELS, MFX_E_ARG,
       "mfx_sgd_epoch: truncated ranks (mfx_set_tmf) run on MFX_SGD_HOGWILD / MFX_SGD_SERIAL / MFX_SGD_TILED");
  NEED(!(ctx->tmf_u && (ctx->ifw || ctx->dimreg)), MFX_E_STATE, "mfx_sgd_epoch: truncated ranks exclude the other SGD variants");
  NEED(o->order >= MFX_ORDER_DEVICE && o->order <= MFX_ORDER_NATURAL, MFX_E_ARG, "mfx_sgd_epoch: order=%d", o->order);
  NEED(o->arith >= MFX_ARITH_REF64 && o->arith <= MFX_ARITH_F32, MFX_E_ARG, "mfx_sgd_epoch: arith=%d", o->arith);
  HIPCHK(hipSetDevice(ctx->device));
  // sampled timing (mfx_prof_enable(ctx, N > 1)): events around the launches of every N-th epoch only -- an event pair per
  // launch costs 6 % of a 1 ms epoch (scripts/event_overhead.py)
  struct ProfGate {
    mfx_ctx* c; bool saved;
    ProfGate(mfx_ctx* c_) : c(c_), saved(c_->prof_on) { if (saved && c->prof_period > 1) c->prof_on = (c->prof_tick++ % c->prof_period) == 0; }
    ~ProfGate() { c->prof_on = saved; }
  } gate(ctx);

  if (o->mode == MFX_SGD_USERS) {
    // user list: host order (shuffled valid users) or 0..nrows-1
    int64_t nu = o->order == MFX_ORDER_HOST ? ctx->order_n : m.nrows;
    NEED(o->order != MFX_ORDER_DEVICE, MFX_E_ARG, "mfx_sgd_epoch: MFX_SGD_USERS needs HOST or NATURAL order");
    if (nu == 0) return MFX_OK;
    if (ctx->ulist_cap < nu || !ctx->ulist) {
      dev_free(ctx->ulist);
      int rc;
      if ((rc = dev_alloc(ctx, &ctx->ulist, (size_t)nu))) return rc;
      ctx->ulist_cap = nu;
    }
    hipLaunchKernelGGL(build_user_list_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, ctx->stream,
                       o->order == MFX_ORDER_HOST ? ctx->order : nullptr, nu, m.nrows, ctx->ulist);
    HIPCHK(hipGetLastError());
    return mfx_launch_sgd_users(ctx, o, nu);
  }

  if (o->mode == MFX_SGD_TILED) {
    NEED(o->order == MFX_ORDER_DEVICE, MFX_E_ARG, "mfx_sgd_epoch: MFX_SGD_TILED needs MFX_ORDER_DEVICE");
    NEED(o->count <= 0, MFX_E_ARG, "mfx_sgd_epoch: MFX_SGD_TILED visits the whole list");
    if (m.nnz == 0) return MFX_OK;
    return mfx_launch_sgd_tiled(ctx, o);
  }

  int64_t n = m.nnz;
  if (o->order == MFX_ORDER_HOST) {
    NEED(ctx->order && ctx->order_n > 0, MFX_E_STATE, "mfx_sgd_epoch: MFX_ORDER_HOST without mfx_sgd_set_order");
    n = ctx->order_n;
  }
  if (n == 0) return MFX_OK;
  int rc;
  if ((rc = ensure_elist(ctx, n))) return rc;
  {
    ProfScope ps(ctx, MFX_K_PERMUTE);
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 8192);
    if (o->order == MFX_ORDER_DEVICE) {
      int bits = 1;
      while (((int64_t)1 << bits) < n) bits++;
      if (bits < 2) bits = 2;
      const int abits = bits / 2, bbits = bits - abits;
      const uint32_t k0 = mfx_mix32(o->seed ^ 0x3c6ef372U) + (uint32_t)o->epoch * 0x9e3779b9U;
      const uint32_t k1 = mfx_mix32(o->seed * 0x85ebca6bU + 0xdaa66d2bU) ^ mfx_mix32((uint32_t)o->epoch + 0x1b873593U);
      hipLaunchKernelGGL(build_epoch_list_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, m.rowid, m.rowind,
                         m.rowval, m.nnz, (const uint64_t*)nullptr, n, abits, bbits, k0, k1, ctx->eu, ctx->ei, ctx->er);
    } else if (o->order == MFX_ORDER_HOST) {
      hipLaunchKernelGGL(build_epoch_list_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, m.rowid, m.rowind,
                         m.rowval, m.nnz, ctx->order, n, 0, 0, 0u, 0u, ctx->eu, ctx->ei, ctx->er);
    } else {
      HIPCHK(hipMemcpyAsync(ctx->eu, m.rowid, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->ei, m.rowind, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->er, m.rowval, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(hipGetLastError());
  }
  ctx->elist_n = n;
  int64_t first = o->count > 0 ? o->first : 0;
  int64_t count = o->count > 0 ? o->count : n;
  NEED(first >= 0 && first + count <= n, MFX_E_ARG, "mfx_sgd_epoch: sub-range [%lld,+%lld) outside list of %lld",
       (long long)first, (long long)count, (long long)n);
  return mfx_launch_sgd(ctx, o, first, count);
}

extern "C" int mfx_debug_epoch_list(mfx_ctx* ctx, int32_t* u, int32_t* i, float* r, int64_t cap, int64_t* n) {
  if (!ctx) return MFX_E_ARG;
  NEED(n, MFX_E_ARG, "mfx_debug_epoch_list: n NULL");
  if (ctx->elist_n < 0) {   // last epoch was MFX_SGD_TILED: rebuild its visiting order
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_elist(ctx, ctx->mat[MFX_MAT_TRAIN].nnz);
    if (rc) return rc;
    if ((rc = mfx_slots_materialise_order(ctx))) return rc;
  }
  *n = ctx->elist_n;
  if (!u && !i && !r) return MFX_OK;
  NEED(cap >= ctx->elist_n, MFX_E_ARG, "mfx_debug_epoch_list: cap too small");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t k = (size_t)ctx->elist_n;
  if (k == 0) return MFX_OK;
  if (u) HIPCHK(hipMemcpy(u, ctx->eu, sizeof(int32_t) * k, hipMemcpyDeviceToHost));
  if (i) HIPCHK(hipMemcpy(i, ctx->ei, sizeof(int32_t) * k, hipMemcpyDeviceToHost));
  if (r) HIPCHK(hipMemcpy(r, ctx->er, sizeof(float) * k, hipMemcpyDeviceToHost));
  return MFX_OK;
}

extern "C" int mfx_eval(mfx_ctx* ctx, int which, int snapshot, int with_norms, mfx_eval_out* out) {
  if (!ctx) return MFX_E_ARG;
  NEED(out, MFX_E_ARG, "mfx_eval: out NULL");
  NEED(which >= 0 && which < 3, MFX_E_ARG, "mfx_eval: which=%d", which);
  NEED(ctx->mat[which].present, MFX_E_STATE, "mfx_eval: matrix %d not set", which);
  NEED(ctx->U, MFX_E_STATE, "mfx_eval: no model");
  NEED(ctx->have_invalid, MFX_E_STATE, "mfx_eval: call mfx_compute_invalid first");
  NEED(snapshot == MFX_SNAP_CURRENT || snapshot == MFX_SNAP_BEST, MFX_E_ARG, "mfx_eval: snapshot");
  HIPCHK(hipSetDevice(ctx->device));
  return mfx_launch_eval(ctx, ctx->mat[which], snapshot ? ctx->Ubest : ctx->U, snapshot ? ctx->Vbest : ctx->V,
                         with_norms, out);
}

extern "C" int mfx_eval2(mfx_ctx* ctx, int whichA, int normsA, int whichB, int normsB, int snapshot, mfx_eval_out* outA,
                         mfx_eval_out* outB) {
  if (!ctx) return MFX_E_ARG;
  NEED(outA && outB, MFX_E_ARG, "mfx_eval2: out NULL");
  NEED(whichA >= 0 && whichA < 3 && whichB >= 0 && whichB < 3, MFX_E_ARG, "mfx_eval2: which=%d,%d", whichA, whichB);
  NEED(ctx->mat[whichA].present && ctx->mat[whichB].present, MFX_E_STATE, "mfx_eval2: matrix not set");
  NEED(ctx->U, MFX_E_STATE, "mfx_eval2: no model");
  NEED(ctx->have_invalid, MFX_E_STATE, "mfx_eval2: call mfx_compute_invalid first");
  NEED(snapshot == MFX_SNAP_CURRENT || snapshot == MFX_SNAP_BEST, MFX_E_ARG, "mfx_eval2: snapshot");
  HIPCHK(hipSetDevice(ctx->device));
  return mfx_launch_eval2(ctx, ctx->mat[whichA], normsA, ctx->mat[whichB], normsB, snapshot ? ctx->Ubest : ctx->U,
                          snapshot ? ctx->Vbest : ctx->V, outA, outB);
}

__global__ void mask_combine_kernel(const uint8_t* __restrict__ inv, const uint8_t* __restrict__ keep, int32_t n, uint8_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = inv[t] | (keep[t] ? 0 : 1);
}

extern "C" int mfx_eval_filtered(mfx_ctx* ctx, int which, int snapshot, const uint8_t* keepUsers, const uint8_t* keepItems,
                                 mfx_eval_out* out) {
  if (!ctx) return MFX_E_ARG;
  NEED(out, MFX_E_ARG, "mfx_eval_filtered: out NULL");
  NEED(which >= 0 && which < 3, MFX_E_ARG, "mfx_eval_filtered: which=%d", which);
  NEED(ctx->mat[which].present, MFX_E_STATE, "mfx_eval_filtered: matrix %d not set", which);
  NEED(ctx->U, MFX_E_STATE, "mfx_eval_filtered: no model");
  NEED(ctx->have_invalid, MFX_E_STATE, "mfx_eval_filtered: call mfx_compute_invalid first");
  NEED(snapshot == MFX_SNAP_CURRENT || snapshot == MFX_SNAP_BEST, MFX_E_ARG, "mfx_eval_filtered: snapshot");
  HIPCHK(hipSetDevice(ctx->device));
  // the evaluation kernels mask with invU / invI: run them with (invalid OR not kept) in their place
  uint8_t *mu = nullptr, *mi = nullptr, *stage = nullptr;
  uint8_t *saveU = ctx->invU, *saveI = ctx->invI;
  int rc = MFX_OK;
  auto combine = [&](const uint8_t* keep, const uint8_t* inv, int32_t n, uint8_t** out_mask) -> int {
    int r;
    if ((r = dev_alloc(ctx, out_mask, (size_t)n))) return r;
    if ((r = dev_alloc(ctx, &stage, (size_t)n))) return r;
    HIPCHK(hipMemcpyAsync(stage, keep, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mask_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, inv, stage, n, *out_mask);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(stage);
    return MFX_OK;
  };
  if (keepUsers) rc = combine(keepUsers, ctx->invU, ctx->nU, &mu);
  if (!rc && keepItems) rc = combine(keepItems, ctx->invI, ctx->nI, &mi);
  if (!rc) {
    if (mu) ctx->invU = mu;
    if (mi) ctx->invI = mi;
    ctx->force_masks = true;
    rc = mfx_launch_eval(ctx, ctx->mat[which], snapshot ? ctx->Ubest : ctx->U, snapshot ? ctx->Vbest : ctx->V, 0, out);
    ctx->force_masks = false;
    ctx->invU = saveU;
    ctx->invI = saveI;
  }
  dev_free(stage); dev_free(mu); dev_free(mi);
  return rc;
}

// ---------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------
extern "C" int mfx_prof_enable(mfx_ctx* ctx, int on) {
  if (!ctx) return MFX_E_ARG;
  ctx->prof_on = on != 0;
  ctx->prof_period = on > 1 ? on : 1;
  ctx->prof_tick = 0;
  return MFX_OK;
}
static int prof_resolve(mfx_ctx* ctx) {
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (auto& s : ctx->prof) {
    for (auto& p : s.pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { s.ms += ms; s.launches++; }
      s.pool.push_back(p);
    }
    s.pending.clear();
  }
  return MFX_OK;
}
extern "C" int mfx_prof_reset(mfx_ctx* ctx) {
  if (!ctx) return MFX_E_ARG;
  int rc = prof_resolve(ctx);
  if (rc) return rc;
  for (auto& s : ctx->prof) { s.ms = 0; s.launches = 0; }
  return MFX_OK;
}
extern "C" int mfx_prof_get(mfx_ctx* ctx, int kernel, double* total_ms, int64_t* launches) {
  if (!ctx) return MFX_E_ARG;
  NEED(kernel >= 0 && kernel < MFX_K_COUNT, MFX_E_ARG, "mfx_prof_get: kernel=%d", kernel);
  int rc = prof_resolve(ctx);
  if (rc) return rc;
  if (total_ms) *total_ms = ctx->prof[kernel].ms;
  if (launches) *launches = ctx->prof[kernel].launches;
  return MFX_OK;
}
