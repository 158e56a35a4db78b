// csr.h -- the subset of GKlib's gk_csr_t that the MF path touches, owned by this library.
//
// The reference uses gk_csr_t (GKlib, not vendored) and only these members:
// nrows, ncols, rowptr/rowind/rowval, colptr/colind/colval (grep over modelMF.cpp, model.cpp,
// util.cpp, datastruct.cpp).  Field names and types are kept so that code written against
// gk_csr_t reads the same (rowptr is GKlib's ssize_t).
#ifndef MFHOST_CSR_H_
#define MFHOST_CSR_H_
#include <cstdint>
#include <string>

struct csr_t {
  int32_t nrows = 0, ncols = 0;
  int64_t* rowptr = nullptr;
  int32_t* rowind = nullptr;
  float* rowval = nullptr;
  int64_t* colptr = nullptr;
  int32_t* colind = nullptr;
  float* colval = nullptr;
  int64_t nnz() const { return rowptr ? rowptr[nrows] : 0; }
};

// gk_csr_Read(file, GK_CSR_FMT_CSR, readvals=1, numbering=0): one line per row,
// "col val col val ...", empty line = empty row, '%' lines skipped, ncols = max col + 1.
// Returns nullptr (and fills err) when the file cannot be read or is malformed.
csr_t* csr_read_text(const char* path, std::string* err);
int csr_write_text(const csr_t* m, const char* path);
// gk_csr_CreateIndex(mat, GK_CSR_COL): stable counting sort, rows ascending inside a column
void csr_create_col_index(csr_t* m);
csr_t* csr_dup(const csr_t* m);                     // gk_csr_Dup
void csr_free(csr_t** m);                           // gk_csr_Free
csr_t* csr_from_arrays(int32_t nrows, int32_t ncols, const int64_t* rowptr, const int32_t* rowind,
                       const float* rowval);
#endif
