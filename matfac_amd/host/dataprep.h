// dataprep.h -- the file-level data preparation steps in front of the training path:
// the train/test/val splitter and the synthetic-matrix writer of the reference's io.cpp, same names and arguments.
#ifndef MFHOST_DATAPREP_H_
#define MFHOST_DATAPREP_H_
#include <vector>

#include "csr.h"

// io.cpp:410-459: colour every rating (0 train, 1 test, 2 val) with std::mt19937(seed) -- testPc*nnz draws WITH
// replacement for test, then draws until valPc*nnz uncoloured ratings became val -- split (gk_csr_Split keeps the
// shape and the order inside a row) and write the three matrices as text CSR.
void writeTrainTestValMat(csr_t* mat, const char* trainFileName, const char* testFileName, const char* valFileName,
                          float testPc, float valPc, int seed);
// the colouring alone (the three files are a pure function of it); color has nnz entries
void trainTestValColors(int64_t nnz, float testPc, float valPc, int seed, int* color);
// gk_csr_Split(mat, color) for one colour: same nrows / ncols, the entries of that colour in row order
csr_t* csr_take_color(const csr_t* mat, const int* color, int which);

// io.cpp:726-787: nnz distinct (user, item) pairs -- one random item per user, one random user per item, then uniform
// pairs until at least nnz are distinct -- written as text CSR with rating = uFac[u] . iFac[item], items ascending.
void writeRandMatCSR(const char* opFileName, std::vector<std::vector<double>>& uFac, std::vector<std::vector<double>>& iFac,
                     int facDim, int seed, int nnz);

// gk_csr_Write(mat, file, GK_CSR_FMT_CSR, writevals = 1, numbering = 0) as GKlib prints it: " %d %f" per entry
int csr_write_text_gk(const csr_t* m, const char* path);
#endif
