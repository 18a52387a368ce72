/* The reference's demo program over this library: every command-line argument is a MatrixMarket file; SparseFrame() reads,
 * analyses, factorizes (MI355X), solves and validates each of them and prints the reference's report
 * (Cholesky/Demo/demo.c + Cholesky/Source/SparseFrame.c:3323-3467; LU/Demo likewise).
 *   sf_demo matrix.mtx [more.mtx ...]        Cholesky: symmetric positive definite matrices
 *   sf_demo_lu matrix.mtx [more.mtx ...]     LU: general matrices
 * Built by `make -C sparse-matrix-factorization-library_amd/csrc` (the same source against either library: the two libraries
 * export the same entry point over their own struct layouts, as the reference's two libSparseFrame.so do). */
#include <stdio.h>

int SparseFrame(int argc, char **argv);

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s matrix.mtx [matrix.mtx ...]\n", argv[0]);
        return 2;
    }
    return SparseFrame(argc, argv) ? 1 : 0;
}
