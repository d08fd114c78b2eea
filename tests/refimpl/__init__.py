"""Second, independent float64 numpy implementations of reference operators, written from the reference's source text (not from oracle/*.c): they pin the C oracle and the
HIP kernels where the reference itself cannot be built or imported in this image.  Test infrastructure only."""
