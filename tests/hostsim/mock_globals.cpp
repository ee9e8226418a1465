// TEST INFRASTRUCTURE ONLY: storage for the mock HIP runtime's per-lane indices (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
