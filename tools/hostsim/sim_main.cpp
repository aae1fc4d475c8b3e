#include <hip/hip_runtime.h>
SimCtx* g_sim = nullptr;
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
