#include <hip/hip_runtime.h>
SimCtx* g_sim = nullptr;
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
char g_lds_anchor = 0;
