#include "common.hpp"
SimCtx* g_sim = nullptr;
thread_local dim3 threadIdx, blockIdx;
#include "lstm.hip"
