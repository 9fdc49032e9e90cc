// which lane does row_shl:n read?  out[lane] = value taken by lane `lane` (own lane id where the source is outside the row)
#include <hip/hip_runtime.h>
__global__ void k(int* out)
{
    const int own = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(own, own, 0x101, 0xf, 0xf, false);
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(own, own, 0x113, 0xf, 0xf, false);
}
extern "C" int dpp_probe(int* out) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out); return (int)hipDeviceSynchronize(); }
