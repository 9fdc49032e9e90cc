// Does the immediate `offset:` of an LDS-DMA (buffer_load_dwordx4 ... lds) move BOTH the global address and the LDS destination?
// One M0 write then serves several 1 KB pieces of a stage.   hipcc --offload-arch=gfx950 -O3 -o m0_offset_probe m0_offset_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, unsigned nbytes)
{
    __shared__ __attribute__((aligned(16))) char smem[8192];
    for (int i = threadIdx.x; i < 2048; i += 64) reinterpret_cast<float*>(smem)[i] = -1.f;
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    i32x4 r; r.x = (int)(size_t)x; r.y = (int)(((size_t)x >> 32) & 0xffff); r.z = nbytes; r.w = 0x00020000;
    r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y); r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
    const unsigned vo = threadIdx.x * 16;
    asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(vo), "s"(r), "{m0}"(base) : "memory");
    asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:2048 lds" :: "v"(vo), "s"(r), "{m0}"(base) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) y[i] = reinterpret_cast<float*>(smem)[i];
}
int main()
{
    float *x, *y;
    hipMalloc(&x, 16384); hipMalloc(&y, 8192);
    std::vector<float> h(4096), o(2048);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMemcpy(x, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 16384u);
    hipMemcpy(o.data(), y, 8192, hipMemcpyDeviceToHost);
    // piece 0 -> LDS floats [0, 256) = x[0..256).  piece with offset:2048 -> where, and from where?
    int first = -1;
    for (int i = 256; i < 2048; ++i) if (o[i] >= 0.f) { first = i; break; }
    printf("piece0: lds[0]=%g lds[255]=%g | offset:2048 piece landed at lds float index %d (byte %d) holding x[%g]\n", o[0], o[255], first, first * 4, first >= 0 ? o[first] : -1.f);
    printf("=> LDS destination %s by the immediate offset, global address %s\n", first == 512 ? "MOVES" : "does NOT move", (first >= 0 && o[first] == 512.f) ? "MOVES" : "does NOT move");
    return 0;
}
