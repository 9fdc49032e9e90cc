// Round 6: what does a vector-memory instruction cost a wave that streams v_mfma_f32_32x32x16 from one wave per SIMD?  (conv64m_kernel's
// ablations: 25 VMEM instructions per tile and wave cost 57 of 180 us whatever their addresses, coalescing or the tile-end wait.)
// Kernel<KIND, EVERY>: 8 independent MFMAs per pass + 2 v_fma fillers each; behind every EVERY-th MFMA one instruction of KIND
// (1: buffer_store_dwordx4, 32-byte segments at a 128-byte stride like the kernel's; 2: the same, linear; 3: buffer_load_dwordx4 ... lds;
// 4: global_load_dwordx4 into a register (never waited for inside the loop); 5: buffer_store_dwordx4 with an out-of-range offset).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_vmem_probe mfma_vmem_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int EVERY>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, char* buf, size_t bytes, int iters, float seed)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    i32x4 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = i32x4{0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x, 0x3f803f80};
    for (int i = 0; i < 4; ++i) b[i] = i32x4{0x3f803f80, 0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x};
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(b[i]));
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // this block's 64 KB window of the buffer
    char* base = buf + ((size_t)blockIdx.x * 4 + wv) * 16384;
    i32x4 rsrc;
    rsrc.x = (int)(size_t)base; rsrc.y = (int)(((size_t)base >> 32) & 0xffff); rsrc.z = 16384; rsrc.w = 0x00020000;
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
    // KIND 6 / 7 / 8: a 2 MB region per wave, every instruction a fresh 1 KB (misses all the way to HBM)
    char* base2 = buf + (size_t)256 * 4 * 16384 + ((size_t)blockIdx.x * 4 + wv) * (2u << 20);
    i32x4 rsrc2;
    rsrc2.x = __builtin_amdgcn_readfirstlane((int)(size_t)base2); rsrc2.y = __builtin_amdgcn_readfirstlane((int)(((size_t)base2 >> 32) & 0xffff)); rsrc2.z = 2 << 20; rsrc2.w = 0x00020000;
    const unsigned strided = (unsigned)((lane & 31) * 128 + (lane >> 5) * 16), linear = (unsigned)lane * 16u;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wv * 1024);
    i32x4 data = {1, 2, 3, (int)threadIdx.x}, ld = {0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        const unsigned rot = (unsigned)(it & 3) * 4096u;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a[m]), "v"(b[m & 3]));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[m]) : "v"(seed));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(m + 3) & 7]) : "v"(seed));
            if ((EVERY <= 8 && m % EVERY == EVERY - 1) || (EVERY > 8 && m == 7 && (it % (EVERY / 8)) == 0)) {
                if (KIND == 1) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(data), "v"(strided + rot), "s"(rsrc) : "memory");
                if (KIND == 2) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(data), "v"(linear + rot), "s"(rsrc) : "memory");
                if (KIND == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_dst), "v"(linear + rot), "s"(rsrc) : "memory");
                if (KIND == 4) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld) : "v"(linear + rot), "s"(rsrc) : "memory");
                if (KIND == 6) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld) : "v"(linear + (unsigned)(EVERY <= 8 ? it * (8 / EVERY) + m / EVERY : it / (EVERY / 8)) * 1024u), "s"(rsrc2) : "memory");
                if (KIND == 7) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_dst), "v"(linear + (unsigned)(EVERY <= 8 ? it * (8 / EVERY) + m / EVERY : it / (EVERY / 8)) * 1024u), "s"(rsrc2) : "memory");
                if (KIND == 8) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(data), "v"(linear + (unsigned)(EVERY <= 8 ? it * (8 / EVERY) + m / EVERY : it / (EVERY / 8)) * 1024u), "s"(rsrc2) : "memory");
                if (KIND == 5) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(data), "v"(0x80000000u + linear), "s"(rsrc) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    s += (float)ld.x;
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND, int EVERY>
void run(unsigned long long* d, char* buf, size_t bytes, const char* name)
{
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, EVERY>), dim3(256), dim3(256), 4096, 0, d, buf, bytes, iters, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<KIND, EVERY>), dim3(256), dim3(256), 4096, 0, d, buf, bytes, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t;
    hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);

    printf("%-34s every %d MFMAs: %6.1f ticks / MFMA  %6.2f ns / MFMA", name, EVERY, (double)t / (iters * 8.0), ms * 1e6 / (iters * 8.0));
    static double base_ns = 0.0;
    if (!KIND) base_ns = ms * 1e6 / (iters * 8.0);
    else printf("   -> %6.1f ns per VMEM instruction over the MFMA + 2 VALU stream", (ms * 1e6 / (iters * 8.0) - base_ns) * EVERY);
    printf("\n");
}

int main()
{
    unsigned long long* d;
    char* buf;
    const size_t bytes = (size_t)256 * 4 * 16384 + (size_t)256 * 4 * (2u << 20);
    hipMalloc(&d, 64);
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    run<0, 8>(d, buf, bytes, "MFMA + 2 v_fma");
    run<1, 8>(d, buf, bytes, "+ store, 32 B segments / 128 B");
    run<1, 4>(d, buf, bytes, "+ store, 32 B segments / 128 B");
    run<1, 2>(d, buf, bytes, "+ store, 32 B segments / 128 B");
    run<2, 8>(d, buf, bytes, "+ store, linear");
    run<2, 4>(d, buf, bytes, "+ store, linear");
    run<5, 4>(d, buf, bytes, "+ store, out of range");
    run<3, 8>(d, buf, bytes, "+ buffer_load ... lds");
    run<3, 4>(d, buf, bytes, "+ buffer_load ... lds");
    run<3, 2>(d, buf, bytes, "+ buffer_load ... lds");
    run<4, 4>(d, buf, bytes, "+ buffer_load to a register");
    run<6, 8>(d, buf, bytes, "+ STREAMING load to a register");
    run<6, 4>(d, buf, bytes, "+ STREAMING load to a register");
    run<7, 8>(d, buf, bytes, "+ STREAMING load ... lds");
    run<7, 4>(d, buf, bytes, "+ STREAMING load ... lds");
    run<8, 8>(d, buf, bytes, "+ STREAMING store");
    run<8, 4>(d, buf, bytes, "+ STREAMING store");
    run<0, 8>(d, buf, bytes, "MFMA + 2 v_fma");
    run<6, 16>(d, buf, bytes, "+ STREAMING load to a register");
    run<6, 32>(d, buf, bytes, "+ STREAMING load to a register");
    run<6, 64>(d, buf, bytes, "+ STREAMING load to a register");
    run<7, 16>(d, buf, bytes, "+ STREAMING load ... lds");
    run<7, 32>(d, buf, bytes, "+ STREAMING load ... lds");
    run<8, 16>(d, buf, bytes, "+ STREAMING store");
    run<8, 32>(d, buf, bytes, "+ STREAMING store");
    run<4, 16>(d, buf, bytes, "+ L1-hit load to a register");
    run<4, 32>(d, buf, bytes, "+ L1-hit load to a register");
    return 0;
}
