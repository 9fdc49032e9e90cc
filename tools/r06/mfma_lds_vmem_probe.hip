// Round 6: which ingredient of conv64m_kernel's loop makes a vector-memory instruction cost its wave ~150 cycles?  One wave per SIMD streams
// v_mfma_f32_32x32x16 (A operands in accumulation registers) + 2 VALU each; optionally (LDS) a ds_read_b128 ring -- one read per two MFMAs, used
// three reads later, as the kernel's B ring; optionally (VM) one 1 KB vector-memory instruction per wave every 8 MFMAs: a streaming load
// (fresh addresses) or a streaming store.  Times per MFMA in s_memtime ticks.   hipcc --offload-arch=gfx950 -O3 -o ... this.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool LDS, int VM, bool BAR>       // VM: 0 none, 1 streaming load to a register, 2 streaming load ... lds, 3 streaming store
__global__ __launch_bounds__(256) void probe(unsigned long long* out, char* buf, int iters, float seed)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384 / 16; i += 256) reinterpret_cast<i32x4*>(smem)[i] = i32x4{0x3f803f80, 0x3f803f80, i, 0x3f803f80};
    __syncthreads();
    i32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = i32x4{0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x, 0x3f803f80};
    for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(a[i]));
    i32x4 b[4];
    for (int i = 0; i < 4; ++i) b[i] = i32x4{0x3f803f80, 0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x};
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    char* base2 = buf + ((size_t)blockIdx.x * 4 + wv) * (2u << 20);
    i32x4 rsrc2;
    rsrc2.x = __builtin_amdgcn_readfirstlane((int)(size_t)base2); rsrc2.y = __builtin_amdgcn_readfirstlane((int)(((size_t)base2 >> 32) & 0xffff)); rsrc2.z = 2 << 20; rsrc2.w = 0x00020000;
    const unsigned linear = (unsigned)lane * 16u;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + 16384 + wv * 1024);
    const char* lp = smem + wv * 4096 + lane * 16;
    i32x4 data = {1, 2, 3, (int)threadIdx.x}, ld = {0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (LDS) { b[0] = *reinterpret_cast<const i32x4*>(lp); b[1] = *reinterpret_cast<const i32x4*>(lp + 1024); b[2] = *reinterpret_cast<const i32x4*>(lp + 2048); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (LDS && (m & 1) == 0) b[(m / 2 + 3) & 3] = *reinterpret_cast<const i32x4*>(lp + ((m / 2 + 3) & 3) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "a"(a[m]), "v"(b[(m / 2) & 3]));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[m]) : "v"(seed));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(m + 3) & 7]) : "v"(seed));
            if (m == 5) {
                const unsigned off = linear + (unsigned)(it & 2047) * 1024u;
                if (VM == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld) : "v"(off), "s"(rsrc2) : "memory");
                if (VM == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_dst), "v"(off), "s"(rsrc2) : "memory");
                if (VM == 3) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(data), "v"(off), "s"(rsrc2) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR && (it & 7) == 7) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15] + (float)b[i].x;
    s += (float)ld.x;
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <bool LDS, int VM, bool BAR>
void run(unsigned long long* d, char* buf, const char* name)
{
    const int iters = 1024;
    hipLaunchKernelGGL((probe<LDS, VM, BAR>), dim3(256), dim3(256), 32768, 0, d, buf, iters, 1.0f);
    hipLaunchKernelGGL((probe<LDS, VM, BAR>), dim3(256), dim3(256), 32768, 0, d, buf, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long t;
    hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
    printf("%-64s %6.1f ticks / MFMA\n", name, (double)t / (iters * 8.0));
}

int main()
{
    unsigned long long* d;
    char* buf;
    hipMalloc(&d, 64);
    hipMalloc(&buf, (size_t)256 * 4 * (2u << 20));
    hipMemset(buf, 0, (size_t)256 * 4 * (2u << 20));
    run<false, 0, false>(d, buf, "MFMA + 2 VALU");
    run<false, 1, false>(d, buf, "MFMA + 2 VALU, 1 streaming load / 8 MFMA");
    run<false, 2, false>(d, buf, "MFMA + 2 VALU, 1 streaming load...lds / 8 MFMA");
    run<false, 3, false>(d, buf, "MFMA + 2 VALU, 1 streaming store / 8 MFMA");
    run<true, 0, false>(d, buf, "MFMA + 2 VALU + ds_read ring");
    run<true, 1, false>(d, buf, "MFMA + 2 VALU + ds_read ring, 1 streaming load / 8 MFMA");
    run<true, 2, false>(d, buf, "MFMA + 2 VALU + ds_read ring, 1 streaming load...lds / 8 MFMA");
    run<true, 3, false>(d, buf, "MFMA + 2 VALU + ds_read ring, 1 streaming store / 8 MFMA");
    run<true, 0, true>(d, buf, "MFMA + 2 VALU + ds_read ring + barrier / 64 MFMA");
    run<true, 2, true>(d, buf, "... + barrier, 1 streaming load...lds / 8 MFMA");
    run<true, 3, true>(d, buf, "... + barrier, 1 streaming store / 8 MFMA");
    return 0;
}
