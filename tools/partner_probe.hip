// What can a wave do while its SIMD partner streams v_mfma_f32_16x16x4_f32 back to back?
// 256 blocks x 8 waves, 128 KB LDS each -> one block per CU; a workgroup's waves are dealt to the 4
// SIMDs cyclically, so wave w and wave w+4 share a SIMD.  Waves 0-3 stream MFMAs (or exit at once,
// baseline); waves 4-7 are "victims" that time short instruction sequences with s_memtime at
// priority 0 or 3.  HW_ID is recorded to verify the SIMD pairing.
//   hipcc --offload-arch=gfx950 -O3 -o tools/partner_probe tools/partner_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void probe(float* buf, unsigned long long* out, int mfma_iters, int prio, int partner_on)
{
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 256) {
        if (lane == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); out[(size_t)(blockIdx.x * 4 + (tid >> 6)) * 8 + 6] = hw; }
        if (!partner_on) return;
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = buf[tid], b = buf[tid + 256];
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        f32x4 s = acc[0];
        for (int i = 1; i < 16; ++i) s += acc[i];
        buf[1 << 20 | (blockIdx.x * 256 + tid)] = s.x + s.y + s.z + s.w;
        return;
    }
    // victim: let the partner get going
    __builtin_amdgcn_s_sleep(100);
    for (int k = 0; k < 500; ++k) __builtin_amdgcn_s_sleep(10);
    if (prio) __builtin_amdgcn_s_setprio(3);
    unsigned long long t[6];
    float v0 = buf[tid], v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0); t[0] = clock64(); __builtin_amdgcn_sched_barrier(0);
    // (1) 256 independent-ish VALU (4 chains x 64 fma)
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f);
        v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
    }
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    __builtin_amdgcn_sched_barrier(0); t[1] = clock64(); __builtin_amdgcn_sched_barrier(0);
    // (2) 16 global_store_dwordx4, lane stride 256 B (the NHWC epilogue pattern)
    float* o = buf + (2 << 20) + (size_t)(blockIdx.x * 4 + ((tid >> 6) & 3)) * 64 * 64;
    f32x4 val = {v0, v1, v2, v3};
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(o + (lane & 15) * 64 + (lane >> 4) * 4 + (i & 3) * 16 + (i >> 2) * 1024) = val;
    __builtin_amdgcn_sched_barrier(0); t[2] = clock64(); __builtin_amdgcn_sched_barrier(0);
    // (3) 8 ds_write_b128 + wait
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(smem + (tid + i * 256) * 16) = val;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0); t[3] = clock64(); __builtin_amdgcn_sched_barrier(0);
    // (4) 8 global loads (L2 hits) + wait
    f32x4 ld[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ld[i] = *reinterpret_cast<const f32x4*>(buf + (tid + i * 256) * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0); t[4] = clock64(); __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_sched_barrier(0); t[5] = clock64(); __builtin_amdgcn_sched_barrier(0);
    f32x4 sum = ld[0];
    for (int i = 1; i < 8; ++i) sum += ld[i];
    if (sum.x == 12345.f) buf[0] = sum.y;
    if (lane == 0) {
        unsigned long long* d = out + (size_t)(blockIdx.x * 4 + ((tid >> 6) & 3)) * 8;
        for (int i = 0; i < 5; ++i) d[i] = t[i + 1] - t[i];
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); d[7] = hw;
    }
}

int main()
{
    float* buf; unsigned long long* out;
    hipMalloc(&buf, (size_t)128 << 20); hipMemset(buf, 0, (size_t)128 << 20);
    hipMalloc(&out, 256 * 4 * 8 * 8);
    const char* names[5] = {"256 v_fma", "16 strided store_x4", "8 ds_write_b128+wait", "8 load_x4 (L2)+wait", "barrier"};
    for (int partner = 0; partner <= 1; ++partner)
        for (int prio = 0; prio <= 1; ++prio) {
            hipMemset(out, 0, 256 * 4 * 8 * 8);
            probe<<<256, 512>>>(buf, out, 6000, prio, partner);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256 * 4 * 8);
            hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            printf("partner MFMA stream %s, victim prio %d:\n", partner ? "ON " : "off", prio ? 3 : 0);
            int same = 0;
            for (int w = 0; w < 1024; ++w) same += ((h[w * 8 + 6] >> 4) & 3) == ((h[w * 8 + 7] >> 4) & 3) && ((h[w * 8 + 6] >> 8) & 0xf) == ((h[w * 8 + 7] >> 8) & 0xf);
            printf("   wave w and w+4 on the same SIMD of the same CU: %d / 1024\n", same);
            for (int k = 0; k < 4; ++k) {
                std::vector<unsigned long long> v;
                for (int w = 0; w < 1024; ++w) v.push_back(h[w * 8 + k]);
                std::sort(v.begin(), v.end());
                printf("   %-22s p10 %6llu  p50 %6llu  p90 %6llu cycles\n", names[k], v[102], v[512], v[921]);
            }
        }
    return 0;
}
