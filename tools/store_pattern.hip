// Cost of one wave issuing 16 x global_store_dwordx4 / global_load_dwordx4 under different lane->address
// maps (all write/read the same 16 KB per wave = 64 pixels x 64 channels fp32 NHWC).
//   A  MFMA-native: lane (p=l&15, kq=l>>4) -> pixel p, 16 B at channel 4kq   (64 requests of 16 B)
//   B  quad-contiguous: lane l -> pixel l>>2, 16 B at channel 4(l&3)         (16 segments of 64 B)
//   C  fully contiguous: lane l -> byte l*16                                 (1 KB contiguous)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, bool LOAD>
__global__ __launch_bounds__(256) void k(float* buf, unsigned long long* out)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* o = buf + (size_t)(blockIdx.x * 4 + wv) * 64 * 64;       // this wave's 64 px x 64 ch
    f32x4 v = {1.f * tid, 2.f, 3.f, 4.f}, acc = {0, 0, 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int off;   // float offset
        if (MODE == 0) off = ((i >> 2) * 16 + (lane & 15)) * 64 + (i & 3) * 16 + (lane >> 4) * 4;
        else if (MODE == 1) off = ((i >> 2) * 16 + (lane >> 2)) * 64 + (i & 3) * 16 + (lane & 3) * 4;
        else off = i * 256 + lane * 4;
        if (LOAD) acc += *reinterpret_cast<const f32x4*>(o + off);
        else *reinterpret_cast<f32x4*>(o + off) = v;
    }
    if (LOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    if (LOAD && acc.x == 123.f) buf[0] = acc.y;
    if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
}
template <int MODE, bool LOAD> void run(const char* n, float* buf, unsigned long long* out)
{
    const int blocks = 2048;
    k<MODE, LOAD><<<blocks, 256>>>(buf, out); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k<MODE, LOAD><<<blocks, 256>>>(buf, out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s issue 16 instr: p10 %5llu p50 %5llu p90 %5llu cycles | kernel %.3f ms = %.2f TB/s\n", n, h[h.size() / 10],
           h[h.size() / 2], h[h.size() * 9 / 10], ms, blocks * 4 * 16384.0 / ms / 1e9);
}
int main()
{
    float* buf; unsigned long long* out;
    hipMalloc(&buf, (size_t)2048 * 4 * 16384 + 4096); hipMemset(buf, 0, (size_t)2048 * 4 * 16384);
    hipMalloc(&out, 2048 * 4 * 8);
    run<0, false>("STORE A mfma-native (64 x 16 B / instr)", buf, out);
    run<1, false>("STORE B quad-contiguous (16 x 64 B / instr)", buf, out);
    run<2, false>("STORE C contiguous 1 KB / instr", buf, out);
    run<0, true>("LOAD  A mfma-native (64 x 16 B / instr)", buf, out);
    run<1, true>("LOAD  B quad-contiguous (16 x 64 B / instr)", buf, out);
    run<2, true>("LOAD  C contiguous 1 KB / instr", buf, out);
    return 0;
}
