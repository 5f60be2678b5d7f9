// Dev microbenchmark (round 3): LDS atomic throughput under the LANE-CONTIGUOUS address pattern of embed_mix_bwd_lc/plain_kernel
// (a lane owns 12 consecutive elements of a 768-wide row: 4 lanes per 48-wide byte slot, 16 slots per wave-instruction, each slot on
// its own random table row), for the candidate accumulator types.  Build: hipcc -O3 --offload-arch=gfx950 -o lds_atomic2 lds_atomic2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int kRows = 256, kDb = 48, kIters = 2048, kThreads = 1024, kWaves = kThreads / 64;

// MODE 0 ds_add_u64 (row stride 49 qwords)      1 ds_add_u32 (stride 49 dwords)         2 ds_add_f64 (stride 49 qwords)
//      3 ds_add_u64, odd slots masked off        4 ds_add_u64, all 16 slots on distinct consecutive rows (no random collisions)
//      5 ds_add_u64 row stride 48                6 two ds_add_u32 per element (lo, hi)   7 ds_add_f32 (stride 49 dwords)
//      8 ds_add_u64, ids constant over 4 consecutive positions (L1-like reuse: does the same address pattern repeat cheaper?)
template <int MODE>
__global__ __launch_bounds__(kThreads) void k(const int *__restrict__ ids, float *__restrict__ out) {
    extern __shared__ unsigned long long lq[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stride = MODE == 5 ? 48 : 49;
    const int words = kRows * stride * ((MODE == 1 || MODE == 7) ? 1 : 2);
    unsigned *l32 = (unsigned *)lq;
    for (int i = tid; i < words; i += kThreads) l32[i] = 0u;
    __syncthreads();
    const int slot = lane >> 2, wi0 = (lane & 3) * 12;
    const int *myids = ids + ((blockIdx.x * kWaves + wave) * 64) * 16;   // 64 positions of 16 slots, reused
    for (int it = 0; it < kIters; ++it) {
        int p = it & 63;
        if (MODE == 8) p &= ~3;
        int id = myids[p * 16 + slot];
        if (MODE == 4) id = (slot * 3 + it) & 255;
        const int base = id * stride + wi0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const unsigned v = (unsigned)(it + j);
            if (MODE == 1) atomicAdd(l32 + base + j, v);
            else if (MODE == 7) atomicAdd((float *)l32 + base + j, (float)v);
            else if (MODE == 2) atomicAdd((double *)lq + base + j, (double)v);
            else if (MODE == 6) { atomicAdd(l32 + 2 * (base + j), v); atomicAdd(l32 + 2 * (base + j) + 1, v >> 3); }
            else if (MODE == 3) { if (!(slot & 1)) atomicAdd(lq + base + j, (unsigned long long)v); }
            else atomicAdd(lq + base + j, (unsigned long long)v);
        }
    }
    __syncthreads();
    unsigned s = 0;
    for (int i = tid; i < words; i += kThreads) s += l32[i];
    if (s == 0x12345678u) out[0] = (float)s;
}

template <int MODE>
static void run(const char *name, const int *ids, float *out) {
    size_t lds = (size_t)kRows * 49 * 8;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, kThreads, lds>>>(ids, out);
    hipEventRecord(a);
    k<MODE><<<256, kThreads, lds>>>(ids, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)kIters * 12 * kWaves * (MODE == 6 ? 2 : 1);
    printf("%-58s %.3f ms, %.1f ns per wave-instruction per CU = %.1f cycles at 2.1 GHz; per position (12 elements/lane): %.1f ns\n", name, ms,
           ms * 1e6 / instr_per_cu, ms * 1e-3 * 2.1e9 / instr_per_cu, ms * 1e6 / ((double)kIters * kWaves));
}

int main() {
    std::vector<int> h(256 * kWaves * 64 * 16);
    srand(1);
    for (auto &x : h) { int r = rand() % 100; x = r < 15 ? 32 : (r < 25 ? 101 : (r < 33 ? 116 : (r < 60 ? 97 + rand() % 26 : rand() % 256))); }  // text-like: space, e, t, lower case, rest
    int *ids; float *out;
    hipMalloc(&ids, h.size() * 4); hipMalloc(&out, 4);
    hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("ds_add_u64, text-like ids, row stride 49", ids, out);
    run<5>("ds_add_u64, row stride 48", ids, out);
    run<1>("ds_add_u32, row stride 49", ids, out);
    run<6>("2 x ds_add_u32 per element", ids, out);
    run<2>("ds_add_f64", ids, out);
    run<7>("ds_add_f32", ids, out);
    run<3>("ds_add_u64, every other slot masked off", ids, out);
    run<4>("ds_add_u64, 16 distinct consecutive rows", ids, out);
    run<8>("ds_add_u64, ids repeat over 4 positions", ids, out);
    return 0;
}
