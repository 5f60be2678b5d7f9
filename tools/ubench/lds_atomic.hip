// Dev microbenchmark: throughput of LDS float atomics (ds_add_f32) under the address patterns of the
// byte-table gradient.  Build: hipcc -O3 --offload-arch=gfx950 -o lds_atomic lds_atomic.hip ; run: ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int kRows = 458, kDb = 48, kIters = 4096;

// mode 0: conflict-free (lane -> consecutive dwords, row advances per iteration)
// mode 1: byte-gradient pattern: 64 consecutive elements e of a 768-wide row -> slot e/48 -> row ids[slot], col e%48
// mode 2: as 1 but every id is the same row (pad)
// mode 3: as 1 with plain read-add-write instead of the atomic (races; timing only)
// mode 4: as 1 but ds_add_rtn (returning)
// mode 5: as 1 with the pad lanes (30 %) masked off          mode 6: as 1 with only 16 of 64 lanes active
// mode 7: as 1 on uint32 (ds_add_u32)                          mode 8: as 1 on uint64 (ds_add_u64; table of 24 columns)
template <int MODE>
__global__ __launch_bounds__(512) void k(const int *__restrict__ ids, float *__restrict__ out, int stride) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kRows * stride; i += 512) lds[i] = 0.f;
    __syncthreads();
    float acc = 0.f;
    const int *myids = ids + (blockIdx.x * 8 + wave) * 16 * 64;  // 64 "tokens" of 16 slots, reused
    for (int it = 0; it < kIters; ++it) {
        const int tokn = it & 63;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int e = lane + 64 * j;
            int addr;
            if (MODE == 0) addr = ((it * 12 + j) % (kRows - 1)) * stride + (lane % kDb) + (lane >= kDb ? stride : 0);
            else {
                const int slot = e / kDb, wi = e - slot * kDb;
                const int id = MODE == 2 ? 456 : myids[tokn * 16 + slot];
                addr = id * stride + wi;
            }
            const float v = (float)(it + j);
            if (MODE == 5) { if (addr / stride != 456) atomicAdd(&lds[addr], v); continue; }
            if (MODE == 6) { if ((lane & 3) == 0) atomicAdd(&lds[addr], v); continue; }
            if (MODE == 7) { atomicAdd((unsigned *)&lds[addr], (unsigned)(it + j)); continue; }
            if (MODE == 8) { atomicAdd((unsigned long long *)lds + (addr >> 1), (unsigned long long)(it + j)); continue; }
            if (MODE == 3) lds[addr] += v;
            else if (MODE == 4) acc += atomicAdd(&lds[addr], v);
            else atomicAdd(&lds[addr], v);
        }
    }
    __syncthreads();
    float s = acc;
    for (int i = tid; i < kRows * stride; i += 512) s += lds[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char *name, const int *ids, float *out, int stride) {
    size_t lds = (size_t)kRows * stride * 4;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 512, lds>>>(ids, out, stride);
    hipEventRecord(a);
    k<MODE><<<256, 512, lds>>>(ids, out, stride);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)kIters * 12 * 8;   // wave-instructions per CU
    printf("%-44s stride %2d: %.3f ms, %.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, stride, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
    std::vector<int> h(256 * 8 * 16 * 64);
    srand(1);
    for (auto &x : h) { int r = rand() % 100; x = r < 30 ? 456 : (r < 45 ? 32 : rand() % 256); }  // 30 % pad, 15 % space, rest spread
    int *ids; float *out;
    hipMalloc(&ids, h.size() * 4); hipMalloc(&out, 4);
    hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int stride : {48}) {
        run<0>("conflict-free ds_add_f32", ids, out, stride);
        run<1>("byte-gradient pattern ds_add_f32", ids, out, stride);
        run<2>("all lanes on the pad row ds_add_f32", ids, out, stride);
        run<3>("byte-gradient pattern, plain read-add-write", ids, out, stride);
        run<4>("byte-gradient pattern, ds_add_rtn_f32", ids, out, stride);
        run<5>("byte-gradient pattern, pad lanes masked", ids, out, stride);
        run<6>("byte-gradient pattern, 16 of 64 lanes", ids, out, stride);
        run<7>("byte-gradient pattern, ds_add_u32", ids, out, stride);
        run<8>("byte-gradient pattern, ds_add_u64", ids, out, stride);
    }
    return 0;
}
