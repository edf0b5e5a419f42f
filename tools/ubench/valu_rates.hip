// valu_rates.hip -- issue cost (cycles per wave-instruction) of the VALU ops the traversal uses,
// measured with s_memtime on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 512
template <int OP>
__global__ void bench(double *out, unsigned long long *cycles, double seed) {
    double a[8]; float f[8]; int k[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i * 0.37 + threadIdx.x * 1e-3; f[i] = (float)a[i]; k[i] = i + threadIdx.x; }
    const double m = seed * 1.0000001, c = seed * 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] * m;                               // v_mul_f64
            if (OP == 1) a[i] = a[i] + c;                               // v_add_f64
            if (OP == 2) a[i] = __builtin_fma(a[i], m, c);              // v_fma_f64
            if (OP == 3) { f[i] = (float)a[i]; a[i] = a[i] + c; }       // v_cvt_f32_f64 (+ add to keep a chain)
            if (OP == 4) f[i] = f[i] * 1.0001f + 0.5f;                  // v_fma_f32 (contracted or mul+add)
            if (OP == 5) f[i] = (f[i] < (float)c) ? f[i] + 1.0f : f[i] - 1.0f;   // cmp + cndmask (+add)
            if (OP == 6) k[i] = (k[i] ^ (k[i] >> 3)) + 7;               // int ops
            if (OP == 7) a[i] = a[i] / m;                               // f64 divide
            if (OP == 8) a[i] = 1.0 / a[i] + c;                         // f64 reciprocal-ish
            if (OP == 9) f[i] = 1.0f / f[i] + 0.5f;                     // f32 divide
            if (OP == 10) a[i] = __builtin_sqrt(a[i]) + c;              // f64 sqrt
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i] + f[i] + k[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int waves_per_block, int blocks) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * 64 * waves_per_block * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("%-28s waves/block %2d blocks %5d : %8.1f cycles per op-instance per wave (x%d waves/SIMD)\n", name, waves_per_block, blocks,
           avg / (REP * 8.0), waves_per_block / 4 > 0 ? waves_per_block / 4 : 1);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 4, 16}) {
        run<0>("v_mul_f64", w, 256); run<1>("v_add_f64", w, 256); run<2>("v_fma_f64", w, 256);
        run<3>("v_cvt_f32_f64 + v_add_f64", w, 256); run<4>("f32 mul+add", w, 256); run<5>("f32 cmp+cndmask+add", w, 256);
        run<6>("int xor/shift/add (3 ops)", w, 256); run<7>("f64 divide", w, 256); run<8>("f64 1/x + add", w, 256);
        run<9>("f32 divide + add", w, 256); run<10>("f64 sqrt + add", w, 256);
        printf("\n");
    }
    return 0;
}
