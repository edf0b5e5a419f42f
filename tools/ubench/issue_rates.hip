// issue_rates.hip -- how many wave-instructions per second one SIMD sustains, by instruction kind and by the
// number of waves resident on the SIMD, timed with HIP events (wall clock), plus the tick rate of s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 4096
template <int OP>
__global__ void bench(float *out, unsigned long long *ticks, float seed) {
    float f[8]; double a[8]; typedef float f2 __attribute__((ext_vector_type(2))); f2 p[8]; int k[8];
    for (int i = 0; i < 8; i++) { f[i] = seed + i * 0.37f + threadIdx.x * 1e-3f; a[i] = f[i]; p[i] = f2{f[i], f[i] + 1.0f}; k[i] = i + (int)blockIdx.x; }
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[i]));
            if (OP == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
            if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(seed));
            if (OP == 6) asm volatile("s_add_u32 %0, %0, %0" : "+s"(k[i]) :: "scc");
            if (OP == 7) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(seed) : "vcc");
            if (OP == 8) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(k[i]) : "v"(f[i]));
            if (OP == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; i++) s += f[i] + (float)a[i] + p[i].x + p[i].y + k[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = m1 - m0; ticks[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int OP>
void run(const char *name, int ops_per_instance, int waves_per_simd, int cus) {
    // blocks of four waves (one per SIMD of a CU); waves_per_simd blocks per CU make the load per SIMD
    const int blocks = cus * waves_per_simd;
    float *out; unsigned long long *tk;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipMalloc(&tk, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, tk, 1.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, tk, 1.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), tk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double mt = 0, wc = 0, wc_max = 0; for (int i = 0; i < blocks; i++) { mt += h[2 * i]; wc += h[2 * i + 1]; if (h[2 * i + 1] > wc_max) wc_max = h[2 * i + 1]; }
    mt /= blocks; wc /= blocks;
    const double instr_per_wave = (double)REP * 8 * ops_per_instance;
    // SIMD throughput from the whole kernel (events, minus ~8 us of launch): every wave's instructions over the time
    // the kernel took -- the per-wave timer alone flatters (the oldest wave of a SIMD is served first)
    const double ns_total = (ms * 1e6 - 8000.0);
    printf("%-18s waves/SIMD %2d : a wave issues one every %5.2f ns (mean; slowest %5.2f) ; the SIMD one every %5.2f ns = %4.2f clocks at 2.4 GHz  (s_memtime %4.0f MHz, kernel %.3f ms)\n",
           name, waves_per_simd, wc * 10.0 / instr_per_wave, wc_max * 10.0 / instr_per_wave, ns_total / (instr_per_wave * waves_per_simd),
           ns_total / (instr_per_wave * waves_per_simd) * 2.4, mt / (wc * 10.0) * 1e3, ms);
    hipFree(out); hipFree(tk);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    for (int w : {1, 2, 4, 7, 8}) {
        run<0>("v_fma_f32", 1, w, cus); run<1>("v_pk_fma_f32", 1, w, cus); run<2>("v_fma_f64", 1, w, cus); run<9>("v_mul_f64", 1, w, cus);
        run<3>("v_add_f32", 1, w, cus); run<4>("v_min_f32", 1, w, cus); run<5>("v_max3_f32", 1, w, cus); run<6>("s_add_u32", 1, w, cus);
        run<7>("v_cmp+v_cndmask", 2, w, cus); run<8>("v_readlane_b32", 1, w, cus);
        printf("\n");
    }
    return 0;
}
