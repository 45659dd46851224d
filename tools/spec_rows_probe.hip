// Stage 1 of the speculative-evaluation study: what a wavefront pays for a FOUR-row network pass (both
// children of both of its trees' new leaves) against today's two-row pass, with the product's own device code
// (smz_mlp::recurrent_rows on the compact LDS weight image of the LDS-resident search kernel), one or two wavefronts per SIMD,
// all 256 CUs busy.  s_memtime ticks (100 MHz constant clock on gfx950: x 24 = core cycles at 2.4 GHz) per pass and wave.
//   variants: 2 rows, same branch (weights read once: today's pass when the wave's two leaves share a branch)
//             2 rows, two branches (today's pass otherwise)
//             2 x (2 rows, same branch) back to back   = four rows with today's code, sibling pairs
//             4 rows, one branch (weights read once for four rows: both trees' leaves on the same branch)
//             4 rows as two sibling pairs on two branches (two weight sets, each applied to two rows)
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o tools/spec_rows_probe tools/spec_rows_probe.hip \
//        stochastic-muzero_amd/csrc/smz_mlp.o   (smz_mlp_layout lives there)   -- or see tools/spec_rows_probe.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../stochastic-muzero_amd/csrc/smz_mlp_device.hpp"
using namespace smz_mlp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
extern __shared__ float4 lds4[];

// four rows as two sibling pairs: rows 0,1 use matrix set 0, rows 2,3 use set 1 (dense<U,R> with per-row pointers reads a
// pair's weights twice; this variant reads each set once -- what a purpose-built four-row pass would do)
template <int VAR>
__global__ void __launch_bounds__(512) probe(smz_mlp_desc d, const float *weights, float *hidden_out, unsigned long long *out, int waves, int reps) {
    // the dimensions are compile-time constants of everything inlined below, as in the specialised search kernel (AEX)
    d.A = 2; d.S = 31; d.H = 64; d.L = 0; d.OP = kWave; d.obs = 4;
    float *lds = reinterpret_cast<float *>(lds4);
    const smz_mlp_desc dl = lds_desc_compact(d);
    stage_weights_compact(lds, weights, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x / 64;
    if (wave >= waves) return;
    const int K4in = up4(d.S + d.A);
    const int rs = row_scratch_floats(d);
    float *scratch = lds + ((dl.total_floats + 3) & ~3) + wave * (4 * rs + 4 * K4in + 16);
    float *xall = scratch + 4 * rs;
    float *outs = xall + 4 * K4in;
    for (int k = lane; k < 4 * K4in; k += 64) xall[k] = (k % K4in) < d.S ? 0.01f * (k % 31) : ((k % K4in) == d.S + (k / K4in) % 2 ? 1.f : 0.f);
    lds_sync();
    float *gh = hidden_out + ((size_t)blockIdx.x * waves + wave) * 4 * 32;
    float sink = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        // the branch pattern alternates so that the compiler cannot fold it: wave-uniform, as in the kernel (readlane of L.branch)
        const bool b0 = ((r + blockIdx.x) & 1) != 0;
        if (VAR == 0 || VAR == 2) {                       // 2 rows same branch (x2 for VAR 2)
            for (int rep = 0; rep < (VAR == 2 ? 2 : 1); rep++) {
                const float *xin[2] = {xall + rep * 2 * K4in, xall + (rep * 2 + 1) * K4in};
                const bool dyn[2] = {b0, b0}, live[2] = {true, true};
                float *dh[2] = {gh + rep * 64, gh + rep * 64 + 32}, *dp[2] = {outs + rep * 8, outs + rep * 8 + 4};
                float rw[2], vl[2];
                recurrent_rows<1, 2, true, true>(lds, dl, scratch, xin, dyn, live, dh, dp, rw, vl);
                sink += rw[0] + vl[1];
            }
        } else if (VAR == 1) {                            // 2 rows, two branches
            const float *xin[2] = {xall, xall + K4in};
            const bool dyn[2] = {b0, !b0}, live[2] = {true, true};
            float *dh[2] = {gh, gh + 32}, *dp[2] = {outs, outs + 4};
            float rw[2], vl[2];
            recurrent_rows<1, 2, false, true>(lds, dl, scratch, xin, dyn, live, dh, dp, rw, vl);
            sink += rw[0] + vl[1];
        } else if (VAR == 3) {                            // 4 rows, one branch
            const float *xin[4] = {xall, xall + K4in, xall + 2 * K4in, xall + 3 * K4in};
            const bool dyn[4] = {b0, b0, b0, b0}, live[4] = {true, true, true, true};
            float *dh[4] = {gh, gh + 32, gh + 64, gh + 96}, *dp[4] = {outs, outs + 4, outs + 8, outs + 12};
            float rw[4], vl[4];
            recurrent_rows<1, 4, true, true>(lds, dl, scratch, xin, dyn, live, dh, dp, rw, vl);
            sink += rw[0] + vl[3];
        } else {                                          // 4 rows, per-row matrices (two branches, pairs): upper bound (weights read 4x)
            const float *xin[4] = {xall, xall + K4in, xall + 2 * K4in, xall + 3 * K4in};
            const bool dyn[4] = {b0, b0, !b0, !b0}, live[4] = {true, true, true, true};
            float *dh[4] = {gh, gh + 32, gh + 64, gh + 96}, *dp[4] = {outs, outs + 4, outs + 8, outs + 12};
            float rw[4], vl[4];
            recurrent_rows<1, 4, false, true>(lds, dl, scratch, xin, dyn, live, dh, dp, rw, vl);
            sink += rw[0] + vl[3];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { atomicAdd(&out[0], t1 - t0); if (sink == 12345.f) out[1] = 1; }
}

// One pass of the two-row variants on per-wave pseudo-random inputs; everything the pass produces goes to `dump`
// ([workgroup][wave][variant 0 / 1][row][40 floats]: 32 hidden | 4 policy | reward | value): two builds of this probe (e.g.
// -DSMZ_PAIR_TAILS=0 / 1) must write identical files.
__global__ void __launch_bounds__(512) dump_kernel(smz_mlp_desc d, const float *weights, float *dump, int waves) {
    d.A = 2; d.S = 31; d.H = 64; d.L = 0; d.OP = kWave; d.obs = 4;
    float *lds = reinterpret_cast<float *>(lds4);
    const smz_mlp_desc dl = lds_desc_compact(d);
    stage_weights_compact(lds, weights, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x / 64;
    if (wave >= waves) return;
    const int K4in = up4(d.S + d.A), rs = row_scratch_floats(d);
    float *scratch = lds + ((dl.total_floats + 3) & ~3) + wave * (4 * rs + 4 * K4in + 16);
    float *xall = scratch + 4 * rs, *outs = xall + 4 * K4in;
    const unsigned id = blockIdx.x * waves + wave;
    for (int k = lane; k < 2 * K4in; k += 64) {
        unsigned h = (id * 2654435761u) ^ (k * 40503u + 12345u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        const int kk = k % K4in;
        xall[k] = kk < d.S ? (float)(h & 0xffff) / 65536.f : (kk == d.S + (int)((h >> 16) & 1) ? 1.f : 0.f);
    }
    lds_sync();
    for (int var = 0; var < 2; var++) {
        for (int b = 0; b < 2; b++) {
            float *o = dump + (((size_t)id * 2 + var) * 2 + b) * 2 * 40;
            const float *xin[2] = {xall, xall + K4in};
            const bool b0 = b != 0;
            const bool dyn[2] = {b0, var ? !b0 : b0}, live[2] = {true, true};
            float *dh[2] = {o, o + 40}, *dp[2] = {outs, outs + 4};
            float rw[2], vl[2];
            if (var == 0) recurrent_rows<1, 2, true, true>(lds, dl, scratch, xin, dyn, live, dh, dp, rw, vl);
            else recurrent_rows<1, 2, false, true>(lds, dl, scratch, xin, dyn, live, dh, dp, rw, vl);
            lds_sync();
            if (lane < 2) { o[32 + lane] = outs[lane]; o[40 + 32 + lane] = outs[4 + lane]; }
            if (lane == 0) { o[36] = rw[0]; o[37] = vl[0]; o[40 + 36] = rw[1]; o[40 + 37] = vl[1]; }
        }
    }
}

template <int VAR>
static double run(const smz_mlp_desc &d, const float *dw, float *dh, unsigned long long *dout, int waves, size_t lds, int reps) {
    CK(hipFuncSetAttribute((const void *)probe<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dout, 0, 64));
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), lds, 0, d, dw, dh, dout, waves, reps);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipMemset(dout, 0, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), lds, 0, d, dw, dh, dout, waves, reps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[8]; CK(hipMemcpy(h, dout, 64, hipMemcpyDeviceToHost));
    const double ticks = (double)h[0] / (256.0 * waves) / reps;
    printf("  ticks/pass %.1f (x24 = %.0f core cycles at 2.4 GHz) | launch %.3f ms = %.2f us per pass\n", ticks, ticks * 24, ms, ms * 1e3 / reps);
    return ticks;
}

int main(int argc, char **argv) {
    smz_mlp_desc d = {}; d.obs = 4; d.A = 2; d.S = 31; d.H = 64; d.L = 0;
    if (smz_mlp_layout(&d) != 0) { printf("layout failed\n"); return 1; }
    std::vector<float> w(d.total_floats);
    for (int i = 0; i < d.total_floats; i++) w[i] = 0.01f * ((i * 37) % 19 - 9);
    float *dw, *dh; unsigned long long *dout;
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dh, (size_t)256 * 8 * 4 * 32 * 4));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)((compact_total_floats(d) + 3) & ~3) + 8 * (4 * row_scratch_floats(d) + 4 * up4(d.S + d.A) + 16)) * 4;
    printf("LDS per workgroup %zu bytes\n", lds);
    const char *names[5] = {"2 rows, one branch (today, same-branch pair)", "2 rows, two branches (today, mixed pair)",
                            "2 x (2 rows, one branch) back to back (four rows with today's pass)",
                            "4 rows, one branch (weights read once)", "4 rows, per-row matrices (weights read four times)"};
    if (argc > 1) {     // dump mode: the outputs of the two-row passes on 2048 distinct inputs -> file
        const size_t n = (size_t)256 * 8 * 2 * 2 * 2 * 40;
        float *dd; CK(hipMalloc(&dd, n * 4)); CK(hipMemset(dd, 0, n * 4));
        CK(hipFuncSetAttribute((const void *)dump_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dump_kernel, dim3(256), dim3(512), lds, 0, d, dw, dd, 8);
        CK(hipDeviceSynchronize());
        std::vector<float> h(n); CK(hipMemcpy(h.data(), dd, n * 4, hipMemcpyDeviceToHost));
        FILE *f = fopen(argv[1], "wb"); fwrite(h.data(), 4, n, f); fclose(f);
        double sum = 0; int nan = 0; for (float v : h) { if (v != v) nan++; else sum += v; }
        printf("dumped %zu floats to %s (sum %.6f, %d NaN)\n", n, argv[1], sum, nan);
        return 0;
    }
    for (int waves : {4, 8}) {
        printf("== %d wavefronts per CU (%d per SIMD), 256 workgroups\n", waves, waves / 4);
        const int reps = 400;
        double t[5];
        printf("%s\n", names[0]); t[0] = run<0>(d, dw, dh, dout, waves, lds, reps);
        printf("%s\n", names[1]); t[1] = run<1>(d, dw, dh, dout, waves, lds, reps);
        printf("%s\n", names[2]); t[2] = run<2>(d, dw, dh, dout, waves, lds, reps);
        printf("%s\n", names[3]); t[3] = run<3>(d, dw, dh, dout, waves, lds, reps);
        printf("%s\n", names[4]); t[4] = run<4>(d, dw, dh, dout, waves, lds, reps);
        printf("  ratios to the same-branch two-row pass: mixed pair %.2f | 2 x 2 rows %.2f | 4 rows one branch %.2f | 4 rows per-row %.2f\n",
               t[1] / t[0], t[2] / t[0], t[3] / t[0], t[4] / t[0]);
    }
    return 0;
}
