// Cycle breakdown of one recurrent row evaluation (diagnostic; includes the product's device header).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../stochastic-muzero_amd/csrc/smz_mlp_device.hpp"
using namespace smz_mlp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
extern __shared__ float4 lds4[];
__global__ void __launch_bounds__(512) probe(smz_mlp_desc d, const float *weights, unsigned long long *out, int waves_active, int reps) {
    float *lds = reinterpret_cast<float *>(lds4);
    const smz_mlp_desc dl = lds_desc_without_rep(d);
    stage_weights_without_rep(lds, weights, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x / 64;
    if (wave >= waves_active) return;
    float *scratch = lds + dl.total_floats + wave * scratch_floats(d);
    const int S = d.S, A = d.A, K4in = up4(S + A), K4h = up4(d.H), K4s = up4(S);
    float *xb = scratch + row_scratch_floats(d) - 0;  // reuse region after scratch as input (allocated by host)
    xb = lds + dl.total_floats + 8 * scratch_floats(d) + wave * K4in;
    for (int k = lane; k < K4in; k += 64) xb[k] = 0.01f * k;
    lds_sync();
    unsigned long long t[8] = {0};
    unsigned long long a, b;
    float sink = 0.f;
    for (int r = 0; r < reps; r++) {
        const int rs = row_scratch_floats(d), kin = rs - K4h - K4s;
        float *tA[1] = {scratch + kin};
        float *hbuf = tA[0] + K4h;
        const float *xin[1] = {xb};
        const MatOff m1[1] = {pick(dl, true, M_DYN_IN, M_DYN_IN)}, m1m[1] = {pick(dl, true, M_DYN_MID, M_DYN_MID)},
                     m3[1] = {pick(dl, true, M_PRE_IN, M_PRE_IN)}, m3m[1] = {pick(dl, true, M_PRE_MID, M_PRE_MID)};
        float acc[1][1];
        a = __builtin_amdgcn_s_memtime();
        trunk<1, 1>(lds, dl, m1, m1m, xin, K4in, tA, lane);
        b = __builtin_amdgcn_s_memtime(); t[0] += b - a; a = b;
        { const float *W[1] = {lds + dl.off[M_DYN_OUT]}, *Bv[1] = {lds + dl.off[M_COUNT + M_DYN_OUT]}, *Ac[1] = {tA[0]};
          dense<1, 1>(W, Bv, Ac, K4h, dl.OP, lane, acc); }
        b = __builtin_amdgcn_s_memtime(); t[1] += b - a; a = b;
        float rew = decode_scale_lanes<1>(acc[0], S, lane, hbuf, nullptr);
        b = __builtin_amdgcn_s_memtime(); t[2] += b - a; a = b;
        lds_sync();
        b = __builtin_amdgcn_s_memtime(); t[3] += b - a; a = b;
        const float *Hc[1] = {hbuf};
        trunk<1, 1>(lds, dl, m3, m3m, Hc, K4s, tA, lane);
        b = __builtin_amdgcn_s_memtime(); t[4] += b - a; a = b;
        { const float *W[1] = {lds + dl.off[M_PRE_OUT]}, *Bv[1] = {lds + dl.off[M_COUNT + M_PRE_OUT]}, *Ac[1] = {tA[0]};
          dense<1, 1>(W, Bv, Ac, K4h, dl.OP, lane, acc); }
        b = __builtin_amdgcn_s_memtime(); t[5] += b - a; a = b;
        float pol[1];
        float val = softmax_decode_lanes<1>(acc[0], A, S, lane, hbuf);
        lds_sync();
        b = __builtin_amdgcn_s_memtime(); t[6] += b - a; a = b;
        sink += rew + val;
    }
    if (lane == 0 && wave == 0) { for (int i = 0; i < 7; i++) out[i] = t[i]; out[7] = (unsigned long long)sink; }
}
int main() {
    smz_mlp_desc d = {}; d.obs = 4; d.A = 2; d.S = 31; d.H = 64; d.L = 0;
    if (smz_mlp_layout(&d) != 0) { printf("layout failed\n"); return 1; }
    std::vector<float> w(d.total_floats);
    for (int i = 0; i < d.total_floats; i++) w[i] = 0.01f * ((i * 37) % 19 - 9);
    float *dw; unsigned long long *dout;
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dout, 64));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)d.total_floats + 8 * scratch_floats(d) + 8 * 64) * 4;
    CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char *names[7] = {"trunk1(in+elu)", "dense out1", "decode reward", "scale+sync", "trunk2", "dense out2", "softmax+decode"};
    for (int wa : {1, 8}) {
        const int reps = 200;
        hipLaunchKernelGGL(probe, dim3(wa == 1 ? 1 : 256), dim3(512), lds, 0, d, dw, dout, wa, reps);
        CK(hipDeviceSynchronize());
        unsigned long long h[8]; CK(hipMemcpy(h, dout, 64, hipMemcpyDeviceToHost));
        unsigned long long tot = 0; for (int i = 0; i < 7; i++) tot += h[i];
        printf("waves/CU %d (memtime ticks per row): total %.0f |", wa, (double)tot / reps);
        for (int i = 0; i < 7; i++) printf(" %s %.0f", names[i], (double)h[i] / reps);
        printf("\n");
    }
    return 0;
}
