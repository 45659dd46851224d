// smz_mlp.hip -- stand-alone fused `mlp_model` head kernels (C ABI: smz_mlp_layout / _initial / _recurrent).
//
// The networks are tiny (ckpt 421: 10 k multiply-adds per leaf) and at 4096 leaves per round every library GEMM is
// launch-bound, so one kernel does the whole chain for a row: input layer -> ELU -> (shared hidden layer -> ELU) x L ->
// output heads -> min-max scaling / softmax / support decode, weights staged once per workgroup in LDS.  A wavefront
// processes its rows one after another; each row uses only the pair of networks its branch flag selects
// (monte_carlo_tree_search.py:333-342).  Network outputs are held to the 1e-5 class, not bit parity.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/smz.h"
#include "smz_mlp_device.hpp"

using namespace smz_mlp;

namespace {

extern __shared__ float4 smz_mlp_lds4[];

// LDS map: [0, total_floats) the packed weights at their buffer offsets; then per-wave scratch
// FASTD: the reference's shipped network shape (S 31, H 64, L 0) with the dimensions as compile-time constants
// (layer loops unroll; see k_search_mlp in smz_kernels.hip); A stays a run-time value here.
constexpr int kFastS = 31, kFastH = 64, kFastL = 0;
template <int U, bool FASTD>
__global__ void __launch_bounds__(512) k_mlp_recurrent(smz_mlp_desc d, const float *weights, const float *x,
                                                       const uint8_t *branch, float *hidden_out, float *reward_out,
                                                       float *policy_out, float *value_out, int B, int rows_per_wave) {
    float *lds = reinterpret_cast<float *>(smz_mlp_lds4);
    stage_recurrent_weights(lds, weights, d);
    if (FASTD) { d.S = kFastS; d.H = kFastH; d.L = kFastL; d.OP = kWave; }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, waves = blockDim.x / kWave;
    float *scratch = lds + d.total_floats + wave * scratch_floats(d);
    const int row0 = (blockIdx.x * waves + wave) * rows_per_wave;
    const int S = d.S, A = d.A, K4in = up4(S + A), rs = row_scratch_floats(d);
    for (int i = 0; i < rows_per_wave; i += kRows) {
        if (row0 + i >= B) break;                  // wave-uniform
        const float *xin[kRows];
        bool dyn[kRows], live[kRows];
        float *dh[kRows], *dp[kRows];
        float reward[kRows], value[kRows];
#pragma unroll
        for (int r = 0; r < kRows; r++) {
            const int row = row0 + i + r;
            live[r] = (i + r < rows_per_wave) && row < B;
            const int rr = live[r] ? row : row0 + i;
            float *xb = scratch + r * rs;
            for (int k = lane; k < K4in; k += kWave) xb[k] = (k < S + A) ? x[(size_t)rr * (S + A) + k] : 0.f;
            xin[r] = xb;
            dyn[r] = branch[rr] != 0;
            dh[r] = hidden_out + (size_t)rr * S;
            dp[r] = policy_out + (size_t)rr * A;
        }
        lds_sync();
        recurrent_rows<U, kRows>(lds, d, scratch, xin, dyn, live, dh, dp, reward, value);
#pragma unroll
        for (int r = 0; r < kRows; r++) {
            if (live[r] && lane == 0) {
                if (reward_out) reward_out[row0 + i + r] = reward[r];
                value_out[row0 + i + r] = value[r];
            }
        }
    }
}

template <int U>
__global__ void __launch_bounds__(512) k_mlp_initial(smz_mlp_desc d, const float *weights, const float *obs,
                                                     float *hidden_out, float *policy_out, int B, int rows_per_wave) {
    float *lds = reinterpret_cast<float *>(smz_mlp_lds4);
    stage_initial_weights(lds, weights, d);
    const int wave = threadIdx.x / kWave, waves = blockDim.x / kWave;
    float *scratch = lds + d.total_floats + wave * scratch_floats(d);
    const int row0 = (blockIdx.x * waves + wave) * rows_per_wave;
    for (int i = 0; i < rows_per_wave; i++) {
        const int row = row0 + i;
        if (row >= B) break;
        initial_row<U>(lds, d, lds, d, scratch, obs + (size_t)row * d.obs, hidden_out + (size_t)row * d.S, nullptr,
                       policy_out + (size_t)row * d.A);
    }
}

constexpr int kLdsBytes = 160 * 1024;
constexpr int kWavesPerWg = 8;

template <typename Kern>
int allow_lds(Kern kern, size_t bytes) {
    // HIP caps dynamic LDS at 64 KB unless the kernel opts in (gfx950 has 160 KB per CU).  The opt-in is a host-side
    // runtime call: made once per kernel and size, not per launch.
    static size_t granted[64] = {};     // per device (the attribute belongs to the device's code object)
    static const void *who[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SMZ_ERR_HIP;
    if (who[dev] == reinterpret_cast<const void *>(kern) && bytes <= granted[dev]) return SMZ_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
        return SMZ_ERR_HIP;
    who[dev] = reinterpret_cast<const void *>(kern);
    granted[dev] = bytes;
    return SMZ_OK;
}

}  // namespace

extern "C" {

int smz_mlp_layout(smz_mlp_desc *d) {
    if (!d || d->obs < 1 || d->A < 1 || d->S < 1 || d->H < 1 || d->L < 0) return SMZ_ERR_INVALID;
    int maxo = d->H;
    if (2 * d->S > maxo) maxo = 2 * d->S;
    if (d->A + d->S > maxo) maxo = d->A + d->S;
    d->OP = kWave * ((maxo + kWave - 1) / kWave);
    if (d->OP > kWave) return SMZ_ERR_INVALID;           // one output neuron per lane: wider layers cannot fit the 160 KB LDS anyway
    const int mid = d->L > 0 ? d->H : 0;     // the shared hidden layer exists only when number_of_hidden_layer > 0
    const int K[M_COUNT] = {d->S + d->A, d->S + d->A, mid, mid, d->H, d->H, d->S, d->S, mid, mid, d->H, d->H,
                            d->obs, mid, d->H};
    int off = 0;
    for (int m = 0; m < M_COUNT; m++) { d->off[m] = off; off += up4(K[m]) * d->OP; }
    for (int m = 0; m < M_COUNT; m++) { d->off[M_COUNT + m] = off; off += d->OP; }
    d->total_floats = off;
    const size_t need = ((size_t)off + (size_t)kWavesPerWg * scratch_floats(*d)) * sizeof(float);
    return need <= (size_t)kLdsBytes ? SMZ_OK : SMZ_ERR_INVALID;
}

static int mlp_check(const smz_mlp_desc *d, const void *w) {
    if (!d || !w) return SMZ_ERR_INVALID;
    smz_mlp_desc t = *d;
    if (smz_mlp_layout(&t) != SMZ_OK || t.total_floats != d->total_floats || t.OP != d->OP) return SMZ_ERR_INVALID;
    return SMZ_OK;
}

static void mlp_geometry(int B, int &blocks, int &rows_per_wave) {
    // one workgroup of 8 waves per CU (the weights occupy most of a CU's LDS); rows spread over 256 workgroups
    rows_per_wave = (B + 256 * kWavesPerWg - 1) / (256 * kWavesPerWg);
    if (rows_per_wave < 1) rows_per_wave = 1;
    const int rows_per_block = rows_per_wave * kWavesPerWg;
    blocks = (B + rows_per_block - 1) / rows_per_block;
}

int smz_mlp_initial(const smz_mlp_desc *d, const float *weights_dev, const float *obs_dev, float *hidden_out_dev,
                    float *policy_out_dev, int B, smz_stream stream) {
    if (mlp_check(d, weights_dev) != SMZ_OK || !obs_dev || !hidden_out_dev || !policy_out_dev || B < 1) return SMZ_ERR_INVALID;
    int blocks, rpw;
    mlp_geometry(B, blocks, rpw);
    const size_t lds = ((size_t)d->total_floats + (size_t)kWavesPerWg * scratch_floats(*d)) * sizeof(float);
    if (allow_lds(k_mlp_initial<1>, lds) != SMZ_OK) return SMZ_ERR_HIP;
    hipLaunchKernelGGL((k_mlp_initial<1>), dim3(blocks), dim3(kWavesPerWg * kWave), lds, (hipStream_t)stream, *d,
                       weights_dev, obs_dev, hidden_out_dev, policy_out_dev, B, rpw);
    return hipGetLastError() == hipSuccess ? SMZ_OK : SMZ_ERR_HIP;
}

int smz_mlp_recurrent(const smz_mlp_desc *d, const float *weights_dev, const float *mlp_input_dev,
                      const uint8_t *branch_dev, float *hidden_out_dev, float *reward_out_dev, float *policy_out_dev,
                      float *value_out_dev, int B, smz_stream stream) {
    if (mlp_check(d, weights_dev) != SMZ_OK || !mlp_input_dev || !branch_dev || !hidden_out_dev || !policy_out_dev ||
        !value_out_dev || B < 1)
        return SMZ_ERR_INVALID;
    int blocks, rpw;
    mlp_geometry(B, blocks, rpw);
    const size_t lds = ((size_t)d->total_floats + (size_t)kWavesPerWg * scratch_floats(*d)) * sizeof(float);
    if (d->S == kFastS && d->H == kFastH && d->L == kFastL) {
        if (allow_lds(k_mlp_recurrent<1, true>, lds) != SMZ_OK) return SMZ_ERR_HIP;
        hipLaunchKernelGGL((k_mlp_recurrent<1, true>), dim3(blocks), dim3(kWavesPerWg * kWave), lds, (hipStream_t)stream, *d,
                           weights_dev, mlp_input_dev, branch_dev, hidden_out_dev, reward_out_dev, policy_out_dev,
                           value_out_dev, B, rpw);
    } else {
        if (allow_lds(k_mlp_recurrent<1, false>, lds) != SMZ_OK) return SMZ_ERR_HIP;
        hipLaunchKernelGGL((k_mlp_recurrent<1, false>), dim3(blocks), dim3(kWavesPerWg * kWave), lds, (hipStream_t)stream, *d,
                           weights_dev, mlp_input_dev, branch_dev, hidden_out_dev, reward_out_dev, policy_out_dev,
                           value_out_dev, B, rpw);
    }
    return hipGetLastError() == hipSuccess ? SMZ_OK : SMZ_ERR_HIP;
}

}  // extern "C"
