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
#include <stdlib.h>
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

// -------------------------------------------------------------------------------------------------------------------
// Large batches: the recurrent networks of the shipped shape (S 31, H 64, L 0) on the matrix cores, 16 leaves per wavefront.
//
// k_mlp_recurrent evaluates one or two leaves per wavefront pass and re-reads the weights from LDS for each: at 10^5 - 10^6
// leaves per launch that is the bound (1.2 ms per million leaves, ~11 % of the f32 FMA peak).  Here a wavefront takes a
// TILE of 16 leaves: a layer is v_mfma_f32_16x16x4_f32 steps with A = 16 output neurons x 4 inputs of the LDS weight
// image and B = 4 inputs x 16 leaves, so every weight read serves 16 leaves.  To round exactly like smz_mlp::dense() --
// even inputs in one accumulator (from the bias), odd inputs in another (from zero), added at the end -- an MFMA step
// takes four EVEN inputs (8c, 8c+2, 8c+4, 8c+6) or four ODD ones; inside a step the matrix core adds its four products
// in input order to the accumulator with one rounding each (an f32 MFMA is an fma chain: profiles/r02_mfma_heads_ab.txt).
// Activations sit in LDS as [input pair][leaf][2], so one ds_read_b64 per lane yields the B operands of an even and an odd
// step; the packed weight layout [input / 4][neuron][4] yields both A operands with one ds_read_b64 as well.
// The tails work on the MFMA output layout (lane = (neuron group g, leaf j), registers = neurons 16t + 4g + r of the four
// 16-neuron tiles t): sums in the association of smz_mlp::wave_sum -- registers, lanes ^ 16, ^ 32, then tiles.
// Bit-identical to k_mlp_recurrent (tests/test_gpu_mlp_heads.py).
typedef float v4f __attribute__((ext_vector_type(4)));
#ifndef SMZ_MFMA_COMPACT
#define SMZ_MFMA_COMPACT 1         // 0: round 5's launch -- the 64-wide LDS weight image (100 KB) and twelve wavefronts
#endif
// Round 6: the COMPACT LDS weight image of the single-launch search (smz_mlp_device.hpp mat_op<true>: the afterstate-dynamics output
// layer 32 wide, the two prediction output layers 36, 82 KB instead of 100) leaves room for SIXTEEN wavefronts (four per SIMD:
// 82 KB + 16 x 4 KB of tiles + 8 KB of lists = 154 of 160 KB), and the output tiles a narrow layer does not have are not
// computed (afterstate rows: 152 instead of 200 MFMAs per tile, dynamics rows 184).  Each wavefront runs one dependent chain per
// tile (MFMAs, then tails on their results): the matrix pipe was busy 32 % of the launch with three chains per SIMD
// (profiles/r06_d_pmc_262k.json); and 32 tiles per CU at 131 072 leaves split 2 + 2 over sixteen wavefronts instead of 3 + 3 + 2.
constexpr bool kMfmaCompact = SMZ_MFMA_COMPACT != 0;
constexpr int kTileLeaves = 16, kMfmaWaves = kMfmaCompact ? 16 : 12;
constexpr int kTileFloats = 32 * 32;                 // one activation tile per wavefront: [32 input pairs][16 leaves][2]

__device__ inline float lane_xor16(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(((threadIdx.x & 16) ? r[0] : r[1]));
}
__device__ inline float lane_xor32(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(((threadIdx.x & 32) ? r[0] : r[1]));
}
// sum over the 64 output positions o = 16 t + 4 g + r of a leaf, a[t][r] = the lane's values (zeros for non-members)
__device__ inline float tile_sum(const float (&a)[4][4]) {
    float s[4];
#pragma unroll
    for (int t = 0; t < 4; t++) s[t] = (a[t][0] + a[t][1]) + (a[t][2] + a[t][3]);
#pragma unroll
    for (int t = 0; t < 4; t++) s[t] = s[t] + lane_xor16(s[t]);
#pragma unroll
    for (int t = 0; t < 4; t++) s[t] = s[t] + lane_xor32(s[t]);
    return (s[0] + s[1]) + (s[2] + s[3]);
}
__device__ inline float tile_max(float m) { m = fmaxf(m, lane_xor16(m)); return fmaxf(m, lane_xor32(m)); }
__device__ inline float tile_min(float m) { m = fminf(m, lane_xor16(m)); return fminf(m, lane_xor32(m)); }

// y[t][r] = bias[16t + 4g + r] + sum_k W[k][16t + 4g + r] * x[k][leaf j]: K8 groups of eight inputs, W = LDS weight image
// (4-way interleaved, 64 outputs wide), xp = activation tile [input pair][leaf][2]
// NT: output tiles of 16 neurons the layer has (the rest of y stays zero); op: neurons per 4-input group in the LDS image
template <int K8, int NT = 4>
__device__ inline void tile_layer(const float *W, int op, const float *bias, const float *xp, int lane, v4f (&y)[4]) {
    const int g = lane >> 4, j = lane & 15;
    v4f e[4], o[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * g);
        e[t] = t < NT ? v4f{b.x, b.y, b.z, b.w} : v4f{0.f, 0.f, 0.f, 0.f};
        o[t] = v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < K8; c++) {
        // B operands: inputs 8c + 2g (even step) and 8c + 2g + 1 (odd step) of leaf j
        const float2 xb = *reinterpret_cast<const float2 *>(xp + ((4 * c + g) * kTileLeaves + j) * 2);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            // A operands: the weights of inputs 8c + 2g, 8c + 2g + 1 for neuron 16t + j (a lane beyond a narrow layer's width
            // reads the next group's weights: a finite value for an output nobody uses -- the tails mask by output index)
            const float2 wa = *reinterpret_cast<const float2 *>(W + ((2 * c + (g >> 1)) * op + 16 * t + j) * 4 + 2 * (g & 1));
            e[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, xb.x, e[t], 0, 0, 0);
            o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, xb.y, o[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) y[t] = e[t] + o[t];
}
// ELU of a trunk layer's outputs into the trunk tile (neuron n = input n of the next layer)
__device__ inline void store_trunk(float *hp, const v4f (&y)[4], int lane) {
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int n = 16 * t + 4 * g;
        *reinterpret_cast<float2 *>(hp + (((n >> 1) + 0) * kTileLeaves + j) * 2) = make_float2(elu(y[t][0]), elu(y[t][1]));
        *reinterpret_cast<float2 *>(hp + (((n >> 1) + 1) * kTileLeaves + j) * 2) = make_float2(elu(y[t][2]), elu(y[t][3]));
    }
}

// Leaves of the two branches need different networks, so a workgroup first sorts the rows of its chunk by branch (two index
// lists in LDS, filled with LDS atomics: the order inside a list is arbitrary, a leaf's result does not depend on its tile
// mates) and forms single-branch tiles; one 64 x 16 activation tile per wavefront serves every layer in turn (each layer's
// inputs are dead once its MFMAs have been issued and its outputs sit in registers).
constexpr int kMaxChunk = 2048;
// rows per workgroup pass: the batch spread over 256 workgroups (one per CU: the weights fill its LDS), whole tiles, at most
// kMaxChunk -- a small batch then is ONE round of tiles on every CU instead of two on some of them
inline int mfma_launch_waves() {
    int w = kMfmaWaves;
    if (const char *e = getenv("SMZ_MLP_MFMA_WAVES")) w = atoi(e);
    return w < 1 ? 1 : (w > kMfmaWaves ? kMfmaWaves : w);
}
inline int mfma_chunk(int B) {
    int chunk = ((B + 255) / 256 + kTileLeaves - 1) / kTileLeaves * kTileLeaves;
    if (const char *e = getenv("SMZ_MLP_CHUNK")) chunk = atoi(e);
    if (chunk < 2 * kTileLeaves) chunk = 2 * kTileLeaves;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    return chunk;
}
// ROWS: network inputs and new hidden rows live in a search handle's hidden-state storage (smz_mlp_recurrent_rows): row of
// node n of tree t = tree_hidden + (t * tree_n + n) * tree_hs; ids [B][2] = (leaf node, parent node) per tree, < 0: skip.
struct TreeRows {
    float *hidden;
    const int32_t *ids, *last_action;
    int n, hs;
};
template <int A, bool ROWS>
__global__ void __launch_bounds__(kMfmaWaves *kWave) k_mlp_recurrent_mfma(smz_mlp_desc d, const float *__restrict__ weights,
                                                                          const float *__restrict__ x, const uint8_t *__restrict__ branch,
                                                                          float *__restrict__ hidden_out, float *__restrict__ reward_out,
                                                                          float *__restrict__ policy_out, float *__restrict__ value_out,
                                                                          int B, int chunk, TreeRows tr) {
    float *lds = reinterpret_cast<float *>(smz_mlp_lds4);
    d.S = kFastS; d.H = kFastH; d.L = kFastL; d.OP = kWave; d.A = A;
    constexpr bool CP = kMfmaCompact;
    // LDS image: everything but the representation matrices (CP: narrow output layers stored narrow, mat_op<true>)
    const smz_mlp_desc dl = CP ? lds_desc_compact(d) : lds_desc_without_rep(d);
    if (CP) stage_weights_compact(lds, weights, d);
    else stage_weights_without_rep(lds, weights, d);
    constexpr int S = kFastS, half = S / 2, XW = S + A;
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int g = lane >> 4, j = lane & 15;
    float *tile = lds + dl.total_floats + wave * kTileFloats;
    const int W = (int)blockDim.x / kWave;                 // wavefronts of this launch (<= kMfmaWaves: mfma_launch_waves)
    unsigned short *list = reinterpret_cast<unsigned short *>(lds + dl.total_floats + W * kTileFloats);   // [2][kMaxChunk]
    int *cnt = reinterpret_cast<int *>(list + 2 * kMaxChunk);
    for (int base = blockIdx.x * chunk; base < B; base += gridDim.x * chunk) {
        const int n = B - base < chunk ? B - base : chunk;
        if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (ROWS && tr.ids[2 * (size_t)(base + i)] < 0) continue;        // a tree that is switched off
            const int w = branch[base + i] != 0 ? 0 : 1;
            list[w * kMaxChunk + atomicAdd(&cnt[w], 1)] = (unsigned short)i;
        }
        __syncthreads();
        const int n0 = cnt[0], n1 = cnt[1], t0 = (n0 + kTileLeaves - 1) / kTileLeaves, t1 = (n1 + kTileLeaves - 1) / kTileLeaves;
        // network inputs of a tile: 16 leaves x 40 inputs = 10 values per lane, fetched one tile ahead (the rows of a tile are
        // scattered over the batch -- or, ROWS, over the trees' hidden-state storage: a DRAM round trip the MFMAs of the
        // current tile hide)
        auto fetch = [&](int t, float (&v)[10]) {
            const bool ady = t >= t0;
            const int tt = ady ? t - t0 : t, count = (ady ? n1 : n0) - tt * kTileLeaves;
            const unsigned short *li = list + (ady ? kMaxChunk : 0) + tt * kTileLeaves;
#pragma unroll
            for (int u = 0; u < 10; u++) {
                const int i = lane + kWave * u, lf = i / 40, k = i % 40;
                const int rr = base + li[lf < count ? lf : 0];
                float val = 0.f;
                if (ROWS) {
                    if (k < S) val = tr.hidden[((size_t)rr * tr.n + tr.ids[2 * (size_t)rr + 1]) * tr.hs + k];
                    else if (k < XW) val = (k - S) == tr.last_action[rr] ? 1.f : 0.f;
                } else if (k < XW) val = x[(size_t)rr * XW + k];
                v[u] = val;
            }
        };
        // ROWS: a row of the hidden-state storage is 32 floats on a 128-byte line (31 values + a zero): lane -> leaves
        // lane / 8 and lane / 8 + 8, 16-byte piece lane % 8 -- two 16-byte loads per lane instead of ten 4-byte ones
        auto fetch_rows = [&](int t, float4 (&v)[2], int (&av)[2]) {
            const bool ady = t >= t0;
            const int tt = ady ? t - t0 : t, count = (ady ? n1 : n0) - tt * kTileLeaves;
            const unsigned short *li = list + (ady ? kMaxChunk : 0) + tt * kTileLeaves;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int lf = (lane >> 3) + 8 * u;
                const int rr = base + li[lf < count ? lf : 0];
                v[u] = *reinterpret_cast<const float4 *>(tr.hidden + ((size_t)rr * tr.n + tr.ids[2 * (size_t)rr + 1]) * tr.hs + 4 * (lane & 7));
                av[u] = tr.last_action[rr];
            }
        };
        float xin[10];
        float4 xrow[2];
        int xact[2];
        if (wave < t0 + t1) {
            if (ROWS) fetch_rows(wave, xrow, xact);
            else fetch(wave, xin);
        }
        for (int t = wave; t < t0 + t1; t += W) {
            const bool ady = t >= t0;                                        // wave-uniform
            const int tt = ady ? t - t0 : t, count = (ady ? n1 : n0) - tt * kTileLeaves;   // leaves in this tile (>= 1; may exceed 16)
            const unsigned short *li = list + (ady ? kMaxChunk : 0) + tt * kTileLeaves;
            const bool mine = j < count;
            const int row = base + li[mine ? j : 0];                         // (a ragged tile repeats its first row)
            float *hrow = ROWS ? nullptr : hidden_out + (size_t)row * S;
            float *hrow4[2] = {nullptr, nullptr};                            // ROWS: where the lane's two 16-byte pieces of the new rows go
            if (ROWS) {
                const int q = lane & 7;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int lf = (lane >> 3) + 8 * u;
                    float4 v = xrow[u];
                    if (q == 7) v.w = xact[u] == 0 ? 1.f : 0.f;              // input S = the first action's one-hot slot
                    *reinterpret_cast<float2 *>(tile + ((2 * q) * kTileLeaves + lf) * 2) = make_float2(v.x, v.y);
                    *reinterpret_cast<float2 *>(tile + ((2 * q + 1) * kTileLeaves + lf) * 2) = make_float2(v.z, v.w);
                    if (q == 7) {                                            // inputs S + 1 .. 39: the other actions, then zeros
#pragma unroll
                        for (int pr = 16; pr < 20; pr++) {
                            const int a0 = 2 * pr - S, a1 = a0 + 1;
                            *reinterpret_cast<float2 *>(tile + (pr * kTileLeaves + lf) * 2) =
                                make_float2(a0 < A && xact[u] == a0 ? 1.f : 0.f, a1 < A && xact[u] == a1 ? 1.f : 0.f);
                        }
                    }
                    if (lf < count) {
                        const int rr = base + li[lf];
                        hrow4[u] = tr.hidden + ((size_t)rr * tr.n + tr.ids[2 * (size_t)rr]) * tr.hs + 4 * q;
                    }
                }
                if (t + W < t0 + t1) fetch_rows(t + W, xrow, xact);   // (in flight during this tile's layers)
            } else {
#pragma unroll
                for (int u = 0; u < 10; u++) {
                    const int i = lane + kWave * u, lf = i / 40, k = i % 40;
                    tile[((k >> 1) * kTileLeaves + lf) * 2 + (k & 1)] = xin[u];
                }
                if (t + W < t0 + t1) fetch(t + W, xin);      // (in flight during this tile's layers)
            }
            lds_sync();
            v4f y[4];
            const MatOff m_in = pick<CP>(dl, !ady, M_DYN_IN, M_ADY_IN), m_out = pick<CP>(dl, !ady, M_DYN_OUT, M_ADY_OUT);
            const MatOff p_in = pick<CP>(dl, !ady, M_PRE_IN, M_APR_IN), p_out = pick<CP>(dl, !ady, M_PRE_OUT, M_APR_OUT);
            tile_layer<5>(lds + m_in.w, m_in.op, lds + m_in.b, tile, lane, y);
            lds_sync();
            store_trunk(tile, y, lane);
            lds_sync();
            // (afterstate dynamics: S = 31 outputs, two tiles; dynamics: reward + state, 62 outputs, all four)
            if (CP && ady) tile_layer<8, 2>(lds + m_out.w, m_out.op, lds + m_out.b, tile, lane, y);
            else tile_layer<8>(lds + m_out.w, m_out.op, lds + m_out.b, tile, lane, y);
            // dynamics: [reward logits 0..S-1 | next state S..2S-1]; afterstate dynamics: next state 0..S-1
            float reward = 0.f;
            {
                const int lo = ady ? 0 : S;
                float mr = -__builtin_inff(), mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t4 + 4 * g + r;
                        if (!ady && o < S) mr = fmaxf(mr, y[t4][r]);
                        if (o >= lo && o < lo + S) { mn = fminf(mn, y[t4][r]); mx = fmaxf(mx, y[t4][r]); }
                    }
                mn = tile_min(mn); mx = tile_max(mx);
                if (!ady) {
                    mr = tile_max(mr);
                    float de[4][4], nu[4][4];
#pragma unroll
                    for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int o = 16 * t4 + 4 * g + r;
                            const float ev = o < S ? smz_exp(y[t4][r] - mr) : 0.f;
                            de[t4][r] = ev;
                            nu[t4][r] = o < S ? 0.f + (float)(o - half) * ev : 0.f;
                        }
                    const float den = tile_sum(de), num = tile_sum(nu);
                    reward = support_to_scalar(num, den);
                }
                float sc = mx - mn;
                if (sc < 1e-5f) sc += 1e-5f;
                // the new hidden state: prediction input (tile inputs 0..S-1; input S: zero) and, for the tile's leaves, out
                lds_sync();
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t4 + 4 * g + r, k = o - lo;
                        if (k >= 0 && k < 32) {
                            const float hv = k < S ? __fdividef(y[t4][r] - mn, sc) : 0.f;
                            tile[((k >> 1) * kTileLeaves + j) * 2 + (k & 1)] = hv;
                            if (!ROWS && k < S && mine) hrow[k] = hv;
                        }
                    }
            }
            lds_sync();
            if (ROWS) {                                                      // the new rows leave through the tile: 16 bytes per lane and row
                const int q = lane & 7;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int lf = (lane >> 3) + 8 * u;
                    const float2 lo = *reinterpret_cast<const float2 *>(tile + ((2 * q) * kTileLeaves + lf) * 2);
                    const float2 hi = *reinterpret_cast<const float2 *>(tile + ((2 * q + 1) * kTileLeaves + lf) * 2);
                    if (hrow4[u]) *reinterpret_cast<float4 *>(hrow4[u]) = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            tile_layer<4>(lds + p_in.w, p_in.op, lds + p_in.b, tile, lane, y);
            lds_sync();
            store_trunk(tile, y, lane);
            lds_sync();
            // (A + S <= 35 outputs: three tiles)
            tile_layer<8, (CP && A + kFastS <= 48) ? 3 : 4>(lds + p_out.w, p_out.op, lds + p_out.b, tile, lane, y);
            {   // [policy logits 0..A-1 | value logits A..A+S-1]
                float mp = -__builtin_inff(), mv = -__builtin_inff();
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t4 + 4 * g + r;
                        if (o < A) mp = fmaxf(mp, y[t4][r]);
                        else if (o < A + S) mv = fmaxf(mv, y[t4][r]);
                    }
                mp = tile_max(mp); mv = tile_max(mv);
                float ep[4][4], ev[4][4], nv[4][4];
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t4 + 4 * g + r;
                        const bool pol = o < A, val = !pol && o < A + S;
                        const float e = (pol || val) ? smz_exp(y[t4][r] - (pol ? mp : mv)) : 0.f;
                        ep[t4][r] = pol ? e : 0.f;
                        ev[t4][r] = val ? e : 0.f;
                        nv[t4][r] = val ? 0.f + (float)(o - A - half) * e : 0.f;
                    }
                const float dp = tile_sum(ep), dv = tile_sum(ev), nvs = tile_sum(nv);
                const float value = support_to_scalar(nvs, dv);
                if (mine && g == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) if (r < A) policy_out[(size_t)row * A + r] = __fdividef(ep[0][r], dp);
                    value_out[row] = value;
                    if (reward_out) reward_out[row] = reward;
                }
            }
            lds_sync();
        }
        __syncthreads();                                                     // the lists are refilled for the next chunk
    }
}

// -------------------------------------------------------------------------------------------------------------------
// Networks too wide for LDS residency (the reference's config/experiment_434_config.json: state_space_dimensions 61,
// hidden_layer_dimensions 126, and its checkpoint 450 with number_of_hidden_layer 4; any shape with H <= 128, 2 S <= 128,
// A + S <= 128 -- hidden layers are the same Linear(H, H) applied L times, neural_network_mlp_model.py:122-142): the same
// 16-leaf tiles with the weights streamed from L2 -- A operands as 8-byte global loads of a 128-wide packed image
// (smz_mlp_layout_wide), prefetched one input group ahead of the MFMAs that consume them; layers are 8 tiles of 16
// neurons, dimensions are run-time values.  Outputs agree with the torch-GEMM heads / the reference's tapes within the
// measured float tolerances (no bit-identity partner exists for these shapes).
constexpr int kWideOP = 128, kWideTiles = 8, kWideWaves = 8;
constexpr int kWideTileFloats = 64 * 32;              // [64 input pairs][16 leaves][2]

__device__ inline void wide_layer(const float *__restrict__ W, const float *__restrict__ bias, const float *xp, int K8, int lane, v4f (&y)[kWideTiles]) {
    const int g = lane >> 4, j = lane & 15;
    v4f o[kWideTiles];                                      // y = the even-input accumulators (from the bias), o = the odd ones
    float2 wa[3][kWideTiles];                               // weights of three consecutive input groups: two in flight ahead
    const float *Wl = W + ((size_t)(g >> 1) * kWideOP + j) * 4 + 2 * (g & 1);
    auto wload = [&](int slot, int c) {
#pragma unroll
        for (int t = 0; t < kWideTiles; t++) wa[slot][t] = *reinterpret_cast<const float2 *>(Wl + ((size_t)2 * c * kWideOP + 16 * t) * 4);
    };
#pragma unroll
    for (int t = 0; t < kWideTiles; t++) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * g);
        y[t] = v4f{b.x, b.y, b.z, b.w};
        o[t] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    wload(0, 0);
    if (K8 > 1) wload(1, 1);
    for (int c = 0; c < K8; c += 3) {                       // three input groups per trip: static indices into wa[]
#pragma unroll
        for (int h = 0; h < 3; h++) {
            const int cc = c + h;
            if (cc < K8) {                                  // wave-uniform
                const float2 xb = *reinterpret_cast<const float2 *>(xp + ((4 * cc + g) * kTileLeaves + j) * 2);
                if (cc + 2 < K8) wload((h + 2) % 3, cc + 2);
#pragma unroll
                for (int t = 0; t < kWideTiles; t++) {
                    y[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[h][t].x, xb.x, y[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[h][t].y, xb.y, o[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < kWideTiles; t++) y[t] = y[t] + o[t];
}
__device__ inline float wide_sum(float s) { s = s + lane_xor16(s); return s + lane_xor32(s); }

__global__ void __launch_bounds__(kWideWaves *kWave) k_mlp_recurrent_wide(smz_mlp_desc d, const float *__restrict__ weights,
                                                                          const float *__restrict__ x, const uint8_t *__restrict__ branch,
                                                                          float *__restrict__ hidden_out, float *__restrict__ reward_out,
                                                                          float *__restrict__ policy_out, float *__restrict__ value_out,
                                                                          int B, int chunk) {
    float *lds = reinterpret_cast<float *>(smz_mlp_lds4);
    const int S = d.S, A = d.A, H = d.H, half = S / 2, XW = S + A;
    const int K8x = (XW + 7) >> 3, K8h = (H + 7) >> 3, K8s = (S + 7) >> 3;
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int g = lane >> 4, j = lane & 15;
    const int waves = blockDim.x / kWave;                 // 2 .. kWideWaves: small batches spread their tiles over more workgroups
    float *tile = lds + wave * kWideTileFloats;
    unsigned short *list = reinterpret_cast<unsigned short *>(lds + waves * kWideTileFloats);   // [2][kMaxChunk]
    int *cnt = reinterpret_cast<int *>(list + 2 * kMaxChunk);
    for (int base = blockIdx.x * chunk; base < B; base += gridDim.x * chunk) {
        const int n = B - base < chunk ? B - base : chunk;
        if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int w = branch[base + i] != 0 ? 0 : 1;
            list[w * kMaxChunk + atomicAdd(&cnt[w], 1)] = (unsigned short)i;
        }
        __syncthreads();
        const int n0 = cnt[0], n1 = cnt[1], t0 = (n0 + kTileLeaves - 1) / kTileLeaves, t1 = (n1 + kTileLeaves - 1) / kTileLeaves;
        for (int tl = wave; tl < t0 + t1; tl += waves) {
            const bool ady = tl >= t0;                                       // wave-uniform
            const int tt = ady ? tl - t0 : tl, count = (ady ? n1 : n0) - tt * kTileLeaves;
            const unsigned short *li = list + (ady ? kMaxChunk : 0) + tt * kTileLeaves;
            const bool mine = j < count;
            const int row = base + li[mine ? j : 0];
            // network inputs [hidden | one-hot] -> tile, zero beyond them up to the layer's 8-input groups
            for (int i = lane; i < kTileLeaves * 8 * K8x; i += kWave) {
                const int lf = i / (8 * K8x), k = i % (8 * K8x);
                const int rr = base + li[lf < count ? lf : 0];
                tile[((k >> 1) * kTileLeaves + lf) * 2 + (k & 1)] = k < XW ? x[(size_t)rr * XW + k] : 0.f;
            }
            lds_sync();
            // (selects between constant-index descriptor entries: a run-time index would put the table in scratch)
            const int w_in = ady ? d.off[M_ADY_IN] : d.off[M_DYN_IN], b_in = ady ? d.off[M_COUNT + M_ADY_IN] : d.off[M_COUNT + M_DYN_IN];
            const int w_out = ady ? d.off[M_ADY_OUT] : d.off[M_DYN_OUT], b_out = ady ? d.off[M_COUNT + M_ADY_OUT] : d.off[M_COUNT + M_DYN_OUT];
            const int wp_in = ady ? d.off[M_APR_IN] : d.off[M_PRE_IN], bp_in = ady ? d.off[M_COUNT + M_APR_IN] : d.off[M_COUNT + M_PRE_IN];
            const int wp_out = ady ? d.off[M_APR_OUT] : d.off[M_PRE_OUT], bp_out = ady ? d.off[M_COUNT + M_APR_OUT] : d.off[M_COUNT + M_PRE_OUT];
            const int w_mid = ady ? d.off[M_ADY_MID] : d.off[M_DYN_MID], b_mid = ady ? d.off[M_COUNT + M_ADY_MID] : d.off[M_COUNT + M_DYN_MID];
            const int wp_mid = ady ? d.off[M_APR_MID] : d.off[M_PRE_MID], bp_mid = ady ? d.off[M_COUNT + M_APR_MID] : d.off[M_COUNT + M_PRE_MID];
            v4f y[kWideTiles];
            auto trunk_store = [&]() {
#pragma unroll
                for (int t = 0; t < kWideTiles; t++) {
                    const int nn = 16 * t + 4 * g;
                    *reinterpret_cast<float2 *>(tile + (((nn >> 1) + 0) * kTileLeaves + j) * 2) = make_float2(elu(y[t][0]), elu(y[t][1]));
                    *reinterpret_cast<float2 *>(tile + (((nn >> 1) + 1) * kTileLeaves + j) * 2) = make_float2(elu(y[t][2]), elu(y[t][3]));
                }
            };
            wide_layer(weights + w_in, weights + b_in, tile, K8x, lane, y);
            lds_sync();
            trunk_store();
            lds_sync();
            for (int l = 0; l < d.L; l++) {          // the SAME Linear(H, H) + ELU applied L times (neural_network_mlp_model.py:122-142)
                wide_layer(weights + w_mid, weights + b_mid, tile, K8h, lane, y);
                lds_sync();
                trunk_store();
                lds_sync();
            }
            wide_layer(weights + w_out, weights + b_out, tile, K8h, lane, y);
            float reward = 0.f;
            {   // dynamics: [reward logits 0..S-1 | next state S..2S-1]; afterstate dynamics: next state 0..S-1
                const int lo = ady ? 0 : S;
                float mr = -__builtin_inff(), mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
                for (int t = 0; t < kWideTiles; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t + 4 * g + r;
                        if (!ady && o < S) mr = fmaxf(mr, y[t][r]);
                        if (o >= lo && o < lo + S) { mn = fminf(mn, y[t][r]); mx = fmaxf(mx, y[t][r]); }
                    }
                mn = tile_min(mn); mx = tile_max(mx);
                if (!ady) {
                    mr = tile_max(mr);
                    float den = 0.f, num = 0.f;
#pragma unroll
                    for (int t = 0; t < kWideTiles; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int o = 16 * t + 4 * g + r;
                            if (o < S) { const float ev = smz_exp(y[t][r] - mr); den += ev; num += (float)(o - half) * ev; }
                        }
                    reward = support_to_scalar(wide_sum(num), wide_sum(den));
                }
                float sc = mx - mn;
                if (sc < 1e-5f) sc += 1e-5f;
                lds_sync();
#pragma unroll
                for (int t = 0; t < kWideTiles; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t + 4 * g + r, k = o - lo;
                        if (k >= 0 && k < S) {
                            const float hv = __fdividef(y[t][r] - mn, sc);
                            tile[((k >> 1) * kTileLeaves + j) * 2 + (k & 1)] = hv;
                            if (mine) hidden_out[(size_t)row * S + k] = hv;
                        }
                    }
                for (int i = lane; i < kTileLeaves * (8 * K8s - S); i += kWave) {          // zero inputs S .. 8 K8s - 1
                    const int lf = i / (8 * K8s - S), k = S + i % (8 * K8s - S);
                    tile[((k >> 1) * kTileLeaves + lf) * 2 + (k & 1)] = 0.f;
                }
            }
            lds_sync();
            wide_layer(weights + wp_in, weights + bp_in, tile, K8s, lane, y);
            lds_sync();
            trunk_store();
            lds_sync();
            for (int l = 0; l < d.L; l++) {
                wide_layer(weights + wp_mid, weights + bp_mid, tile, K8h, lane, y);
                lds_sync();
                trunk_store();
                lds_sync();
            }
            wide_layer(weights + wp_out, weights + bp_out, tile, K8h, lane, y);
            {   // [policy logits 0..A-1 | value logits A..A+S-1]
                float mp = -__builtin_inff(), mv = -__builtin_inff();
#pragma unroll
                for (int t = 0; t < kWideTiles; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t + 4 * g + r;
                        if (o < A) mp = fmaxf(mp, y[t][r]);
                        else if (o < A + S) mv = fmaxf(mv, y[t][r]);
                    }
                mp = tile_max(mp); mv = tile_max(mv);
                float dp = 0.f, dv = 0.f, nv = 0.f;
#pragma unroll
                for (int t = 0; t < kWideTiles; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = 16 * t + 4 * g + r;
                        if (o < A) { y[t][r] = smz_exp(y[t][r] - mp); dp += y[t][r]; }
                        else if (o < A + S) { const float ev = smz_exp(y[t][r] - mv); dv += ev; nv += (float)(o - A - half) * ev; }
                    }
                dp = wide_sum(dp);
                const float value = support_to_scalar(wide_sum(nv), wide_sum(dv));
                if (mine) {
#pragma unroll
                    for (int t = 0; t < kWideTiles; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int o = 16 * t + 4 * g + r;
                            if (o < A) policy_out[(size_t)row * A + o] = __fdividef(y[t][r], dp);
                        }
                    if (g == 0) {
                        value_out[row] = value;
                        if (reward_out) reward_out[row] = reward;
                    }
                }
            }
            lds_sync();
        }
        __syncthreads();
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
    if (d->S == kFastS && d->H == kFastH && d->L == kFastL && (d->A == 2 || d->A == 4)) {
        // large batches: 16-leaf tiles on the matrix cores (bit-identical; SMZ_MLP_MFMA_MIN = smallest batch that takes it)
        int min_rows = 8192;      // (measured: 242 vs 217 M simulations/s at 8192 trees, 367 vs 278 at 12 288, 132 vs 144 at 4096)
        if (const char *e = getenv("SMZ_MLP_MFMA_MIN")) min_rows = atoi(e);
        const size_t lds2 = ((size_t)(kMfmaCompact ? compact_total_floats(*d) : d->total_floats - rep_floats(*d)) + (size_t)kMfmaWaves * kTileFloats) * sizeof(float) +
                            2 * kMaxChunk * sizeof(unsigned short) + 16;
        if (min_rows == 0 && lds2 > (size_t)kLdsBytes) return SMZ_ERR_TOO_LARGE;     // (forced: say so instead of falling back)
        if (min_rows >= 0 && B >= min_rows && lds2 <= (size_t)kLdsBytes) {
            int chunk = mfma_chunk(B);
            int wgs = (B + chunk - 1) / chunk;
            if (wgs > 256) wgs = 256;
            if (d->A == 2) {
                if (allow_lds(k_mlp_recurrent_mfma<2, false>, lds2) != SMZ_OK) return SMZ_ERR_HIP;
                hipLaunchKernelGGL((k_mlp_recurrent_mfma<2, false>), dim3(wgs), dim3(kMfmaWaves * kWave), lds2, (hipStream_t)stream, *d, weights_dev,
                                   mlp_input_dev, branch_dev, hidden_out_dev, reward_out_dev, policy_out_dev, value_out_dev, B, chunk, TreeRows{});
            } else {
                if (allow_lds(k_mlp_recurrent_mfma<4, false>, lds2) != SMZ_OK) return SMZ_ERR_HIP;
                hipLaunchKernelGGL((k_mlp_recurrent_mfma<4, false>), dim3(wgs), dim3(kMfmaWaves * kWave), lds2, (hipStream_t)stream, *d, weights_dev,
                                   mlp_input_dev, branch_dev, hidden_out_dev, reward_out_dev, policy_out_dev, value_out_dev, B, chunk, TreeRows{});
            }
            return hipGetLastError() == hipSuccess ? SMZ_OK : SMZ_ERR_HIP;
        }
    }
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

int smz_mlp_recurrent_rows(const smz_mlp_desc *d, const float *weights_dev, float *hidden_dev, int nodes_per_tree,
                           int row_stride, const int32_t *ids_dev, const int32_t *last_action_dev, const uint8_t *branch_dev,
                           float *reward_out_dev, float *policy_out_dev, float *value_out_dev, int B, smz_stream stream) {
    if (mlp_check(d, weights_dev) != SMZ_OK || !hidden_dev || !ids_dev || !last_action_dev || !branch_dev || !policy_out_dev ||
        !value_out_dev || B < 1 || nodes_per_tree < 1 || row_stride < d->S)
        return SMZ_ERR_INVALID;
    const int waves = mfma_launch_waves();
    const size_t lds2 = ((size_t)(kMfmaCompact ? compact_total_floats(*d) : d->total_floats - rep_floats(*d)) + (size_t)waves * kTileFloats) * sizeof(float) +
                        2 * kMaxChunk * sizeof(unsigned short) + 16;
    if (!(d->S == kFastS && d->H == kFastH && d->L == kFastL && (d->A == 2 || d->A == 4)) || lds2 > (size_t)kLdsBytes)
        return SMZ_ERR_TOO_LARGE;
    int chunk = mfma_chunk(B);
    int wgs = (B + chunk - 1) / chunk;
    if (wgs > 256) wgs = 256;
    const TreeRows tr = {hidden_dev, ids_dev, last_action_dev, nodes_per_tree, row_stride};
    if (d->A == 2) {
        if (allow_lds(k_mlp_recurrent_mfma<2, true>, lds2) != SMZ_OK) return SMZ_ERR_HIP;
        hipLaunchKernelGGL((k_mlp_recurrent_mfma<2, true>), dim3(wgs), dim3(waves * kWave), lds2, (hipStream_t)stream, *d, weights_dev,
                           nullptr, branch_dev, nullptr, reward_out_dev, policy_out_dev, value_out_dev, B, chunk, tr);
    } else {
        if (allow_lds(k_mlp_recurrent_mfma<4, true>, lds2) != SMZ_OK) return SMZ_ERR_HIP;
        hipLaunchKernelGGL((k_mlp_recurrent_mfma<4, true>), dim3(wgs), dim3(waves * kWave), lds2, (hipStream_t)stream, *d, weights_dev,
                           nullptr, branch_dev, nullptr, reward_out_dev, policy_out_dev, value_out_dev, B, chunk, tr);
    }
    return hipGetLastError() == hipSuccess ? SMZ_OK : SMZ_ERR_HIP;
}

int smz_mlp_layout_wide(smz_mlp_desc *d) {
    if (!d || d->obs < 1 || d->A < 1 || d->S < 1 || d->H < 1 || d->L < 0) return SMZ_ERR_INVALID;
    if (d->H > kWideOP || 2 * d->S > kWideOP || d->A + d->S > kWideOP) return SMZ_ERR_TOO_LARGE;
    d->OP = kWideOP;
    const int mid = d->L > 0 ? d->H : 0;     // one shared Linear(H, H) per trunk, applied L times
    const int K[M_COUNT] = {d->S + d->A, d->S + d->A, mid, mid, d->H, d->H, d->S, d->S, mid, mid, d->H, d->H, d->obs, mid, d->H};
    int off = 0;
    // every matrix padded to a multiple of 8 input rows: the tile kernel consumes inputs in groups of eight
    for (int m = 0; m < M_COUNT; m++) { d->off[m] = off; off += ((K[m] + 7) & ~7) * d->OP; }
    for (int m = 0; m < M_COUNT; m++) { d->off[M_COUNT + m] = off; off += d->OP; }
    d->total_floats = off;
    return SMZ_OK;
}

int smz_mlp_recurrent_wide(const smz_mlp_desc *d, const float *weights_dev, const float *mlp_input_dev,
                           const uint8_t *branch_dev, float *hidden_out_dev, float *reward_out_dev, float *policy_out_dev,
                           float *value_out_dev, int B, smz_stream stream) {
    if (!d || !weights_dev || !mlp_input_dev || !branch_dev || !hidden_out_dev || !policy_out_dev || !value_out_dev || B < 1)
        return SMZ_ERR_INVALID;
    smz_mlp_desc t = *d;
    if (smz_mlp_layout_wide(&t) != SMZ_OK || t.total_floats != d->total_floats || d->OP != kWideOP) return SMZ_ERR_INVALID;
    // geometry: a tile is ~9 us of matrix-pipe time whatever the batch, so small batches want their tiles on as many SIMDs as
    // possible -- chunks of 32 rows (one or two tiles per branch) and 4 waves per workgroup up to 8 k rows, then larger
    // chunks (fuller tiles) and 8 waves
    int chunk, waves;
    if (B <= 8192) { chunk = 32; waves = 4; }
    else {
        chunk = ((B + 255) / 256 + 127) / 128 * 128;
        if (chunk > kMaxChunk) chunk = kMaxChunk;
        waves = kWideWaves;
    }
    const size_t lds = (size_t)waves * kWideTileFloats * sizeof(float) + 2 * kMaxChunk * sizeof(unsigned short) + 16;
    int wgs = (B + chunk - 1) / chunk;
    if (wgs > 2048) wgs = 2048;
    if (allow_lds(k_mlp_recurrent_wide, lds) != SMZ_OK) return SMZ_ERR_HIP;
    hipLaunchKernelGGL(k_mlp_recurrent_wide, dim3(wgs), dim3(waves * kWave), lds, (hipStream_t)stream, *d, weights_dev,
                       mlp_input_dev, branch_dev, hidden_out_dev, reward_out_dev, policy_out_dev, value_out_dev, B, chunk);
    return hipGetLastError() == hipSuccess ? SMZ_OK : SMZ_ERR_HIP;
}

}  // extern "C"
