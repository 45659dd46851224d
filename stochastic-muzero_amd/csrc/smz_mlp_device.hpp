// smz_mlp_device.hpp -- device functions of the fused `mlp_model` heads (shared by the stand-alone head kernels in
// smz_mlp.hip and by the whole-search kernel in smz_kernels.hip, so that both produce bit-identical network outputs).
//
// Weights live in LDS in an input-major, 4-way interleaved layout (include/smz.h): lane o owns output neuron o and
// fetches four consecutive input weights with one 16-byte LDS read, activations are broadcast 16-byte reads.  All
// multiply-adds are explicit fmaf(), so the result does not depend on the translation unit's -ffp-contract setting.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/smz.h"

namespace smz_mlp {

constexpr int kWave = 64;
enum { M_DYN_IN, M_ADY_IN, M_DYN_MID, M_ADY_MID, M_DYN_OUT, M_ADY_OUT, M_PRE_IN, M_APR_IN, M_PRE_MID, M_APR_MID,
       M_PRE_OUT, M_APR_OUT, M_REP_IN, M_REP_MID, M_REP_OUT, M_COUNT };

__host__ __device__ inline int up4(int x) { return (x + 3) & ~3; }

// acc[r][u] = bias_r[o] + sum_k W_r[k][o] * act_r[k]  for o = lane + 64 u, for R rows at once (each row may use a
// different matrix: its branch's).  Even and odd inputs accumulate in the two halves of a packed register
// (v_pk_fma_f32: two FMAs per instruction) and are added at the end; the order is fixed, so every caller rounds
// identically.  Four 4-wide steps per iteration with all their 16-byte LDS reads issued back to back: one LDS latency
// covers 16 inputs (the plain loop is a chain of exposed LDS round trips).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline v2f pk_fma(float wx, float wy, float ax, float ay, v2f c) {
    v2f w = {wx, wy}, a = {ax, ay};
    return __builtin_elementwise_fma(w, a, c);
}
// SAME: every row uses row 0's matrix -- the weights are read from LDS once and applied to all R activation vectors
// OP[r]: floats / 4 between consecutive 4-input groups of row r's matrix = its row width (64 in the packed buffer; the
// compact LDS image of the search kernel stores narrow output layers narrower: see mat_op)
template <int U, int R, bool SAME = false>
__device__ inline void dense(const float *const (&W)[R], const float *const (&bias)[R], const float *const (&act)[R], int K4,
                             const int (&OP)[R], int lane, float (&acc)[R][U]) {
    const float4 *a4[R], *w4[R];
    v2f acc2[R][U];
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int u = 0; u < U; u++) { acc2[r][u].x = bias[r][lane + kWave * u]; acc2[r][u].y = 0.f; }
        a4[r] = reinterpret_cast<const float4 *>(act[r]);
        w4[r] = reinterpret_cast<const float4 *>(W[r]) + lane;
    }
    const int n = K4 >> 2;
    int q = 0;
    for (; q + 4 <= n; q += 4) {
        float4 a[R][4], w[R][U][4];
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                a[r][j] = a4[r][q + j];
#pragma unroll
                for (int u = 0; u < U; u++) w[r][u][j] = (SAME && r > 0) ? w[0][u][j] : w4[r][(size_t)(q + j) * OP[r] + kWave * u];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    v2f t = acc2[r][u];
                    t = pk_fma(w[r][u][j].x, w[r][u][j].y, a[r][j].x, a[r][j].y, t);
                    t = pk_fma(w[r][u][j].z, w[r][u][j].w, a[r][j].z, a[r][j].w, t);
                    acc2[r][u] = t;
                }
            }
        }
    }
    for (; q < n; q++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float4 av = a4[r][q];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const float4 wv = w4[SAME ? 0 : r][(size_t)q * OP[SAME ? 0 : r] + kWave * u];
                v2f t = acc2[r][u];
                t = pk_fma(wv.x, wv.y, av.x, av.y, t);
                t = pk_fma(wv.z, wv.w, av.z, av.w, t);
                acc2[r][u] = t;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int u = 0; u < U; u++) acc[r][u] = acc2[r][u].x + acc2[r][u].y;
}

template <int U, int R, bool SAME = false>
__device__ inline void dense(const float *const (&W)[R], const float *const (&bias)[R], const float *const (&act)[R], int K4,
                             int OP, int lane, float (&acc)[R][U]) {
    int op[R];
#pragma unroll
    for (int r = 0; r < R; r++) op[r] = OP;
    dense<U, R, SAME>(W, bias, act, K4, op, lane, acc);
}

// A layer with at most 32 outputs for TWO rows at once on lane halves: lane l computes output l & 31 of row l >> 5 (row 0 reads
// act0, row 1 act1; both use matrix W).  The chain of every output is dense()'s (even / odd inputs in the halves of a packed
// register, four 4-wide steps per iteration, added at the end): bit-identical, in half the multiply-add instructions.
__device__ inline float dense_halves(const float *W, const float *bias, const float *act0, const float *act1, int K4, int OP, int lane) {
    const int o = lane & 31;
    const float4 *a4 = reinterpret_cast<const float4 *>(lane < 32 ? act0 : act1);
    const float4 *w4 = reinterpret_cast<const float4 *>(W) + o;
    v2f acc2 = {bias[o], 0.f};
    const int n = K4 >> 2;
    int q = 0;
    for (; q + 4 <= n; q += 4) {
        float4 a[4], w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { a[j] = a4[q + j]; w[j] = w4[(size_t)(q + j) * OP]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc2 = pk_fma(w[j].x, w[j].y, a[j].x, a[j].y, acc2);
            acc2 = pk_fma(w[j].z, w[j].w, a[j].z, a[j].w, acc2);
        }
    }
    for (; q < n; q++) {
        const float4 av = a4[q], wv = w4[(size_t)q * OP];
        acc2 = pk_fma(wv.x, wv.y, av.x, av.y, acc2);
        acc2 = pk_fma(wv.z, wv.w, av.z, av.w, acc2);
    }
    return acc2.x + acc2.y;
}

// exp for the heads: the hardware exponential (v_exp_f32 on x log2 e, ~1 ulp) -- the network outputs are held to the
// 1e-5 class against the reference's torch-CPU numbers, not to bit parity, and the library expf costs a dozen more
// instructions per call on the leaf-evaluation path (ELU of every hidden unit, three softmaxes per leaf)
__device__ inline float smz_exp(float x) { return __expf(x); }
// torch's ELU evaluates exp(x) - 1 (aten/src/ATen/native/cpu/Activation.cpp elu_kernel)
__device__ inline float elu(float x) { return x > 0.f ? x : smz_exp(x) - 1.0f; }
// Wave-wide reductions on the DPP crossbar (VALU latency, no LDS round trips), written as fused v_<op>_f32_dpp
// instructions: row_shr 1/2/4/8 build the totals of each 16-lane row in lanes 15/31/47/63 (a lane whose source lies
// outside its row keeps its value: bound_ctrl off), row_bcast:15 / row_bcast:31 carry them across rows, lane 63 holds
// the result and is broadcast through an SGPR.  hipcc lowers the same chain written with __builtin_amdgcn_update_dpp
// to 5 instructions per step (identity mov + hazard nop + v_mov_dpp + NaN canonicalisation + op), hence the asm;
// `s_nop 1` covers the 2 wait states a DPP read needs after the VALU write of the same register.  All 64 lanes must be
// active (the callers are wave-uniform).
// SMZ_DPP_VOLATILE: `volatile` keeps the reduction chains of a pass in program order (two rows' tails one after the other);
// without it (-DSMZ_DPP_VOLATILE= builds) the scheduler may interleave independent chains.  (A/B: profiles/r05_d_*.)
#ifndef SMZ_DPP_VOLATILE
#define SMZ_DPP_VOLATILE volatile
#endif
// SMZ_PAIR_TAILS (round 5): the tails of a two-row pass are evaluated for both rows at once (softmax_decode_pair and friends
// below: bit-identical to the per-row tails).  -DSMZ_PAIR_TAILS=0 builds keep the per-row tails.
#ifndef SMZ_PAIR_TAILS
#define SMZ_PAIR_TAILS 1
#endif
// SMZ_ONE_DECODE (round 5): the reward's and the value's support decodes of a paired pass run as ONE instruction sequence
#ifndef SMZ_ONE_DECODE
#define SMZ_ONE_DECODE 1
#endif
// SMZ_DENSE_HALVES (round 5): a pair of afterstate rows runs its narrow dynamics layer on lane halves (dense_halves)
#ifndef SMZ_DENSE_HALVES
#define SMZ_DENSE_HALVES 1
#endif
#define SMZ_DPP_REDUCE(NAME, INSN)                                                               \
    __device__ inline float NAME(float v) {                                                      \
        asm SMZ_DPP_VOLATILE("s_nop 1\n\t"                                                               \
                     INSN " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      \
                     INSN " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      \
                     INSN " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      \
                     INSN " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      \
                     INSN " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"   \
                     INSN " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"        \
                     : "+v"(v));                                                                 \
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));                 \
    }
SMZ_DPP_REDUCE(wave_max, "v_max_f32_dpp")
SMZ_DPP_REDUCE(wave_min, "v_min_f32_dpp")
SMZ_DPP_REDUCE(wave_sum, "v_add_f32_dpp")
#undef SMZ_DPP_REDUCE
__device__ inline float op_max(float a, float b) { return fmaxf(a, b); }
__device__ inline float op_min(float a, float b) { return fminf(a, b); }
// two independent sums in one chain: each instruction fills one of the other's DPP wait states
__device__ inline void wave_sum2(float &a, float &b) {
    asm SMZ_DPP_VOLATILE("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(a), "+v"(b));
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}
__device__ inline void wave_minmax(float &mn, float &mx) {
    asm SMZ_DPP_VOLATILE("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(mn), "+v"(mx));
    mn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mn), 63));
    mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 63));
}

// three independent chains per step: no wait states left to pad
#define SMZ_DPP3_STEP(I0, I1, I2, CTRL)                                                             \
    I0 " %0, %0, %0 " CTRL "\n\t" I1 " %1, %1, %1 " CTRL "\n\t" I2 " %2, %2, %2 " CTRL "\n\t"
#define SMZ_DPP3_REDUCE(NAME, I0, I1, I2)                                                          \
    __device__ inline void NAME(float &a, float &b, float &c) {                                    \
        asm SMZ_DPP_VOLATILE("s_nop 1\n\t"                                                                 \
                     SMZ_DPP3_STEP(I0, I1, I2, "row_shr:1 row_mask:0xf bank_mask:0xf")             \
                     SMZ_DPP3_STEP(I0, I1, I2, "row_shr:2 row_mask:0xf bank_mask:0xf")             \
                     SMZ_DPP3_STEP(I0, I1, I2, "row_shr:4 row_mask:0xf bank_mask:0xf")             \
                     SMZ_DPP3_STEP(I0, I1, I2, "row_shr:8 row_mask:0xf bank_mask:0xf")             \
                     SMZ_DPP3_STEP(I0, I1, I2, "row_bcast:15 row_mask:0xa bank_mask:0xf")          \
                     SMZ_DPP3_STEP(I0, I1, I2, "row_bcast:31 row_mask:0xc bank_mask:0xf")          \
                     "s_nop 1"                                                                     \
                     : "+v"(a), "+v"(b), "+v"(c));                                                 \
        a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));                      \
        b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));                      \
        c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));                      \
    }
SMZ_DPP3_REDUCE(wave_sum3, "v_add_f32_dpp", "v_add_f32_dpp", "v_add_f32_dpp")
SMZ_DPP3_REDUCE(wave_max_min_max, "v_max_f32_dpp", "v_min_f32_dpp", "v_max_f32_dpp")
#undef SMZ_DPP3_REDUCE
#undef SMZ_DPP3_STEP
__device__ inline void wave_max2(float &a, float &b) {
    asm SMZ_DPP_VOLATILE("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(a), "+v"(b));
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}

// Orders this wave's LDS traffic: LDS instructions of one wave execute in issue order, so other lanes' earlier writes are
// visible once they have been issued; the asm is a compiler barrier plus an LDS-counter wait.  (A scoped fence here
// would also drain vmcnt, i.e. wait for every outstanding GLOBAL store of the row before each layer.)
__device__ inline void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// inverse_transform_with_support over the values held by the lanes whose output index is in [lo, lo+S) (muzero_model.py:575-591)
template <int U>
__device__ inline float decode_lanes(const float (&v)[U], int lo, int S, int lane) {
    float m = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o >= lo && o < lo + S) m = op_max(m, v[u]); }
    m = wave_max(m);
    float den = 0.f, num = 0.f;
    const int half = S / 2;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o >= lo && o < lo + S) { const float e = smz_exp(v[u] - m); den += e; num += (float)(o - lo - half) * e; }
    }
    wave_sum2(den, num);
    const float y = num / den;
    const float sg = (y > 0.f) ? 1.f : ((y < 0.f) ? -1.f : 0.f);
    const float r = (sqrtf(1.f + 4.f * 0.001f * (fabsf(y) + 1.f + 0.001f)) - 1.f) * 500.0f;
    return sg * (r * r - 1.f);
}

// scale_to_bound_action over lanes [lo, lo+S) (neural_network_mlp_model.py:349-357); writes act_out[o-lo] (LDS) and dst[o-lo]
template <int U>
__device__ inline void scale_lanes(const float (&v)[U], int lo, int S, int lane, float *act_out, float *dst,
                                   float *dst2 = nullptr) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o >= lo && o < lo + S) { mn = op_min(mn, v[u]); mx = op_max(mx, v[u]); } }
    wave_minmax(mn, mx);
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o >= lo && o < lo + S) {
            const float h = __fdividef(v[u] - mn, sc);
            act_out[o - lo] = h;
            if (dst) dst[o - lo] = h;
            if (dst2) dst2[o - lo] = h;
        }
    }
}

template <int U>
__device__ inline void softmax_lanes(const float (&v)[U], int A, int lane, float *dst) {
    float m = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < A) m = op_max(m, v[u]); }
    m = wave_max(m);
    float e[U], den = 0.f;
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; e[u] = (o < A) ? smz_exp(v[u] - m) : 0.f; den += e[u]; }
    den = wave_sum(den);
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < A) dst[o] = __fdividef(e[u], den); }
}

__device__ inline float support_to_scalar(float num, float den) {
    const float y = num / den;
    const float sg = (y > 0.f) ? 1.f : ((y < 0.f) ? -1.f : 0.f);
    const float r = (sqrtf(1.f + 4.f * 0.001f * (fabsf(y) + 1.f + 0.001f)) - 1.f) * 500.0f;
    return sg * (r * r - 1.f);
}

// prediction tail in two reduction chains: softmax over lanes [0, A) to dst (if non-null) and the support decode of
// lanes [A, A+S) -- the same arithmetic as softmax_lanes + decode_lanes (each lane belongs to one segment, the sums
// of the other segment see exact zeros), sharing one paired max chain, one exponential and one triple sum chain
template <int U>
__device__ inline float softmax_decode_lanes(const float (&v)[U], int A, int S, int lane, float *dst) {
    float mp = -__builtin_inff(), mv = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o < A) mp = op_max(mp, v[u]);
        else if (o < A + S) mv = op_max(mv, v[u]);
    }
    wave_max2(mp, mv);
    float e[U], dp = 0.f, dv = 0.f, nv = 0.f;
    const int half = S / 2;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        const bool pol = o < A, val = !pol && o < A + S;
        e[u] = (pol || val) ? smz_exp(v[u] - (pol ? mp : mv)) : 0.f;
        if (pol) dp += e[u];
        if (val) { dv += e[u]; nv += (float)(o - A - half) * e[u]; }
    }
    wave_sum3(dp, dv, nv);
    if (dst) {
#pragma unroll
        for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < A) dst[o] = __fdividef(e[u], dp); }
    }
    return support_to_scalar(nv, dv);
}

// dynamics tail in two chains: support decode of lanes [0, S) (reward) and scale_to_bound_action of lanes [S, 2S)
template <int U>
__device__ inline float decode_scale_lanes(const float (&v)[U], int S, int lane, float *act_out, float *dst) {
    float mr = -__builtin_inff(), mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o < S) mr = op_max(mr, v[u]);
        else if (o < 2 * S) { mn = op_min(mn, v[u]); mx = op_max(mx, v[u]); }
    }
    wave_max_min_max(mr, mn, mx);
    float den = 0.f, num = 0.f;
    const int half = S / 2;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o < S) { const float e = smz_exp(v[u] - mr); den += e; num += (float)(o - half) * e; }
    }
    wave_sum2(den, num);
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o >= S && o < 2 * S) {
            const float h = __fdividef(v[u] - mn, sc);
            act_out[o - S] = h;
            if (dst) dst[o - S] = h;
        }
    }
    return support_to_scalar(num, den);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the tails of a TWO-row pass evaluated for both rows at once.  v_permlane32_swap packs the rows' accumulators so
// that row h lives in lanes [32 h, 32 h + 32): P = outputs 0..31, Q = outputs 32..63 of both rows.  Every reduction then
// runs ONCE over 32-lane halves (row_shr 1 / 2 / 4 / 8 + row_bcast:15: five steps instead of six, results in lanes 31 and
// 63), the exponentials, quotients and the support decode run once for both rows, and the decode itself is computed in the
// lanes that hold the sums.  Bit-identical to the per-row functions above:
//   * max / min do not depend on the order;
//   * a wave_sum over 64 lanes is the balanced tree ((R0 + R1) + (R2 + R3)) of its 16-lane rows; the half-wide chain yields
//     R0 + R1 of the P part and, run on the Q part, R2 + R3 (a part whose lanes are all zero contributes an exact zero) --
//     their sum is the original total, term by term in the original association;
//   * every element-wise operation is the same instruction on the same operands.
// Requires U == 1, S <= 32 and all 64 lanes active.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ inline void swap32(float &a, float &b) {      // a = [a.lo | b.lo], b = [a.hi | b.hi]  (lo / hi: lanes 0..31 / 32..63)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// the value lane 31 holds for lanes 0..31, the value lane 63 holds for lanes 32..63
__device__ inline float half_bcast(float v, bool hi) {
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return hi ? s1 : s0;
}
#define SMZ_HSTEP1(I0, CTRL) I0 " %0, %0, %0 " CTRL "\n\t"
#define SMZ_HSTEP2(I0, I1, CTRL) I0 " %0, %0, %0 " CTRL "\n\t" I1 " %1, %1, %1 " CTRL "\n\t"
#define SMZ_HSTEP3(I0, I1, I2, CTRL) I0 " %0, %0, %0 " CTRL "\n\t" I1 " %1, %1, %1 " CTRL "\n\t" I2 " %2, %2, %2 " CTRL "\n\t"
#define SMZ_HSTEP5(I, CTRL) I " %0, %0, %0 " CTRL "\n\t" I " %1, %1, %1 " CTRL "\n\t" I " %2, %2, %2 " CTRL "\n\t" I " %3, %3, %3 " CTRL "\n\t" I " %4, %4, %4 " CTRL "\n\t"
#define SMZ_HALF_CHAIN(STEP, ...)                                                          \
    "s_nop 1\n\t" STEP(__VA_ARGS__, "row_shr:1 row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"  \
    STEP(__VA_ARGS__, "row_shr:2 row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"                \
    STEP(__VA_ARGS__, "row_shr:4 row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"                \
    STEP(__VA_ARGS__, "row_shr:8 row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"                \
    STEP(__VA_ARGS__, "row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 1"
// (results: lane 31 for the lower half, lane 63 for the upper)
__device__ inline void half_max2(float &a, float &b) {
    asm SMZ_DPP_VOLATILE(SMZ_HALF_CHAIN(SMZ_HSTEP2, "v_max_f32_dpp", "v_max_f32_dpp") : "+v"(a), "+v"(b));
}
__device__ inline void half_minmax(float &mn, float &mx) {
    asm SMZ_DPP_VOLATILE(SMZ_HALF_CHAIN(SMZ_HSTEP2, "v_min_f32_dpp", "v_max_f32_dpp") : "+v"(mn), "+v"(mx));
}
__device__ inline void half_max_min_max(float &a, float &b, float &c) {
    asm SMZ_DPP_VOLATILE(SMZ_HALF_CHAIN(SMZ_HSTEP3, "v_max_f32_dpp", "v_min_f32_dpp", "v_max_f32_dpp") : "+v"(a), "+v"(b), "+v"(c));
}
__device__ inline void half_sum2(float &a, float &b) {
    asm SMZ_DPP_VOLATILE(SMZ_HALF_CHAIN(SMZ_HSTEP2, "v_add_f32_dpp", "v_add_f32_dpp") : "+v"(a), "+v"(b));
}
__device__ inline void half_sum3(float &a, float &b, float &c) {
    asm SMZ_DPP_VOLATILE(SMZ_HALF_CHAIN(SMZ_HSTEP3, "v_add_f32_dpp", "v_add_f32_dpp", "v_add_f32_dpp") : "+v"(a), "+v"(b), "+v"(c));
}
__device__ inline void half_sum5(float &a, float &b, float &c, float &d, float &e) {
    asm SMZ_DPP_VOLATILE(SMZ_HALF_CHAIN(SMZ_HSTEP5, "v_add_f32_dpp") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}

// softmax_decode_lanes for two rows: policies to dst0 / dst1 (non-null), values returned
// WITH_REWARD: rnum / rden hold the dynamics tail's reward sums in lanes 31 / 63 (dynamics_tail_pair<true>); they move one
// lane down (DPP row_shl:1: lanes 30 / 62) into the registers whose lanes 31 / 63 hold the value's sums, and ONE
// support_to_scalar decodes all four (the same instruction sequence on the same operands: bit-identical).
template <bool WITH_REWARD = false>
__device__ inline void softmax_decode_pair(float a0, float a1, int A, int S, int lane, float *dst0, float *dst1, float &val0,
                                           float &val1, float rnum = 0.f, float rden = 0.f, float *rew0 = nullptr, float *rew1 = nullptr) {
    float P = a0, Q = a1;
    swap32(P, Q);
    const bool hi = lane >= 32;
    const int j = lane & 31, oq = 32 + j, half = S / 2;
    const bool ppol = j < A, pval = !ppol && j < A + S, qval = oq >= A && oq < A + S;      // (A <= 32: the policy lies in P)
    float mp = ppol ? P : -__builtin_inff();
    float mv = op_max(pval ? P : -__builtin_inff(), qval ? Q : -__builtin_inff());
    half_max2(mp, mv);
    mp = half_bcast(mp, hi); mv = half_bcast(mv, hi);
    const float eP = (ppol || pval) ? smz_exp(P - (ppol ? mp : mv)) : 0.f;
    const float eQ = qval ? smz_exp(Q - mv) : 0.f;
    float dp = ppol ? eP : 0.f, dvP = pval ? eP : 0.f, nvP = pval ? (float)(j - A - half) * eP : 0.f;
    float dv, nv;
    if (A + S == 33) {
        // the Q part is ONE output (o = 32, local lane 0): a tree sum over it and zeros is the element itself -- no chains for
        // it; it travels to the lanes that hold the P totals (row_mirror: lane 15 <- lane 0 of a 16-lane row; row_bcast:15:
        // the next row <- lane 15) and the same products / sums as below are formed there
        half_sum3(dp, dvP, nvP);
        int q = __builtin_amdgcn_update_dpp(0, __float_as_int(eQ), 0x140, 0xf, 0xf, false);
        q = __builtin_amdgcn_update_dpp(q, q, 0x142, 0xa, 0xf, false);
        const float e32 = __int_as_float(q);
        dv = dvP + e32;
        nv = nvP + (float)(32 - A - half) * e32;
    } else {
        float dvQ = eQ, nvQ = qval ? (float)(oq - A - half) * eQ : 0.f;
        half_sum5(dp, dvP, nvP, dvQ, nvQ);
        // lanes 31 / 63: the rows' totals in wave_sum3's association; the decode runs there, once for both rows
        dv = dvP + dvQ; nv = nvP + nvQ;
    }
    if constexpr (WITH_REWARD) {
        const float sn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rnum), 0x101, 0xf, 0xf, false));   // row_shl:1
        const float sd = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rden), 0x101, 0xf, 0xf, false));
        const bool rl = (lane & 31) == 30;
        nv = rl ? sn : nv;
        dv = rl ? sd : dv;
    }
    const float v = support_to_scalar(nv, dv);
    val0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    val1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    if constexpr (WITH_REWARD) {
        *rew0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 30));
        *rew1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 62));
    }
    const float dpb = half_bcast(dp, hi);
    float *dst = hi ? dst1 : dst0;
    if (ppol && dst) dst[j] = __fdividef(eP, dpb);
}

// The dynamics-layer tails of two rows, each row on its own branch (d0 / d1: dynamics = decode_scale_lanes -- reward logits
// [0, S) | next state [S, 2 S) --, afterstate = scale_lanes over [0, S)): which lanes are logits and which are state is a lane
// predicate per half.  Rewards returned (0 for an afterstate row); next states to act0 / act1 (LDS) and dst0 / dst1.
// With `defer` the reward's (num, den) -- lanes 31 / 63 of rnum / rden -- are handed on instead of being decoded: the
// prediction tail decodes them together with the value's (softmax_decode_pair: ONE support decode per pass).
template <bool DEFER = false>
__device__ inline void dynamics_tail_pair(float a0, float a1, bool d0, bool d1, int S, int lane, float *act0, float *act1,
                                          float *dst0, float *dst1, float &rew0, float &rew1, float *rnum = nullptr, float *rden = nullptr) {
    float P = a0, Q = a1;
    swap32(P, Q);
    const bool hi = lane >= 32, dyn = hi ? d1 : d0;
    const int j = lane & 31, oq = 32 + j, half = S / 2;
    const bool plog = dyn && j < S;                                                         // (S <= 32: the reward logits lie in P)
    const bool pst = dyn ? j >= S : j < S, qst = dyn && oq < 2 * S;
    float mr = plog ? P : -__builtin_inff();
    float mn = op_min(pst ? P : __builtin_inff(), qst ? Q : __builtin_inff());
    float mx = op_max(pst ? P : -__builtin_inff(), qst ? Q : -__builtin_inff());
    half_max_min_max(mr, mn, mx);
    mr = half_bcast(mr, hi); mn = half_bcast(mn, hi); mx = half_bcast(mx, hi);
    const float e = plog ? smz_exp(P - mr) : 0.f;
    float den = e, num = plog ? (float)(j - half) * e : 0.f;
    half_sum2(den, num);
    if constexpr (DEFER) {
        *rnum = num; *rden = den;
        rew0 = rew1 = 0.f;
    } else {
        const float r = support_to_scalar(num, den);
        rew0 = d0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 31)) : 0.f;
        rew1 = d1 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 63)) : 0.f;
    }
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;
    float *act = hi ? act1 : act0, *dst = hi ? dst1 : dst0;
    const int ip = dyn ? j - S : j;
    if (pst) { const float h = __fdividef(P - mn, sc); act[ip] = h; if (dst) dst[ip] = h; }
    if (qst) { const float h = __fdividef(Q - mn, sc); act[oq - S] = h; if (dst) dst[oq - S] = h; }
}

// scale_lanes (outputs [0, S)) for two rows of the afterstate branch
// PACKED: a0 already holds both rows' outputs on lane halves (dense_halves)
template <bool PACKED = false>
__device__ inline void scale_pair(float a0, float a1, int S, int lane, float *act0, float *act1, float *dst0, float *dst1) {
    float P = a0, Q = a1;
    if (!PACKED) swap32(P, Q);
    const bool hi = lane >= 32;
    const int j = lane & 31;
    const bool in = j < S;                                                                  // (S <= 32)
    float mn = in ? P : __builtin_inff(), mx = in ? P : -__builtin_inff();
    half_minmax(mn, mx);
    mn = half_bcast(mn, hi); mx = half_bcast(mx, hi);
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;
    float *act = hi ? act1 : act0, *dst = hi ? dst1 : dst0;
    if (in) { const float h = __fdividef(P - mn, sc); act[j] = h; if (dst) dst[j] = h; }
}

// Float offsets of one matrix and its bias.  A descriptor's off[] must only ever be indexed with compile-time
// constants: a run-time index (`off[dyn ? M_DYN_IN : M_ADY_IN]`) makes the compiler keep the whole table in scratch
// memory and turns every layer's set-up into a global-memory round trip (measured: the dominant stall of a leaf
// evaluation).  pick() selects between two constant-index entries instead.
struct MatOff {
    int w, b, op;
};
// CP ("compact", the LDS image of the single-launch search kernel when it also keeps its trees in LDS): a matrix is stored only
// as wide as its outputs need, rounded up to 4 -- the afterstate-dynamics output layer 32 instead of 64, the two prediction
// output layers 36 (A + S = 33 or 35 outputs) -- which frees 22 KB of the 100 KB the 64-wide image takes.  A lane beyond a
// matrix's width reads the next group's weights and computes a value nobody uses (the tails mask by output index).
template <bool CP = false>
__device__ __host__ inline int mat_op(const smz_mlp_desc &d, int m) {
    if (!CP) return d.OP;
    const int o = (m == M_ADY_OUT) ? up4(d.S) : ((m == M_PRE_OUT || m == M_APR_OUT) ? up4(d.A + d.S) : ((m == M_DYN_OUT) ? up4(2 * d.S) : up4(d.H)));
    return o < d.OP ? o : d.OP;
}
// Offset of matrix m computed from the dimensions (the formula smz_mlp_layout fills off[] with: matrices in enum order,
// up4(K_m) * OP floats each), so that the search loop keeps S, A, H, L, OP and one bias base live instead of thirty
// table entries (the kernel is short of scalar registers: spilled SGPRs cost a v_readlane each time they are needed).
// m is a compile-time constant at every call site, so the sum folds to a few scalar operations.
template <bool CP = false>
__device__ __host__ inline int mat_off(const smz_mlp_desc &d, int m) {
    const int kin = up4(d.S + d.A), kmid = d.L > 0 ? up4(d.H) : 0, kh = up4(d.H), ks = up4(d.S), ko = up4(d.obs);
    const int K[M_COUNT] = {kin, kin, kmid, kmid, kh, kh, ks, ks, kmid, kmid, kh, kh, ko, kmid, kh};
    int floats = 0;
#pragma unroll
    for (int j = 0; j < M_COUNT; j++) floats += j < m ? K[j] * mat_op<CP>(d, j) : 0;
    return floats;
}
template <bool CP = false>
__device__ inline MatOff pick(const smz_mlp_desc &d, bool first, int ma, int mb) {   // ma, mb: constants at every call site
    MatOff o;
    const int bias0 = d.off[M_COUNT];            // (moved down in the LDS copy that leaves the representation out)
    o.w = first ? mat_off<CP>(d, ma) : mat_off<CP>(d, mb);
    o.b = bias0 + (first ? ma : mb) * d.OP;
    o.op = first ? mat_op<CP>(d, ma) : mat_op<CP>(d, mb);
    return o;
}

// hidden trunk for R rows: in-layer + L repeats of the shared mid layer, ELU after each; results in tA[r] (LDS, zero
// padded to a multiple of 4).  o_in[r] / o_mid[r]: each row's matrices.
template <int U, int R, bool SAME = false>
__device__ inline void trunk(const float *lds, const smz_mlp_desc &d, const MatOff (&o_in)[R], const MatOff (&o_mid)[R],
                             const float *const (&act_in)[R], int K4in, float *const (&tA)[R], int lane) {
    float acc[R][U];
    const float *W[R], *Bv[R];
    int op[R];
#pragma unroll
    for (int r = 0; r < R; r++) { W[r] = lds + o_in[r].w; Bv[r] = lds + o_in[r].b; op[r] = o_in[r].op; }
    dense<U, R, SAME>(W, Bv, act_in, K4in, op, lane, acc);
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < up4(d.H)) tA[r][o] = (o < d.H) ? elu(acc[r][u]) : 0.f; }
    lds_sync();
    for (int l = 0; l < d.L; l++) {
        const float *Wm[R], *Bm[R], *Am[R];
        int opm[R];
#pragma unroll
        for (int r = 0; r < R; r++) { Wm[r] = lds + o_mid[r].w; Bm[r] = lds + o_mid[r].b; Am[r] = tA[r]; opm[r] = o_mid[r].op; }
        dense<U, R, SAME>(Wm, Bm, Am, up4(d.H), opm, lane, acc);
        lds_sync();   // every lane has issued its reads of tA (LDS is in order per wave) before it is overwritten
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < up4(d.H)) tA[r][o] = (o < d.H) ? elu(acc[r][u]) : 0.f; }
        lds_sync();
    }
}

// Copies (offset, count) ranges of the packed buffer to the SAME offsets in LDS.  Loads are issued in batches of 8
// float4 per thread before the first LDS store so that the global latency is paid once per batch, not per element.
__device__ inline void stage_weights(float *lds, const float *weights, const int *ranges, int n_ranges) {
    constexpr int UB = 8;
    for (int r = 0; r < n_ranges; r++) {
        const int off = ranges[2 * r], cnt = ranges[2 * r + 1];
        for (int i0 = threadIdx.x * 4; i0 < cnt; i0 += blockDim.x * 4 * UB) {
            float4 v[UB];       // unconditional (clamped) loads: a conditionally written v[] ends up in scratch memory
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int i = i0 + u * blockDim.x * 4;
                v[u] = *reinterpret_cast<const float4 *>(weights + off + (i < cnt ? i : i0));
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int i = i0 + u * blockDim.x * 4;
                if (i < cnt) *reinterpret_cast<float4 *>(lds + off + i) = v[u];
            }
        }
    }
    __syncthreads();
}

constexpr int kRows = 1;   // rows per pass of the stand-alone heads kernel and of the generic search kernel (the specialised
                           // search kernel evaluates its two leaves as one two-row pass: smz_kernels.hip)

// per-row LDS scratch: input vector | trunk activations | hidden state; a wave owns kRows of them
__host__ __device__ inline int row_scratch_floats(const smz_mlp_desc &d) {
    const int kin = up4(d.S + d.A) > up4(d.obs) ? up4(d.S + d.A) : up4(d.obs);
    return kin + up4(d.H) + up4(d.S);
}
__host__ __device__ inline int scratch_floats(const smz_mlp_desc &d) { return kRows * row_scratch_floats(d); }

// R recurrent evaluations (monte_carlo_tree_search.py:333-342) by one wavefront.  xin[r]: the row's network input
// [hidden | one-hot], K4in floats, zero padded, in LDS.  dyn[r]: the row's branch.  live[r] = false suppresses the
// row's global stores (odd tail).  Writes hidden' to dst_hidden[r] (S floats), the policy to dst_policy[r]; reward and
// value are returned in every lane.
template <int U, int R, bool SAME = false, bool CP = false>
__device__ inline void recurrent_rows(const float *lds, const smz_mlp_desc &d, float *scratch, const float *const (&xin)[R],
                                      const bool (&dyn)[R], const bool (&live)[R], float *const (&dst_hidden)[R],
                                      float *const (&dst_policy)[R], float (&reward)[R], float (&value)[R]) {
    const int S = d.S, A = d.A, K4in = up4(S + A), K4h = up4(d.H), K4s = up4(S);
    const int lane = threadIdx.x & (kWave - 1);
    const int rs = row_scratch_floats(d), kin = rs - K4h - K4s;
    float *tA[R], *hbuf[R];
    MatOff m1[R], m1m[R], m3[R], m3m[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        tA[r] = scratch + r * rs + kin;
        hbuf[r] = tA[r] + K4h;
        m1[r] = pick<CP>(d, dyn[r], M_DYN_IN, M_ADY_IN);   m1m[r] = pick<CP>(d, dyn[r], M_DYN_MID, M_ADY_MID);
        m3[r] = pick<CP>(d, dyn[r], M_PRE_IN, M_APR_IN);   m3m[r] = pick<CP>(d, dyn[r], M_PRE_MID, M_APR_MID);
        for (int k = lane; k < K4s; k += kWave) hbuf[r][k] = 0.f;
    }
    trunk<U, R, SAME>(lds, d, m1, m1m, xin, K4in, tA, lane);
    float acc[R][U];
    constexpr bool PAIRED = SMZ_PAIR_TAILS && U == 1 && R == 2;        // both rows' tails at once (see softmax_decode_pair)
    // both rows on the afterstate branch: its dynamics layer has S <= 32 outputs -- both rows at once on lane halves (dense_halves)
    const bool halves = PAIRED && SAME && SMZ_DENSE_HALVES && S <= 32 && !dyn[0] && !dyn[1];       // (wave-uniform)
    if (halves) {
        if constexpr (PAIRED && SAME) {
            const MatOff m = pick<CP>(d, false, M_DYN_OUT, M_ADY_OUT);
            acc[0][0] = dense_halves(lds + m.w, lds + m.b, tA[0], tA[1], K4h, m.op, lane);
            acc[1][0] = 0.f;
        }
    } else {
        const float *W[R], *Bv[R], *Ac[R];
        int op[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const MatOff m = pick<CP>(d, dyn[r], M_DYN_OUT, M_ADY_OUT);
            W[r] = lds + m.w; Bv[r] = lds + m.b; Ac[r] = tA[r]; op[r] = m.op;
        }
        dense<U, R, SAME>(W, Bv, Ac, K4h, op, lane, acc);
    }
    float pair_rnum = 0.f, pair_rden = 0.f;                            // (the reward's sums of a dynamics row, decoded with the value's)
    bool pair_reward = false;
    if constexpr (PAIRED) {
        if (S <= 32) {
            reward[0] = reward[1] = 0.f;
            if (halves) scale_pair<true>(acc[0][0], 0.f, S, lane, hbuf[0], hbuf[1], live[0] ? dst_hidden[0] : nullptr,
                                         live[1] ? dst_hidden[1] : nullptr);
            else if (!dyn[0] && !dyn[1]) scale_pair(acc[0][0], acc[1][0], S, lane, hbuf[0], hbuf[1], live[0] ? dst_hidden[0] : nullptr,
                                               live[1] ? dst_hidden[1] : nullptr);
            else if (SMZ_ONE_DECODE) {
                dynamics_tail_pair<true>(acc[0][0], acc[1][0], dyn[0], dyn[1], S, lane, hbuf[0], hbuf[1], live[0] ? dst_hidden[0] : nullptr,
                                         live[1] ? dst_hidden[1] : nullptr, reward[0], reward[1], &pair_rnum, &pair_rden);
                pair_reward = true;
            } else dynamics_tail_pair(acc[0][0], acc[1][0], dyn[0], dyn[1], S, lane, hbuf[0], hbuf[1], live[0] ? dst_hidden[0] : nullptr,
                                      live[1] ? dst_hidden[1] : nullptr, reward[0], reward[1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
                reward[r] = 0.f;
                if (dyn[r]) reward[r] = decode_scale_lanes<U>(acc[r], S, lane, hbuf[r], live[r] ? dst_hidden[r] : nullptr);
                else scale_lanes<U>(acc[r], 0, S, lane, hbuf[r], live[r] ? dst_hidden[r] : nullptr);
            }
        }
    } else {
#pragma unroll
    for (int r = 0; r < R; r++) {
        reward[r] = 0.f;
        if (dyn[r]) {       // [reward logits | next state]
            reward[r] = decode_scale_lanes<U>(acc[r], S, lane, hbuf[r], live[r] ? dst_hidden[r] : nullptr);
        } else {
            scale_lanes<U>(acc[r], 0, S, lane, hbuf[r], live[r] ? dst_hidden[r] : nullptr);
        }
    }
    }
    lds_sync();
    {
        const float *Hc[R];
#pragma unroll
        for (int r = 0; r < R; r++) Hc[r] = hbuf[r];
        trunk<U, R, SAME>(lds, d, m3, m3m, Hc, K4s, tA, lane);
        const float *W[R], *Bv[R], *Ac[R];
        int op[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const MatOff m = pick<CP>(d, dyn[r], M_PRE_OUT, M_APR_OUT);
            W[r] = lds + m.w; Bv[r] = lds + m.b; Ac[r] = tA[r]; op[r] = m.op;
        }
        dense<U, R, SAME>(W, Bv, Ac, K4h, op, lane, acc);
    }
    if constexpr (PAIRED) {
        if (S <= 32 && A <= 32 && pair_reward) {                       // (wave-uniform)
            float r0, r1;
            softmax_decode_pair<true>(acc[0][0], acc[1][0], A, S, lane, live[0] ? dst_policy[0] : nullptr,
                                      live[1] ? dst_policy[1] : nullptr, value[0], value[1], pair_rnum, pair_rden, &r0, &r1);
            reward[0] = dyn[0] ? r0 : 0.f;
            reward[1] = dyn[1] ? r1 : 0.f;
        } else if (S <= 32 && A <= 32) {
            softmax_decode_pair(acc[0][0], acc[1][0], A, S, lane, live[0] ? dst_policy[0] : nullptr, live[1] ? dst_policy[1] : nullptr,
                                value[0], value[1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) value[r] = softmax_decode_lanes<U>(acc[r], A, S, lane, live[r] ? dst_policy[r] : nullptr);
        }
    } else {
#pragma unroll
    for (int r = 0; r < R; r++) {
        value[r] = softmax_decode_lanes<U>(acc[r], A, S, lane, live[r] ? dst_policy[r] : nullptr);
    }
    }
    lds_sync();
}

// representation + root prediction for one observation row by one wavefront (muzero_model.py:802-841)
// `rep`/`drep`: where the representation matrices live (LDS, or the packed buffer in global memory: they are used once
// per search, so the whole-search kernel does not spend LDS on them); `lds`/`d`: the prediction matrices.
template <int U, bool CP = false>
__device__ inline void initial_row(const float *lds, const smz_mlp_desc &d, const float *rep, const smz_mlp_desc &drep,
                                   float *scratch, const float *obs_row, float *dst_hidden0, float *dst_hidden1,
                                   float *dst_policy) {
    const int S = d.S, A = d.A, K4o = up4(d.obs), K4h = up4(d.H), K4s = up4(S);
    const int lane = threadIdx.x & (kWave - 1);
    const int rs = row_scratch_floats(d), kin = rs - K4h - K4s;
    float *xbuf = scratch;
    float *tA[1] = {scratch + kin};
    float *hbuf = tA[0] + K4h;
    for (int k = lane; k < K4o; k += kWave) xbuf[k] = (k < d.obs) ? obs_row[k] : 0.f;
    for (int k = lane; k < K4s; k += kWave) hbuf[k] = 0.f;
    lds_sync();
    const MatOff mi[1] = {pick(drep, true, M_REP_IN, M_REP_IN)}, mm[1] = {pick(drep, true, M_REP_MID, M_REP_MID)};
    const float *xi[1] = {xbuf};
    trunk<U, 1>(rep, drep, mi, mm, xi, K4o, tA, lane);
    float acc[1][U];
    {
        const float *W[1] = {rep + drep.off[M_REP_OUT]}, *Bv[1] = {rep + drep.off[M_COUNT + M_REP_OUT]}, *Ac[1] = {tA[0]};
        dense<U, 1>(W, Bv, Ac, K4h, d.OP, lane, acc);
    }
    scale_lanes<U>(acc[0], 0, S, lane, hbuf, dst_hidden0, dst_hidden1);
    lds_sync();
    const MatOff pi[1] = {pick<CP>(d, true, M_PRE_IN, M_PRE_IN)}, pm[1] = {pick<CP>(d, true, M_PRE_MID, M_PRE_MID)};
    const float *hi[1] = {hbuf};
    trunk<U, 1>(lds, d, pi, pm, hi, K4s, tA, lane);
    {
        const MatOff po = pick<CP>(d, true, M_PRE_OUT, M_PRE_OUT);
        const float *W[1] = {lds + po.w}, *Bv[1] = {lds + po.b}, *Ac[1] = {tA[0]};
        dense<U, 1>(W, Bv, Ac, K4h, po.op, lane, acc);
    }
    softmax_lanes<U>(acc[0], A, lane, dst_policy);
    lds_sync();
}

// Descriptor of the LDS copy that leaves the three representation matrices out: matrices 0..11 keep their offsets, the
// bias block moves down by the size of the representation matrices (the rep entries themselves become invalid).
__host__ __device__ inline int rep_floats(const smz_mlp_desc &d) { return d.off[M_COUNT] - d.off[M_REP_IN]; }
__host__ __device__ inline smz_mlp_desc lds_desc_without_rep(const smz_mlp_desc &d) {
    smz_mlp_desc l = d;
    const int rs = rep_floats(d);
    for (int m = 0; m < M_COUNT; m++) l.off[M_COUNT + m] = d.off[M_COUNT + m] - rs;
    l.total_floats = d.total_floats - rs;
    return l;
}
__device__ inline void stage_weights_without_rep(float *lds, const float *weights, const smz_mlp_desc &d) {
    constexpr int UB = 8;
    const int n_mat = d.off[M_REP_IN], n_bias = d.total_floats - d.off[M_COUNT], rs = rep_floats(d);
    for (int pass = 0; pass < 2; pass++) {
        const float *src = pass ? weights + d.off[M_COUNT] : weights;
        float *dst = pass ? lds + d.off[M_COUNT] - rs : lds;
        const int cnt = pass ? n_bias : n_mat;
        for (int i0 = threadIdx.x * 4; i0 < cnt; i0 += blockDim.x * 4 * UB) {
            float4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) { const int i = i0 + u * blockDim.x * 4; v[u] = *reinterpret_cast<const float4 *>(src + (i < cnt ? i : i0)); }
#pragma unroll
            for (int u = 0; u < UB; u++) { const int i = i0 + u * blockDim.x * 4; if (i < cnt) *reinterpret_cast<float4 *>(dst + i) = v[u]; }
        }
    }
    __syncthreads();
}

// The compact LDS image (mat_op<true>): matrices 0..11 at mat_off<true>, every 4-input group mat_op floats x 4 wide instead of
// OP x 4; the bias block (64 wide per matrix, unchanged) right behind them.  Only off[M_COUNT] (the bias base pick() uses) and
// total_floats of the returned descriptor are meaningful.
__host__ __device__ inline int compact_matrix_floats(const smz_mlp_desc &d) { return mat_off<true>(d, M_REP_IN); }
__host__ __device__ inline int compact_total_floats(const smz_mlp_desc &d) { return compact_matrix_floats(d) + (d.total_floats - d.off[M_COUNT]); }
__device__ inline smz_mlp_desc lds_desc_compact(const smz_mlp_desc &d) {
    smz_mlp_desc l = d;
    l.off[M_COUNT] = compact_matrix_floats(d);
    l.total_floats = compact_total_floats(d);
    return l;
}
__device__ inline void stage_weights_compact(float *lds, const float *weights, const smz_mlp_desc &d) {
    const int kin = up4(d.S + d.A), kmid = d.L > 0 ? up4(d.H) : 0, kh = up4(d.H), ks = up4(d.S);
    const int K[M_REP_IN] = {kin, kin, kmid, kmid, kh, kh, ks, ks, kmid, kmid, kh, kh};
    int dst = 0;
#pragma unroll
    for (int m = 0; m < M_REP_IN; m++) {
        const int op = mat_op<true>(d, m), groups = K[m] >> 2, per = op;            // `per` float4 per group
        const float4 *src = reinterpret_cast<const float4 *>(weights + d.off[m]);
        float4 *out = reinterpret_cast<float4 *>(lds + dst);
        for (int i = threadIdx.x; i < groups * per; i += blockDim.x) {
            const int g = i / per, o = i - g * per;
            out[i] = src[g * d.OP + o];
        }
        dst += K[m] * op;
    }
    const int n_bias = d.total_floats - d.off[M_COUNT];
    for (int i = threadIdx.x * 4; i < n_bias; i += blockDim.x * 4)
        *reinterpret_cast<float4 *>(lds + dst + i) = *reinterpret_cast<const float4 *>(weights + d.off[M_COUNT] + i);
    __syncthreads();
}

// which parts of the packed buffer each phase needs
__device__ inline void stage_recurrent_weights(float *lds, const float *weights, const smz_mlp_desc &d) {
    const int ranges[4] = {0, d.off[M_REP_IN], d.off[M_COUNT], d.off[M_COUNT + M_REP_IN] - d.off[M_COUNT]};
    stage_weights(lds, weights, ranges, 2);
}
__device__ inline void stage_all_weights(float *lds, const float *weights, const smz_mlp_desc &d) {
    const int ranges[2] = {0, d.total_floats};
    stage_weights(lds, weights, ranges, 1);
}
__device__ inline void stage_initial_weights(float *lds, const float *weights, const smz_mlp_desc &d) {
    const int ranges[10] = {d.off[M_REP_IN], d.off[M_COUNT] - d.off[M_REP_IN],          // rep_in, rep_mid, rep_out
                            d.off[M_PRE_IN], d.off[M_APR_IN] - d.off[M_PRE_IN],          // pre_in
                            d.off[M_PRE_MID], d.off[M_APR_MID] - d.off[M_PRE_MID],       // pre_mid
                            d.off[M_PRE_OUT], d.off[M_APR_OUT] - d.off[M_PRE_OUT],       // pre_out
                            d.off[M_COUNT], d.total_floats - d.off[M_COUNT]};            // all biases
    stage_weights(lds, weights, ranges, 5);
}

}  // namespace smz_mlp
