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

// acc[u] = bias[o] + sum_k W[k][o] * act[k]  for o = lane + 64 u  (k ascending into one accumulator: every caller
// rounds identically).  Four 4-wide steps per iteration with their eight 16-byte LDS reads issued back to back, so a
// single LDS latency covers 16 inputs (the plain loop is a chain of 49 exposed LDS round trips per row).
template <int U>
__device__ inline void dense(const float *W, const float *bias, const float *act, int K4, int OP, int lane, float (&acc)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) acc[u] = bias[lane + kWave * u];
    const float4 *a4 = reinterpret_cast<const float4 *>(act);
    const float4 *w4 = reinterpret_cast<const float4 *>(W) + lane;
    const int n = K4 >> 2;
    int q = 0;
    for (; q + 4 <= n; q += 4) {
        const float4 a0 = a4[q], a1 = a4[q + 1], a2 = a4[q + 2], a3 = a4[q + 3];
        float4 w0[U], w1[U], w2[U], w3[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            w0[u] = w4[(size_t)q * OP + kWave * u];
            w1[u] = w4[(size_t)(q + 1) * OP + kWave * u];
            w2[u] = w4[(size_t)(q + 2) * OP + kWave * u];
            w3[u] = w4[(size_t)(q + 3) * OP + kWave * u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float r = acc[u];
            r = fmaf(w0[u].x, a0.x, r); r = fmaf(w0[u].y, a0.y, r); r = fmaf(w0[u].z, a0.z, r); r = fmaf(w0[u].w, a0.w, r);
            r = fmaf(w1[u].x, a1.x, r); r = fmaf(w1[u].y, a1.y, r); r = fmaf(w1[u].z, a1.z, r); r = fmaf(w1[u].w, a1.w, r);
            r = fmaf(w2[u].x, a2.x, r); r = fmaf(w2[u].y, a2.y, r); r = fmaf(w2[u].z, a2.z, r); r = fmaf(w2[u].w, a2.w, r);
            r = fmaf(w3[u].x, a3.x, r); r = fmaf(w3[u].y, a3.y, r); r = fmaf(w3[u].z, a3.z, r); r = fmaf(w3[u].w, a3.w, r);
            acc[u] = r;
        }
    }
    for (; q < n; q++) {
        const float4 a = a4[q];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float4 w = w4[(size_t)q * OP + kWave * u];
            float r = acc[u];
            r = fmaf(w.x, a.x, r); r = fmaf(w.y, a.y, r); r = fmaf(w.z, a.z, r); r = fmaf(w.w, a.w, r);
            acc[u] = r;
        }
    }
}

// torch's ELU evaluates exp(x) - 1 (aten/src/ATen/native/cpu/Activation.cpp elu_kernel)
__device__ inline float elu(float x) { return x > 0.f ? x : expf(x) - 1.0f; }
// Wave-wide reductions on the DPP crossbar (VALU latency) instead of ds_bpermute round trips: row_shr 1/2/4/8 build the
// per-16-lane-row totals in lanes 15/31/47/63, row_bcast:15 / row_bcast:31 carry them across rows, lane 63 holds the
// result and is broadcast through an SGPR.  `ident` fills the lanes a shift has no source for.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_move(float ident, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#define SMZ_WAVE_REDUCE(NAME, OP, IDENT)                                   \
    __device__ inline float NAME(float v) {                                \
        const float id = IDENT;                                            \
        v = OP(v, dpp_move<0x111, 0xf>(id, v)); /* row_shr:1 */            \
        v = OP(v, dpp_move<0x112, 0xf>(id, v)); /* row_shr:2 */            \
        v = OP(v, dpp_move<0x114, 0xf>(id, v)); /* row_shr:4 */            \
        v = OP(v, dpp_move<0x118, 0xf>(id, v)); /* row_shr:8 */            \
        v = OP(v, dpp_move<0x142, 0xa>(id, v)); /* row_bcast:15 -> rows 1,3 */ \
        v = OP(v, dpp_move<0x143, 0xc>(id, v)); /* row_bcast:31 -> rows 2,3 */ \
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); \
    }
__device__ inline float op_add(float a, float b) { return a + b; }
SMZ_WAVE_REDUCE(wave_max, fmaxf, -__builtin_inff())
SMZ_WAVE_REDUCE(wave_min, fminf, __builtin_inff())
SMZ_WAVE_REDUCE(wave_sum, op_add, 0.0f)
#undef SMZ_WAVE_REDUCE
// Orders this wave's LDS traffic: LDS instructions of one wave execute in issue order, so other lanes' earlier writes are
// visible once they have been issued; the asm is a compiler barrier plus an LDS-counter wait.  (A scoped fence here
// would also drain vmcnt, i.e. wait for every outstanding GLOBAL store of the row before each layer.)
__device__ inline void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// inverse_transform_with_support over the values held by the lanes whose output index is in [lo, lo+S) (muzero_model.py:575-591)
template <int U>
__device__ inline float decode_lanes(const float (&v)[U], int lo, int S, int lane) {
    float m = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o >= lo && o < lo + S) m = fmaxf(m, v[u]); }
    m = wave_max(m);
    float den = 0.f, num = 0.f;
    const int half = S / 2;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o >= lo && o < lo + S) { const float e = expf(v[u] - m); den += e; num += (float)(o - lo - half) * e; }
    }
    den = wave_sum(den);
    num = wave_sum(num);
    const float y = num / den;
    const float sg = (y > 0.f) ? 1.f : ((y < 0.f) ? -1.f : 0.f);
    const float r = (sqrtf(1.f + 4.f * 0.001f * (fabsf(y) + 1.f + 0.001f)) - 1.f) / (2.f * 0.001f);
    return sg * (r * r - 1.f);
}

// scale_to_bound_action over lanes [lo, lo+S) (neural_network_mlp_model.py:349-357); writes act_out[o-lo] (LDS) and dst[o-lo]
template <int U>
__device__ inline void scale_lanes(const float (&v)[U], int lo, int S, int lane, float *act_out, float *dst,
                                   float *dst2 = nullptr) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o >= lo && o < lo + S) { mn = fminf(mn, v[u]); mx = fmaxf(mx, v[u]); } }
    mn = wave_min(mn);
    mx = wave_max(mx);
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int o = lane + kWave * u;
        if (o >= lo && o < lo + S) {
            const float h = (v[u] - mn) / sc;
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
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < A) m = fmaxf(m, v[u]); }
    m = wave_max(m);
    float den = 0.f;
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < A) den += expf(v[u] - m); }
    den = wave_sum(den);
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < A) dst[o] = expf(v[u] - m) / den; }
}

// hidden trunk: in-layer + L repeats of the shared mid layer, ELU after each; result in `bufA` (LDS, zero padded to K4h)
template <int U>
__device__ inline void trunk(const float *lds, const smz_mlp_desc &d, int m_in, int m_mid, const float *act_in, int K4in,
                             float *bufA, float *bufB, int lane) {
    float acc[U];
    dense<U>(lds + d.off[m_in], lds + d.off[M_COUNT + m_in], act_in, K4in, d.OP, lane, acc);
#pragma unroll
    for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < up4(d.H)) bufA[o] = (o < d.H) ? elu(acc[u]) : 0.f; }
    lds_sync();
    for (int l = 0; l < d.L; l++) {
        dense<U>(lds + d.off[m_mid], lds + d.off[M_COUNT + m_mid], bufA, up4(d.H), d.OP, lane, acc);
        lds_sync();   // every lane has issued its reads of bufA (LDS is in order per wave) before it is overwritten
#pragma unroll
        for (int u = 0; u < U; u++) { const int o = lane + kWave * u; if (o < up4(d.H)) bufA[o] = (o < d.H) ? elu(acc[u]) : 0.f; }
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
            float4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int i = i0 + u * blockDim.x * 4;
                if (i < cnt) v[u] = *reinterpret_cast<const float4 *>(weights + off + i);
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

__host__ __device__ inline int scratch_floats(const smz_mlp_desc &d) {
    const int kin = up4(d.S + d.A) > up4(d.obs) ? up4(d.S + d.A) : up4(d.obs);
    return kin + 2 * up4(d.H) + up4(d.S);
}

// One recurrent evaluation (monte_carlo_tree_search.py:333-342) by one wavefront.  `scratch` = this wave's LDS scratch
// (scratch_floats() floats).  `xsrc_h` = parent hidden row (S floats, global), `action` = last action.
// Writes hidden' to dst_hidden0 and (if non-null) dst_hidden1 (S floats each), policy (A floats), returns reward/value
// in every lane.
template <int U>
__device__ inline void recurrent_row(const float *lds, const smz_mlp_desc &d, float *scratch, const float *xsrc_h,
                                     const float *xsrc_full, int action, bool dyn, float *dst_hidden0,
                                     float *dst_hidden1, float *dst_policy, float &reward, float &value) {
    const int S = d.S, A = d.A, K4in = up4(S + A), K4h = up4(d.H), K4s = up4(S);
    const int lane = threadIdx.x & (kWave - 1);
    float *xbuf = scratch, *tA = xbuf + (up4(d.S + d.A) > up4(d.obs) ? up4(d.S + d.A) : up4(d.obs)), *tB = tA + K4h, *hbuf = tB + K4h;
    for (int k = lane; k < K4in; k += kWave) {
        float v = 0.f;
        if (k < S + A) v = xsrc_full ? xsrc_full[k] : (k < S ? xsrc_h[k] : ((k - S) == action ? 1.f : 0.f));
        xbuf[k] = v;
    }
    for (int k = lane; k < K4s; k += kWave) hbuf[k] = 0.f;
    lds_sync();
    trunk<U>(lds, d, dyn ? M_DYN_IN : M_ADY_IN, dyn ? M_DYN_MID : M_ADY_MID, xbuf, K4in, tA, tB, lane);
    float acc[U];
    reward = 0.f;
    if (dyn) {
        dense<U>(lds + d.off[M_DYN_OUT], lds + d.off[M_COUNT + M_DYN_OUT], tA, K4h, d.OP, lane, acc);
        reward = decode_lanes<U>(acc, 0, S, lane);
        scale_lanes<U>(acc, S, S, lane, hbuf, dst_hidden0, dst_hidden1);
    } else {
        dense<U>(lds + d.off[M_ADY_OUT], lds + d.off[M_COUNT + M_ADY_OUT], tA, K4h, d.OP, lane, acc);
        scale_lanes<U>(acc, 0, S, lane, hbuf, dst_hidden0, dst_hidden1);
    }
    lds_sync();
    trunk<U>(lds, d, dyn ? M_PRE_IN : M_APR_IN, dyn ? M_PRE_MID : M_APR_MID, hbuf, K4s, tA, tB, lane);
    dense<U>(lds + d.off[dyn ? M_PRE_OUT : M_APR_OUT], lds + d.off[M_COUNT + (dyn ? M_PRE_OUT : M_APR_OUT)], tA, K4h,
             d.OP, lane, acc);
    softmax_lanes<U>(acc, A, lane, dst_policy);
    value = decode_lanes<U>(acc, A, S, lane);
    lds_sync();
}

// representation + root prediction for one observation row by one wavefront (muzero_model.py:802-841)
template <int U>
__device__ inline void initial_row(const float *lds, const smz_mlp_desc &d, float *scratch, const float *obs_row,
                                   float *dst_hidden0, float *dst_hidden1, float *dst_policy) {
    const int S = d.S, A = d.A, K4o = up4(d.obs), K4h = up4(d.H), K4s = up4(S);
    const int lane = threadIdx.x & (kWave - 1);
    float *xbuf = scratch, *tA = xbuf + (up4(d.S + d.A) > up4(d.obs) ? up4(d.S + d.A) : up4(d.obs)), *tB = tA + K4h, *hbuf = tB + K4h;
    for (int k = lane; k < K4o; k += kWave) xbuf[k] = (k < d.obs) ? obs_row[k] : 0.f;
    for (int k = lane; k < K4s; k += kWave) hbuf[k] = 0.f;
    lds_sync();
    trunk<U>(lds, d, M_REP_IN, M_REP_MID, xbuf, K4o, tA, tB, lane);
    float acc[U];
    dense<U>(lds + d.off[M_REP_OUT], lds + d.off[M_COUNT + M_REP_OUT], tA, K4h, d.OP, lane, acc);
    scale_lanes<U>(acc, 0, S, lane, hbuf, dst_hidden0, dst_hidden1);
    lds_sync();
    trunk<U>(lds, d, M_PRE_IN, M_PRE_MID, hbuf, K4s, tA, tB, lane);
    dense<U>(lds + d.off[M_PRE_OUT], lds + d.off[M_COUNT + M_PRE_OUT], tA, K4h, d.OP, lane, acc);
    softmax_lanes<U>(acc, A, lane, dst_policy);
    lds_sync();
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
