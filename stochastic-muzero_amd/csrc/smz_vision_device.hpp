// smz_vision_device.hpp -- device pieces of the `vision_model` family shared by the wave-per-leaf kernels (smz_vision.hip)
// and the single-launch vision search (smz_vision_search.hip): the 3x7x7 hidden state lives one pixel per lane with its
// three channels in registers; 3x3 convolutions read a zero-bordered float4 plane in LDS.
// neural_network_vision_model.py:41-515 through muzero_model.py:802-909.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/smz.h"
#include "smz_mlp_device.hpp"

namespace smz_vision {
using namespace smz_mlp;

constexpr int kC = 3, kN = 7, kPix = kN * kN, kFlat = kC * kPix, kFlat4 = 148, kPad = kN + 2;
constexpr int kFrame = 98;
constexpr int kSmallMax = 1280;   // floats of convolution / batch-norm / 1x1 pieces of the four recurrent nets (1160 used)

// -------------------------------------------------------------------------------------------------------------------
// wave-per-leaf pieces
// -------------------------------------------------------------------------------------------------------------------
struct WaveLds {
    float4 plane[kPad * kPad];   // zero border, interior written per layer
    float flat[kFlat4];          // flattened 1x1-conv output, channel major (torch's Flatten of [3,7,7])
    float hid[2][64];            // tower activations (ping-pong)
};

__device__ inline const float *uniform_ptr(const float *base, int off) {
    return base + __builtin_amdgcn_readfirstlane(off);
}

// out[oc] = sum_{tap, ic} w[oc][ic][tap] * plane[pixel + tap][ic]; CIN = 3 or 4 (4th = action plane).
// Association (round 3, every variant below rounds alike): each ROW of taps is one fma chain from zero (taps left to right,
// input channels inside a tap), the three row sums are combined as (r0 + r1) + r2.  Three independent chains of 9 or 12 steps
// instead of one of 27 or 36: on the matrix cores (conv3x3_m) a dependent v_mfma_f32_4x4x1 step costs ~2x the issue interval
// of an independent one, and the vector-unit versions lose their s_nop wait states the same way.
template <int CIN>
__device__ inline void conv3x3(const float4 *plane, int pp, const float *__restrict__ w, float (&out)[kC]) {
    float r[3][kC];
#pragma unroll
    for (int ty = 0; ty < 3; ty++)
#pragma unroll
        for (int oc = 0; oc < kC; oc++) r[ty][oc] = 0.f;
    {
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const float4 v = plane[pp + (t / 3 - 1) * kPad + (t % 3 - 1)];
#pragma unroll
            for (int oc = 0; oc < kC; oc++) {
                r[t / 3][oc] = fmaf(w[(oc * CIN + 0) * 9 + t], v.x, r[t / 3][oc]);
                r[t / 3][oc] = fmaf(w[(oc * CIN + 1) * 9 + t], v.y, r[t / 3][oc]);
                r[t / 3][oc] = fmaf(w[(oc * CIN + 2) * 9 + t], v.z, r[t / 3][oc]);
                if (CIN == 4) r[t / 3][oc] = fmaf(w[(oc * CIN + 3) * 9 + t], v.w, r[t / 3][oc]);
            }
        }
    }
#pragma unroll
    for (int oc = 0; oc < kC; oc++) out[oc] = (r[0][oc] + r[1][oc]) + r[2][oc];
}

__device__ inline void bn_relu_store(float4 *plane, int pp, bool active, const float (&t)[kC], const float *__restrict__ bn) {
    float u[kC];
#pragma unroll
    for (int c = 0; c < kC; c++) u[c] = fmaxf(t[c] * bn[c] + bn[kC + c], 0.f);
    if (active) plane[pp] = make_float4(u[0], u[1], u[2], 0.f);
    lds_sync();
}

// v2 residual block with ONE batch-norm and convA used twice (neural_network_vision_model.py:41-79)
__device__ inline void residual_block(float4 *plane, int pp, bool active, const float *__restrict__ wa,
                                      const float *__restrict__ wb, const float *__restrict__ bn, float (&t)[kC]) {
    float c[kC];
    bn_relu_store(plane, pp, active, t, bn);
    conv3x3<kC>(plane, pp, wa, c);
    lds_sync();                               // every lane has read its neighbours before the plane is rewritten
    bn_relu_store(plane, pp, active, c, bn);
    conv3x3<kC>(plane, pp, wb, c);
    lds_sync();
    bn_relu_store(plane, pp, active, c, bn);
    conv3x3<kC>(plane, pp, wa, c);
    lds_sync();
#pragma unroll
    for (int k = 0; k < kC; k++) t[k] = c[k] + t[k];
}

// The same convolutions on ROW-OF-TAPS-MAJOR weights.  The search kernel's four wavefronts share one LDS, and a convolution
// is bound by the LDS cycles of its reads (a ds_read_b128 costs 4, a ds_read_b96 8: MI355X_MICROARCH.md, LDS), not by its
// multiply-adds -- so the weights of one row of taps are packed densely, 9 * CIN floats in 7 (CIN = 3) or 9 (CIN = 4)
// 16-byte reads: for j = tx * CIN + ic, row[2j], row[2j + 1] = the weights of output channels 0 and 1 (an even-aligned pair
// for v_pk_fma_f32, the input value in both halves), row[6 * CIN + j] = output channel 2 (v_fma_f32).  Rows are 36 floats
// apart.  Operation order per output is unchanged (taps 0..8, input channels 0..CIN-1 inside a tap): bit-identical to conv3x3.
constexpr int kTapRow = 36, kTapFloats = 3 * kTapRow;
__device__ inline void keep_f4(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }   // all four lanes of the
// vector stay live, so the read stays a ds_read_b128
template <int CIN>
__device__ inline void conv3x3_t(const float4 *plane, int pp, const float4 *__restrict__ wt, float (&out)[kC]) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int NJ = 3 * CIN, NQ = (9 * CIN + 3) / 4;
    v2f r01[3];
    float r2[3];
#pragma unroll
    for (int ty = 0; ty < 3; ty++) {   // (one chain per row of taps, combined as (r0 + r1) + r2: conv3x3's association)
        v2f o01 = {0.f, 0.f};
        float o2 = 0.f;
        float4 v[3], wq[NQ];
#pragma unroll
        for (int tx = 0; tx < 3; tx++) v[tx] = plane[pp + (ty - 1) * kPad + (tx - 1)];
#pragma unroll
        for (int q = 0; q < NQ; q++) wq[q] = wt[ty * (kTapRow / 4) + q];
#pragma unroll
        for (int tx = 0; tx < 3; tx++) keep_f4(v[tx]);
        keep_f4(wq[NQ - 1]);
        float w[4 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) { w[4 * q] = wq[q].x; w[4 * q + 1] = wq[q].y; w[4 * q + 2] = wq[q].z; w[4 * q + 3] = wq[q].w; }
#pragma unroll
        for (int tx = 0; tx < 3; tx++) {
            const float vin[4] = {v[tx].x, v[tx].y, v[tx].z, v[tx].w};
#pragma unroll
            for (int ic = 0; ic < CIN; ic++) {
                const int j = tx * CIN + ic;
                const v2f wp = {w[2 * j], w[2 * j + 1]}, vv = {vin[ic], vin[ic]};
                o01 = __builtin_elementwise_fma(wp, vv, o01);
                o2 = fmaf(w[2 * NJ + j], vin[ic], o2);
            }
        }
        r01[ty] = o01;
        r2[ty] = o2;
    }
    out[0] = (r01[0].x + r01[1].x) + r01[2].x;
    out[1] = (r01[0].y + r01[1].y) + r01[2].y;
    out[2] = (r2[0] + r2[1]) + r2[2];
}
// packed copy of a [3][CIN][9] convolution weight piece (kTapFloats floats) in that layout
template <int CIN>
__device__ inline void tap_major(float *dst, const float *src, int tid, int nthreads) {
    for (int i = tid; i < kTapFloats; i += nthreads) {
        const int ty = i / kTapRow, r = i % kTapRow;
        int j = -1, oc = 0;
        if (r < 6 * CIN) { j = r >> 1; oc = r & 1; }
        else if (r < 9 * CIN) { j = r - 6 * CIN; oc = 2; }
        dst[i] = j >= 0 ? src[(oc * CIN + j % CIN) * 9 + ty * 3 + j / CIN] : 0.f;
    }
}
__device__ inline void residual_block_t(float4 *plane, int pp, bool active, const float4 *__restrict__ wa,
                                        const float4 *__restrict__ wb, const float *__restrict__ bn, float (&t)[kC]) {
    float c[kC];
    bn_relu_store(plane, pp, active, t, bn);
    conv3x3_t<kC>(plane, pp, wa, c);
    lds_sync();
    bn_relu_store(plane, pp, active, c, bn);
    conv3x3_t<kC>(plane, pp, wb, c);
    lds_sync();
    bn_relu_store(plane, pp, active, c, bn);
    conv3x3_t<kC>(plane, pp, wa, c);
    lds_sync();
#pragma unroll
    for (int k = 0; k < kC; k++) t[k] = c[k] + t[k];
}

// The same convolutions on the MATRIX CORES (round 3), for the single-launch vision search where the pixel <-> lane layout is
// exactly a v_mfma_f32_4x4x1_16B_f32 operand layout: block b = pixels 4b .. 4b+3 (B operand: the lane's own tap value), rows =
// output channels 0..2 (+ an idle fourth row; A operand: the weight of channel lane % 4, the same in every block), D = four
// registers per lane = the lane's pixel, channels 0..3.  One instruction per (tap, input channel) instead of three multiply-adds,
// one 16-byte LDS read per four weights instead of 21 per layer, and an f32-input MFMA is one fused multiply-add with a single
// rounding per step (tools/mfma4_probe.hip) -- per row of taps one chain (taps left to right, input channels inside a tap), rows
// combined as (r0 + r1) + r2, as conv3x3: bit-identical.
// Weight table of a piece in LDS: wm[4][kMmRow]: wm[i][t * CIN + ic] = w[(i * CIN + ic) * 9 + t] (i < 3), zeros in row 3.
constexpr int kMmRow = 36, kMmFloats = 4 * kMmRow;
template <int CIN>
__device__ inline void mfma_table(float *dst, const float *src, int tid, int nthreads) {
    for (int i = tid; i < kMmFloats; i += nthreads) {
        const int row = i / kMmRow, k = i % kMmRow, t = k / CIN, ic = k % CIN;
        dst[i] = (row < kC && k < 9 * CIN) ? src[(row * CIN + ic) * 9 + t] : 0.f;
    }
}
template <int CIN>
__device__ inline void conv3x3_m(const float4 *plane, int pp, const float *__restrict__ wm, int lane, float (&out)[kC]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int K = 9 * CIN, NQ = (K + 3) / 4;
    const float4 *wr = reinterpret_cast<const float4 *>(wm + (lane & 3) * kMmRow);
    float4 wq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) wq[q] = wr[q];
    float4 v[9];
#pragma unroll
    for (int t = 0; t < 9; t++) v[t] = plane[pp + (t / 3 - 1) * kPad + (t % 3 - 1)];
    float w[4 * NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { w[4 * q] = wq[q].x; w[4 * q + 1] = wq[q].y; w[4 * q + 2] = wq[q].z; w[4 * q + 3] = wq[q].w; }
    v4f acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // one chain per row of taps
#pragma unroll
    for (int tx = 0; tx < 3; tx++)
#pragma unroll
        for (int ic = 0; ic < CIN; ic++)
#pragma unroll
            for (int ty = 0; ty < 3; ty++) {       // (the three chains advance in turn: no step waits for its own predecessor)
                const int t = ty * 3 + tx;
                const float vin[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
                acc[ty] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[t * CIN + ic], vin[ic], acc[ty], 0, 0, 0);
            }
#pragma unroll
    for (int oc = 0; oc < kC; oc++) out[oc] = (acc[0][oc] + acc[1][oc]) + acc[2][oc];
}
__device__ inline void residual_block_m(float4 *plane, int pp, bool active, const float *__restrict__ wa,
                                        const float *__restrict__ wb, const float *__restrict__ bn, int lane, float (&t)[kC]) {
    float c[kC];
    bn_relu_store(plane, pp, active, t, bn);
    conv3x3_m<kC>(plane, pp, wa, lane, c);
    lds_sync();
    bn_relu_store(plane, pp, active, c, bn);
    conv3x3_m<kC>(plane, pp, wb, lane, c);
    lds_sync();
    bn_relu_store(plane, pp, active, c, bn);
    conv3x3_m<kC>(plane, pp, wa, lane, c);
    lds_sync();
#pragma unroll
    for (int k = 0; k < kC; k++) t[k] = c[k] + t[k];
}

// per-pixel min-max scaling across the channels (scale_to_bound_action with dim=1 on [B,3,7,7]: :494-503)
__device__ inline void scale_channels(float (&t)[kC]) {
    const float mn = fminf(fminf(t[0], t[1]), t[2]), mx = fmaxf(fmaxf(t[0], t[1]), t[2]);
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;
#pragma unroll
    for (int c = 0; c < kC; c++) t[c] = (t[c] - mn) / sc;
}

// 1x1 convolution (with bias) of CIN input channels -> flattened [3*49] activations in LDS
template <int CIN>
__device__ inline void mix_to_flat(float *flat, int p, bool active, const float (&x)[4], const float *__restrict__ w,
                                   const float *__restrict__ b) {
#pragma unroll
    for (int oc = 0; oc < kC; oc++) {
        float s = 0.f;
#pragma unroll
        for (int ic = 0; ic < CIN; ic++) s = fmaf(w[oc * CIN + ic], x[ic], s);
        s += b[oc];
        if (active) flat[oc * kPix + p] = s;
    }
    lds_sync();
}

// acc = bias[lane] + sum_k W[k][lane] * act[k], summed as FOUR k-ordered fma chains over contiguous quarters of the inputs
// (quarter length = the number of 4-input groups divided by four, rounded up: 10 + 10 + 10 + 7 groups for the 147-input
// layer, 4 x 4 groups for a 64-wide one), chain 0 starting from the bias, the others from zero, combined as
// (c0 + c1) + (c2 + c3).  This is rounding for rounding what the single-launch vision search (smz_vision_search.hip)
// computes on the matrix cores -- one v_mfma_f32_4x4x1_16B_f32 per input and quarter, an f32-input MFMA being an fma with a
// single rounding (measured on MI355X: tools/mfma4_probe.hip) -- so the wave-per-leaf kernel here and that kernel give
// bit-identical outputs.  The weights live in global memory / L2: the 16-byte loads of CH groups of every quarter are
// issued back to back before the first multiply-add, so a layer costs one or two L2 round trips.
template <int CH>
__device__ inline float dense_stream(const float *__restrict__ W, const float *__restrict__ bias, const float *act, int K4,
                                     int OP, int lane) {
    const float4 *w4 = reinterpret_cast<const float4 *>(W) + lane;
    const float4 *a4 = reinterpret_cast<const float4 *>(act);
    const int n = K4 >> 2, pl = (n + 3) >> 2;
    float c[4] = {bias[lane], 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < pl; q0 += CH) {
        float4 w[4][CH];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < CH; j++) w[i][j] = w4[(size_t)min(i * pl + q0 + j, n - 1) * OP];
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int gk = i * pl + q0 + j;
                if (q0 + j < pl && gk < n) {
                    const float4 a = a4[gk];
                    c[i] = fmaf(w[i][j].x, a.x, c[i]);
                    c[i] = fmaf(w[i][j].y, a.y, c[i]);
                    c[i] = fmaf(w[i][j].z, a.z, c[i]);
                    c[i] = fmaf(w[i][j].w, a.w, c[i]);
                }
            }
    }
    return (c[0] + c[1]) + (c[2] + c[3]);
}

// Linear(147,H) relu [Linear(H,H) relu] x L Linear(H,n_out): off[0..5] = W1,b1,Wm,bm,Wo,bo (float offsets)
__device__ inline void tower(const float *weights, const int32_t *off, WaveLds &l, const smz_vision_desc &d, int lane,
                             float (&acc)[1][1]) {
    const int K4h = up4(d.H);
    float y = dense_stream<5>(weights + off[0], weights + off[1], l.flat, kFlat4, d.OP, lane);
    int cur = 0;
    l.hid[0][lane] = lane < d.H ? fmaxf(y, 0.f) : 0.f;
    lds_sync();
    for (int i = 0; i < d.L; i++) {
        y = dense_stream<4>(weights + off[2], weights + off[3], l.hid[cur], K4h, d.OP, lane);
        cur ^= 1;
        l.hid[cur][lane] = lane < d.H ? fmaxf(y, 0.f) : 0.f;
        lds_sync();
    }
    acc[0][0] = dense_stream<4>(weights + off[4], weights + off[5], l.hid[cur], K4h, d.OP, lane);
}

// prediction / afterstate prediction on the hidden state in t (registers): policy (softmax) to dst_policy, returns value
// `small`: where the convolution / batch-norm / 1x1 pieces are read from (the workgroup's LDS copy, or `weights`)
__device__ inline float predict(const float *weights, const float *small, const smz_vision_desc &d,
                                int net /*SMZ_V_PRE or SMZ_V_APR*/, WaveLds &l, int lane, int p, int pp, bool active,
                                float (&t)[kC], float *dst_policy, bool want_value) {
    const int32_t *o = d.off + SMZ_V_PRED_BASE + (net - SMZ_V_PRE) * SMZ_V_PRED_STRIDE;
    const float *wa = uniform_ptr(small, o[SMZ_VP_RES_A]), *wb = uniform_ptr(small, o[SMZ_VP_RES_B]);
    const float *bn = uniform_ptr(small, o[SMZ_VP_RES_BN]);
    for (int i = 0; i < d.L; i++) residual_block(l.plane, pp, active, wa, wb, bn, t);
    const float x[4] = {t[0], t[1], t[2], 0.f};
    float acc[1][1];
    float value = 0.f;
    if (want_value) {
        mix_to_flat<kC>(l.flat, p, active, x, uniform_ptr(small, o[SMZ_VP_VMIX_W]), uniform_ptr(small, o[SMZ_VP_VMIX_B]));
        tower(weights, o + SMZ_VP_VTOWER, l, d, lane, acc);
        value = decode_lanes<1>(acc[0], 0, d.S, lane);
        lds_sync();
    }
    mix_to_flat<kC>(l.flat, p, active, x, uniform_ptr(small, o[SMZ_VP_PMIX_W]), uniform_ptr(small, o[SMZ_VP_PMIX_B]));
    tower(weights, o + SMZ_VP_PTOWER, l, d, lane, acc);
    softmax_lanes<1>(acc[0], d.A, lane, dst_policy);
    return value;
}

__device__ inline void zero_wave_lds(WaveLds &l, int lane) {
    for (int i = lane; i < kPad * kPad; i += smz_mlp::kWave) l.plane[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane; i < kFlat4; i += smz_mlp::kWave) l.flat[i] = 0.f;
    lds_sync();
}

}  // namespace smz_vision
