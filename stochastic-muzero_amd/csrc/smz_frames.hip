// smz_frames.hip -- frame ingest of the `vision_model` family (C ABI: smz_frames_resize_u8).
//
// What it replaces: Game.transform_rgb of the reference (game.py:82-89, applied to every rendered frame at
// game.py:105-107 / 142-143): uint8 H x W x 3 frame -> ToTensor (CHW, / 255) -> torchvision Resize(shape) -> [1,3,h,w],
// one frame at a time on the CPU.  Here all B frames of an env step are resized by one launch on the engine's stream, from
// the uint8 frames a host environment uploaded through pinned memory, straight into the [B,3,98,98] float32 tensor that
// smz_vision_initial reads.
//
// Arithmetic = ATen's upsample_bilinear2d (align_corners = False, no antialias: what torchvision 0.14's Resize calls for
// tensors), float32 throughout:
//   scale = in / out;  src = fma(scale, dst + 0.5, -0.5), clamped at 0;  i0 = min(int(src), in - 1);  i1 = min(i0 + 1, in - 1);
//   l1 = clamp(src - i0, 0, 1), l0 = 1 - l1;  row_y = fma(v_y0, l0x, v_y1 * l1x);  out = fma(row_0, l0y, row_1 * l1y)
//   (UpSampleKernel.cpp's Interpolate<n>::eval: `out = t0 * w0; out += t1 * w1`; which product is fused is ATen's compiler's
//   choice -- this is the form torch 2.10 (AVX-512 CPU dispatch) computes for 98-wide outputs; any other choice is within 1 ulp),
//   v = uint8 / 255 (IEEE division, a 256-entry table in LDS).
// torchvision / gymnasium are not part of this build, so parity is unpinned against torchvision itself; the test holds the
// kernel to torch's own CPU interpolate on the same frames (tests/test_gpu_frames.py).
//
// Memory shape (HBM-bound, no reuse across blocks): one 128-thread workgroup per (frame, output row); the two source rows it
// needs are fetched as coalesced dwords into LDS (a bilinear tap pattern touches every 64-byte line of those rows anyway), then
// one thread per output pixel blends its 4 taps x 3 channels from LDS and writes the three planes -- coalesced 4-byte stores
// along x.  Algorithmic bytes per frame: out_h * 2 * W * 3 (rows read; rows shared by neighbouring output rows hit in L2) +
// 3 * out_h * out_w * 4 written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/smz.h"

extern thread_local char smz_g_err[512];     // the library's last-error text (smz_kernels.hip)

namespace {

extern __shared__ uint32_t smz_frames_lds[];

__device__ inline void src_index(int dst, float scale, int in, int &i0, int &i1, float &l0, float &l1) {
    float src = fmaf(scale, (float)dst + 0.5f, -0.5f);     // (ATen's build contracts scale * (dst + 0.5) - 0.5 into one fma)
    if (src < 0.f) src = 0.f;
    i0 = min((int)src, in - 1);
    i1 = min(i0 + 1, in - 1);
    l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
}

// frames [n][H][W][3] u8 -> out[row(i)][3][OH][OW] f32, row(i) = rows ? rows[i] : i
__global__ void __launch_bounds__(128) k_frames_resize_u8(const uint8_t *frames, int H, int W, int OH, int OW,
                                                          const int32_t *rows, float *out, int row_words) {
    const int f = blockIdx.x / OH, oy = blockIdx.x % OH;
    float *lut = reinterpret_cast<float *>(smz_frames_lds);                    // [256] u8 -> float / 255
    uint32_t *line = smz_frames_lds + 256;                                      // [2][row_words] the two source rows
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lut[i] = (float)i / 255.0f;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    int y0, y1;
    float ly0, ly1;
    src_index(oy, sy, H, y0, y1, ly0, ly1);
    const size_t row_bytes = (size_t)W * 3;
    const uint8_t *base = frames + (size_t)f * H * row_bytes;
    int skew[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const size_t start = (size_t)reinterpret_cast<uintptr_t>(base) + (size_t)(r ? y1 : y0) * row_bytes;   // byte address
        const size_t a0 = start & ~(size_t)3;
        skew[r] = (int)(start - a0);
        const int n = (int)((start + row_bytes - a0 + 3) >> 2);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a0);
        for (int i = threadIdx.x; i < n; i += blockDim.x) line[r * row_words + i] = src[i];
    }
    __syncthreads();
    const uint8_t *b0 = reinterpret_cast<const uint8_t *>(line) + skew[0];
    const uint8_t *b1 = reinterpret_cast<const uint8_t *>(line + row_words) + skew[1];
    const int orow = rows ? rows[f] : f;
    float *dst = out + (size_t)orow * 3 * OH * OW + (size_t)oy * OW;
    for (int ox = threadIdx.x; ox < OW; ox += blockDim.x) {
        int x0, x1;
        float lx0, lx1;
        src_index(ox, sx, W, x0, x1, lx0, lx1);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v00 = lut[b0[x0 * 3 + c]], v01 = lut[b0[x1 * 3 + c]];
            const float v10 = lut[b1[x0 * 3 + c]], v11 = lut[b1[x1 * 3 + c]];
            // ATen's generic linear kernel: `out = t0 * w0; out += t1 * w1` per dimension, x inside y.  Its optimised build fuses
            // one product of each line into an fma and leaves the association to the compiler; the form below is what
            // torch 2.10's CPU kernel (AVX-512 dispatch) produces for 98-pixel-wide outputs, bit for bit (tests/test_gpu_frames.py)
            const float top = fmaf(v00, lx0, v01 * lx1), bot = fmaf(v10, lx0, v11 * lx1);
            dst[(size_t)c * OH * OW + ox] = fmaf(top, ly0, bot * ly1);
        }
    }
}

// The same resize from TAP-COMPACTED frames: taps [n][2 OH][2 OW][3] u8 holds, of each H x W frame, only the pixels the resize
// reads -- row 2 oy + r = source row y_r(oy), column 2 ox + q = source column x_q(ox) (host_envs.tap_index: src_index's own
// arithmetic) -- 115 KB instead of 720 KB per 400 x 600 frame over PCIe.  Weights come from H and W exactly as above and the
// blend is the same three fmaf lines, so the output is bit-identical to k_frames_resize_u8 on the full frame.
__global__ void __launch_bounds__(128) k_frames_resize_taps_u8(const uint8_t *taps, int H, int W, int OH, int OW,
                                                               const int32_t *rows, float *out) {
    const int f = blockIdx.x / OH, oy = blockIdx.x % OH;
    float *lut = reinterpret_cast<float *>(smz_frames_lds);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lut[i] = (float)i / 255.0f;
    __syncthreads();
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    int y0, y1;
    float ly0, ly1;
    src_index(oy, sy, H, y0, y1, ly0, ly1);
    const size_t trow = (size_t)2 * OW * 3;
    const uint8_t *b0 = taps + ((size_t)f * 2 * OH + 2 * oy) * trow, *b1 = b0 + trow;
    const int orow = rows ? rows[f] : f;
    float *dst = out + (size_t)orow * 3 * OH * OW + (size_t)oy * OW;
    for (int ox = threadIdx.x; ox < OW; ox += blockDim.x) {
        int x0, x1;
        float lx0, lx1;
        src_index(ox, sx, W, x0, x1, lx0, lx1);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v00 = lut[b0[(2 * ox) * 3 + c]], v01 = lut[b0[(2 * ox + 1) * 3 + c]];
            const float v10 = lut[b1[(2 * ox) * 3 + c]], v11 = lut[b1[(2 * ox + 1) * 3 + c]];
            const float top = fmaf(v00, lx0, v01 * lx1), bot = fmaf(v10, lx0, v11 * lx1);
            dst[(size_t)c * OH * OW + ox] = fmaf(top, ly0, bot * ly1);
        }
    }
}

int fail(int code, const char *text) {
    snprintf(smz_g_err, sizeof(smz_g_err), "%s", text);
    return code;
}

}  // namespace

extern "C" {

int smz_frames_resize_u8(const uint8_t *frames_dev, int n_frames, int H, int W, int out_h, int out_w, const int32_t *rows_dev,
                         float *out_dev, smz_stream stream) {
    if (!frames_dev || !out_dev || n_frames < 1 || H < 1 || W < 1 || out_h < 1 || out_w < 1)
        return fail(SMZ_ERR_INVALID, "smz_frames_resize_u8: bad argument");
    const int row_words = (int)(((size_t)W * 3 + 3 + 3) / 4) + 1;          // a row plus the skew of an unaligned start
    const size_t lds = (256 + 2 * (size_t)row_words) * 4;
    if (lds > 64 * 1024) return fail(SMZ_ERR_TOO_LARGE, "smz_frames_resize_u8: frame rows wider than 10 000 pixels");
    if ((size_t)n_frames * out_h > 0x7fffffffu) return fail(SMZ_ERR_TOO_LARGE, "smz_frames_resize_u8: too many frames for one launch");
    hipLaunchKernelGGL(k_frames_resize_u8, dim3((unsigned)(n_frames * out_h)), dim3(128), lds, (hipStream_t)stream, frames_dev,
                       H, W, out_h, out_w, rows_dev, out_dev, row_words);
    return hipGetLastError() == hipSuccess ? SMZ_OK : fail(SMZ_ERR_HIP, "smz_frames_resize_u8: launch failed");
}

int smz_frames_resize_taps_u8(const uint8_t *taps_dev, int n_frames, int H, int W, int out_h, int out_w, const int32_t *rows_dev,
                              float *out_dev, smz_stream stream) {
    if (!taps_dev || !out_dev || n_frames < 1 || H < 1 || W < 1 || out_h < 1 || out_w < 1)
        return fail(SMZ_ERR_INVALID, "smz_frames_resize_taps_u8: bad argument");
    if ((size_t)n_frames * out_h > 0x7fffffffu) return fail(SMZ_ERR_TOO_LARGE, "smz_frames_resize_taps_u8: too many frames for one launch");
    hipLaunchKernelGGL(k_frames_resize_taps_u8, dim3((unsigned)(n_frames * out_h)), dim3(128), 256 * 4, (hipStream_t)stream,
                       taps_dev, H, W, out_h, out_w, rows_dev, out_dev);
    return hipGetLastError() == hipSuccess ? SMZ_OK : fail(SMZ_ERR_HIP, "smz_frames_resize_taps_u8: launch failed");
}

// ---- host-buffer boundary: page-locking a caller's shared mapping and copies on the caller's stream -------------------------
int smz_host_register(void *host_ptr, size_t bytes) {
    if (!host_ptr || !bytes) return fail(SMZ_ERR_INVALID, "smz_host_register: bad argument");
    const hipError_t e = hipHostRegister(host_ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        snprintf(smz_g_err, sizeof(smz_g_err), "smz_host_register: hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e));
        return SMZ_ERR_HIP;
    }
    return SMZ_OK;
}

int smz_host_unregister(void *host_ptr) {
    if (!host_ptr) return fail(SMZ_ERR_INVALID, "smz_host_unregister: bad argument");
    if (hipHostUnregister(host_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(SMZ_ERR_HIP, "smz_host_unregister: hipHostUnregister failed");
    }
    return SMZ_OK;
}

int smz_copy_async(void *dst, const void *src, size_t bytes, int to_device, smz_stream stream) {
    if (!dst || !src) return fail(SMZ_ERR_INVALID, "smz_copy_async: bad argument");
    if (!bytes) return SMZ_OK;
    if (hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(SMZ_ERR_HIP, "smz_copy_async: hipMemcpyAsync failed");
    }
    return SMZ_OK;
}

}  // extern "C"
