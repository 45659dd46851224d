/* smz_host.h -- C ABI of libsmzhost.so: host-side helpers of the host-environment boundary that the env worker
 * processes call.  Plain C, no GPU runtime: the workers of envs.HostVecEnv(workers=N) load this library and nothing else
 * native.  (The device side of the same boundary -- smz_frames_resize_taps_u8, smz_host_register, smz_copy_async -- is in
 * smz.h / libsmz.so.) */
#ifndef SMZ_HOST_H
#define SMZ_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int smzh_abi_version(void);
/* Tap compaction of one rendered frame (replaces, together with smz_frames_resize_taps_u8 on the device, the per-frame
 * ToTensor + Resize of Game.transform_rgb, game.py:82-89): frame [H][W][3] uint8 -> taps [n_rows][n_cols][3] uint8 with
 * taps[r][c] = frame[row_index[r]][col_index[c]].  row_index / col_index are the (i0, i1) pairs of the bilinear resize's
 * source-index rule (host_envs.tap_index: 2 * out_h and 2 * out_w entries).  Returns 0, or -1 for an index outside the frame. */
int smzh_gather_taps_u8(const uint8_t *frame, int H, int W, const int32_t *row_index, int n_rows, const int32_t *col_index,
                        int n_cols, uint8_t *taps);

#ifdef __cplusplus
}
#endif
#endif
