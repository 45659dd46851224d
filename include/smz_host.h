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


/* Hand-off words of the shared control page (host_envs.py: one `done` word per worker, written by that worker only, read by
 * the parent): a release store after the worker's rows are written, an acquire load before the parent reads them -- the
 * ordering numpy stores cannot express (they are enough on x86's total store order, not on aarch64). */
void smzh_store_release_i32(int32_t *word, int32_t value);
int32_t smzh_load_acquire_i32(const int32_t *word);

#ifdef __cplusplus
}
#endif
#endif
