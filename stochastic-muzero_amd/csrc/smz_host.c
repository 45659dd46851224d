/* libsmzhost.so -- see include/smz_host.h.  Built with gcc (no GPU toolchain): make -C stochastic-muzero_amd/csrc host */
#include "../../include/smz_host.h"

#include <string.h>

int smzh_abi_version(void) { return 2; }

void smzh_store_release_i32(int32_t *word, int32_t value) { __atomic_store_n(word, value, __ATOMIC_RELEASE); }
int32_t smzh_load_acquire_i32(const int32_t *word) { return __atomic_load_n(word, __ATOMIC_ACQUIRE); }

int smzh_gather_taps_u8(const uint8_t *frame, int H, int W, const int32_t *row_index, int n_rows, const int32_t *col_index,
                        int n_cols, uint8_t *taps) {
    if (!frame || !row_index || !col_index || !taps || H < 1 || W < 1 || n_rows < 0 || n_cols < 0) return -1;
    for (int c = 0; c < n_cols; c++)
        if (col_index[c] < 0 || col_index[c] >= W) return -1;
    for (int r = 0; r < n_rows; r++) {
        const int y = row_index[r];
        if (y < 0 || y >= H) return -1;
        const uint8_t *src = frame + (size_t)y * W * 3;
        uint8_t *dst = taps + (size_t)r * n_cols * 3;
        /* a pixel moves as ONE 4-byte load / store (the 4th byte is overwritten by the next pixel); the last pixel of a row
         * moves byte by byte, so nothing is read past the frame or written past the row */
        int c = 0;
        for (; c + 1 < n_cols; c++) {
            const int x = col_index[c];
            if (x + 1 < W || y + 1 < H) {
                uint32_t v;
                memcpy(&v, src + (size_t)x * 3, 4);
                memcpy(dst + 3 * (size_t)c, &v, 4);
            } else {
                memcpy(dst + 3 * (size_t)c, src + (size_t)x * 3, 3);
            }
        }
        if (c < n_cols) memcpy(dst + 3 * (size_t)c, src + (size_t)col_index[c] * 3, 3);
    }
    return 0;
}
