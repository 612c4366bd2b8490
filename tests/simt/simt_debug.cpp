// Test-only window into the library's device-side layouts under the SIMT emulator ("device" memory is host memory there): the paced
// column-panel stream of a matrix, so that tools/l2_model's restatement of the layout can be compared with what the LIBRARY builds,
// entry for entry (tests/test_l2_model_host.py).  Compiled into the emulator library only; the product has no such export.
#include "pkg/csrc/sl_internal.hpp"

extern "C" int simt_debug_pw_layout(const sl_matrix *m, const uint32_t **idx, const uint32_t **tile_ptr, uint64_t out[8])
{
    if (!m || !m->d_pw_idx) return 0;
    *idx = m->d_pw_idx; *tile_ptr = m->d_pw_tile_ptr;
    out[0] = m->n_pw_tiles; out[1] = m->pw_chunks; out[2] = m->pw_rpw; out[3] = m->pw_blocks; out[4] = m->pw_deal; out[5] = m->pw_pbits;
    out[6] = m->pw_slack; out[7] = m->pw_xcd;
    return 1;
}
