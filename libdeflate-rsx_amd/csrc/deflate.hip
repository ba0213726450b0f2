// deflate.hip -- batch compression (placeholder until the kernels land).
#include "engine.h"

size_t deflate_ws_bytes(size_t, uint64_t, uint64_t) { return 0; }

int deflate_launch(rsx_engine *e, int, int, const uint8_t *, const uint64_t *, size_t, uint64_t,
                   uint64_t, uint8_t *, const uint64_t *, uint64_t *, int32_t *)
{
    e->err = "compression kernels not built";
    return RSX_E_INVALID;
}
