// pmx_debug.h - launchers of the screening kernels compiled with every PMX_TREE_FLAGS switch live (pmx_screen_debug.hip).
// pmx_api.hip calls them when a bit outside PMX_PRODUCT_FLAGS is set; libpmx's own kernels read those bits as zero.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace pmx_debug {
// `params`: the caller's pmx::ScreenParams (same source, same layout; `bytes` is checked against this side's sizeof).
// G = lanes per slot (1 .. 64, a power of two). Returns false when G or the size is not one this side knows.
bool launch_ligand(int G, bool exact, bool tails, unsigned blocks, unsigned lds, hipStream_t stream, const void *params, size_t bytes);
bool launch_task(int G, unsigned blocks, unsigned lds, hipStream_t stream, const void *params, size_t bytes);
} // namespace pmx_debug
