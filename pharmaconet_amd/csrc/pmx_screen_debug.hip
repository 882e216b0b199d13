// pmx_screen_debug.hip - pmx_screen.hip once more, as namespace pmx_dbg, with every PMX_TREE_FLAGS switch compiled in: the kernels
// behind the validation settings of the tests (no bounds, no fused levels, no cache, term-by-term items, ...) and the phase studies
// of tools/. libpmx's own kernels (namespace pmx) carry none of these switches; see the note at the top of pmx_screen.hip.
#include <hip/hip_runtime.h>
#include <cstring>

#define PMX_DEBUG_KERNELS 1
#define PMX_NS pmx_dbg
#include "pmx_screen.hip"
#include "pmx_debug.h"

namespace pmx_debug {

template <int G>
static void ligand_g(bool exact, bool tails, unsigned blocks, unsigned lds, hipStream_t stream, const pmx_dbg::ScreenParams &p) {
    if (exact) pmx_dbg::ligand_kernel<G, true, false><<<dim3(blocks), dim3(64), lds, stream>>>(p);
    else if (tails) pmx_dbg::ligand_kernel<G, false, true><<<dim3(blocks), dim3(64), lds, stream>>>(p);
    else pmx_dbg::ligand_kernel<G, false, false><<<dim3(blocks), dim3(64), lds, stream>>>(p);
}

bool launch_ligand(int G, bool exact, bool tails, unsigned blocks, unsigned lds, hipStream_t stream, const void *params, size_t bytes) {
    if (bytes != sizeof(pmx_dbg::ScreenParams)) return false;
    pmx_dbg::ScreenParams p;
    std::memcpy(&p, params, sizeof p);
    switch (G) {
    case 1: ligand_g<1>(exact, tails, blocks, lds, stream, p); break;
    case 2: ligand_g<2>(exact, tails, blocks, lds, stream, p); break;
    case 4: ligand_g<4>(exact, tails, blocks, lds, stream, p); break;
    case 8: ligand_g<8>(exact, tails, blocks, lds, stream, p); break;
    case 16: ligand_g<16>(exact, tails, blocks, lds, stream, p); break;
    case 32: ligand_g<32>(exact, tails, blocks, lds, stream, p); break;
    case 64: ligand_g<64>(exact, tails, blocks, lds, stream, p); break;
    default: return false;
    }
    return true;
}

bool launch_task(int G, unsigned blocks, unsigned lds, hipStream_t stream, const void *params, size_t bytes) {
    if (bytes != sizeof(pmx_dbg::ScreenParams)) return false;
    pmx_dbg::ScreenParams p;
    std::memcpy(&p, params, sizeof p);
    switch (G) {
    case 1: pmx_dbg::task_kernel<1><<<dim3(blocks), dim3(64), lds, stream>>>(p); break;
    case 2: pmx_dbg::task_kernel<2><<<dim3(blocks), dim3(64), lds, stream>>>(p); break;
    case 4: pmx_dbg::task_kernel<4><<<dim3(blocks), dim3(64), lds, stream>>>(p); break;
    case 8: pmx_dbg::task_kernel<8><<<dim3(blocks), dim3(64), lds, stream>>>(p); break;
    case 16: pmx_dbg::task_kernel<16><<<dim3(blocks), dim3(64), lds, stream>>>(p); break;
    case 32: pmx_dbg::task_kernel<32><<<dim3(blocks), dim3(64), lds, stream>>>(p); break;
    case 64: pmx_dbg::task_kernel<64><<<dim3(blocks), dim3(64), lds, stream>>>(p); break;
    default: return false;
    }
    return true;
}

} // namespace pmx_debug
