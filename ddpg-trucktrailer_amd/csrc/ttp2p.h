// ttp2p.h -- the peer-to-peer gradient exchange of data-parallel ranks (include/ttenv.h: tt_p2p_*): what the host object
// (csrc/ttp2p.hip) and the optimizer launch that reads the exchange (k_adam_soft_p2p, csrc/ttlearn.hip) share.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ttenv.h"

namespace ttp2p {

constexpr int MAXR = TT_P2P_MAX_RANKS, MAXS = TT_P2P_MAX_SITES;
constexpr size_t FLAG_BYTES = 4096;      // the flag block: arrive[MAXS][MAXR] ints (128 B) at its start

// kernel-side view of the exchange for ONE site
struct Args {
    int world, me, site;
    int *arrive[MAXR];                   // arrive[r]: the arrival words of rank r's flag block ([MAXS][MAXR] ints); r == me: local
    const float *grad[MAXR];             // grad[r]: this site's flat gradient buffer in rank r's gradient block
    unsigned tensor_offset[12];          // where each parameter tensor (tt_mlp_weights order) starts in that buffer, in floats
    int *gave_up_host;                   // one int of pinned host memory (system scope store on give-up)
    unsigned long long wait_ticks;       // bound of the wait in wall_clock64() ticks (100 MHz)
};

}  // namespace ttp2p

struct tt_p2p {
    int device, rank, world, sites;
    int numel[ttp2p::MAXS];
    size_t offset[ttp2p::MAXS];          // byte offset of site s's gradient buffer in a gradient block (the same in every rank's)
    size_t bytes;                        // of a gradient block
    // two allocations per rank, both opened by every peer through IPC handles:
    char *flags[ttp2p::MAXR];            // FINE-grained: peers store arrival words into it while this rank's kernels poll them
    char *block[ttp2p::MAXR];            // gradients.  [rank] = own allocation; [r] = rank r's
    bool attached[ttp2p::MAXR];
    bool fine_grads;                     // gradient block fine-grained as well (TT_P2P_FINE_GRADS=1; measurement aid)
    int *gave_up_host;
    unsigned long long wait_ticks;
    char err[256];
};
