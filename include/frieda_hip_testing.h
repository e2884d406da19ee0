/* frieda_hip_testing.h — test hooks of libfrieda_hip.so.  NOT part of the drop-in boundary (include/frieda_hip.h): nothing here
 * replaces a reference interface; the parity tests use it to force branches that the protocol reaches once in ~10^8 draws. */
#ifndef FRIEDA_HIP_TESTING_H
#define FRIEDA_HIP_TESTING_H

#include "frieda_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The acceptance bound of Channel::draw_felt (stwo: every drawn word must be < 2P, else redraw — a ~4e-9 event).  A lower bound
 * makes the redraw branch fire on most draws so that it can be compared against the oracle (oracle/: fo_test_set_draw_bound); proofs
 * made with a non-default bound do not verify.  0 restores 2P. */
int frieda_ctx_test_set_draw_bound(frieda_ctx* ctx, uint32_t bound);

#ifdef __cplusplus
}
#endif

#endif
