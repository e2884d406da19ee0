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

/* The proof-of-work search scans ordered nonce windows (2^20 nonces first, then doubling).  log_first in 8 .. 40 makes the first
 * window 2^log_first nonces, so that a test reaches the window-growth branch at a few bits of work; 0 restores the default.  The
 * nonce found is the minimum either way. */
int frieda_ctx_test_set_grind_first_log(frieda_ctx* ctx, uint32_t log_first);

/* A device with less memory than this one: workspace requests above `bytes` fail with FRIEDA_ERR_NOMEM (0 = no limit).  For the
 * tests of frieda_prove_many / frieda_commit_many's retry with smaller calls. */
int frieda_ctx_test_set_arena_limit(frieda_ctx* ctx, uint64_t bytes);

/* The parser of sysfs CPU lists ("0-3,8,10-11\n") behind frieda_multi's NUMA placement, for the CPU tests: *n receives the count,
 * out_cpus (cap entries) the CPUs in order.  FRIEDA_ERR_FORMAT for malformed text, FRIEDA_ERR_ARG when cap is too small. */
int frieda_test_parse_cpulist(const char* text, int* out_cpus, size_t cap, size_t* n);

/* The NUMA placement rule of frieda_multi's worker threads on a sysfs tree of the caller's making (no GPU involved): the CPUs of
 * <sysfs_root>/devices/system/node/node<k>/cpulist, k = <sysfs_root>/bus/pci/devices/<pci_bus_id>/numa_node, intersected with the
 * process's affinity; *n = 0 when the tree says nothing (missing files, node -1, malformed list). */
int frieda_test_near_cpus(const char* sysfs_root, const char* pci_bus_id, int* out_cpus, size_t cap, size_t* n);

#ifdef __cplusplus
}
#endif

#endif
