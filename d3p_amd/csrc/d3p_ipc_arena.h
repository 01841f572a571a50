// One hipIpc ARENA per process for the inboxes of the exchange (d3p_xchg_*) and full-mesh (d3p_fmesh_*) communicators (round 6).
//
// Rounds 2 - 5 gave every communicator its own uncached allocation, exported it (hipIpcGetMemHandle), had every peer import it
// (hipIpcOpenMemHandle) and undid all of that at teardown.  With dmabuf IPC that cycle is not reliable when it repeats inside one
// process: tools/soak_teardown.py (4 processes, a mesh and an exchange created, used and closed per round) met, after 10 - 35
// rounds, hipIpcGetMemHandle failing with "invalid argument" on a fresh allocation -- and, when the export was retried on another
// allocation, peers that mapped something else than the new inbox (wrong sums, waits that ran out) -- although every rank had
// unmapped every peer before anybody freed anything (the collective teardown of this round).  A process without peers never
// failed (300 lifetimes).  So the cycle is not repeated: ONE allocation per process is exported ONCE, every peer imports it ONCE
// (kept until the process ends), and a communicator's inbox is a range of it -- the 80-byte handle a communicator hands to its
// peers is the arena's 64-byte hipIpc handle + the range's offset + a marker.  An inbox that does not fit (or D3P_IPC_ARENA_MB=0)
// takes the old path: its own allocation, its own export.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace d3p {

// D3P_IPC_HANDLE_BYTES (include/d3p_hip.h) = 80: 64 (hipIpcMemHandle_t) + 8 (offset into the arena) + 8 (1 = arena range, 0 = own allocation)
#include "../../include/d3p_hip.h"

struct IpcRange {
    char* ptr = nullptr;     // this process's address of the range
    size_t bytes = 0;
    bool in_arena = false;
};

// a zeroed, exported range of `bytes` (multiple of 256) and the 80-byte handle the peers map it with
int ipc_range_create(size_t bytes, IpcRange* out, uint8_t handle_out[D3P_IPC_HANDLE_BYTES], const char* who);
void ipc_range_destroy(IpcRange* r);
// a peer's range in this process (opened[] says whether ipc_peer_close has something to undo: own allocations only)
int ipc_peer_open(const uint8_t handle[D3P_IPC_HANDLE_BYTES], char** ptr_out, bool* opened_out, const char* who, int peer_rank);
void ipc_peer_close(char* ptr, bool opened);

}  // namespace d3p
