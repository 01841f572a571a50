// Shared between d3p_fmesh.hip (the stand-alone collective) and d3p_vae.hip (the VAE step's fused form): the mesh object, the
// arguments of one all-reduce and the tagged store.  See d3p_fmesh.hip for the protocol.
#pragma once
#include "d3p_device.h"
#include "d3p_host.h"

namespace d3p {

#define D3P_FMESH_MAX_WORLD 16
#define D3P_FMESH_WGS 512   // two per CU (no LDS, 4 waves each): 131 072 threads, 5 elements of a 2.76 MB vector per thread (one rank, no peers:
                            // 12.7 us with 512 workgroups, 20.5 with 256, 37.5 with 128 -- the passes are latency, so more threads)
#define D3P_FMESH_WAIT_ROUNDS (1u << 24)   // polls of one word (~ 0.7 us each)

struct FMesh {
    int world, rank;
    uint64_t n;        // floats of the vector
    uint64_t chunk;    // ceil(n / world)
    unsigned long long epoch;
    int wgs;           // workgroups of a launch (d3p_fmesh_set_grid)
    char* inbox;       // [scatter: 2 x world x chunk words | gather: 2 x world x chunk words | status: 16 words]
    size_t inbox_bytes;
    char* peer[D3P_FMESH_MAX_WORLD];
    bool opened[D3P_FMESH_MAX_WORLD];
    bool in_arena;     // the inbox is a range of the process's hipIpc arena (d3p_ipc_arena.h)
};

inline size_t fmesh_region_words(int world, uint64_t chunk) { return (size_t)2 * world * chunk; }

struct FMeshArgs {
    float* buf;
    uint64_t n, chunk;
    int world, rank;
    unsigned parity;
    uint32_t tag;
    char* peer[D3P_FMESH_MAX_WORLD];
    size_t gather_off;      // bytes from the inbox's start to its gather region
    uint32_t* status;       // this rank's status word (in its own inbox)
};

__device__ __forceinline__ void fm_store(char* base, size_t word, float v, uint32_t tag)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(base) + word, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}


// the arguments of the mesh's NEXT all-reduce (advances its epoch)
inline void fmesh_next_args(FMesh* x, float* buf, FMeshArgs* a)
{
    memset(a, 0, sizeof(*a));
    a->buf = buf;
    a->n = x->n;
    a->chunk = x->chunk;
    a->world = x->world;
    a->rank = x->rank;
    const unsigned long long epoch = ++x->epoch;
    a->parity = (unsigned)(epoch & 1ull);
    a->tag = (uint32_t)epoch;
    for (int p = 0; p < x->world; ++p) a->peer[p] = x->peer[p];
    a->gather_off = fmesh_region_words(x->world, x->chunk) * sizeof(unsigned long long);
    a->status = reinterpret_cast<uint32_t*>(x->inbox + 2 * a->gather_off);
}

}  // namespace d3p
