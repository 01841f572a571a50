// Full-mesh sum-all-reduce of a float vector over the GPUs of one node (d3p_fmesh_*): the collective of the data-parallel VAE step.
//
// The reference is single-device; a data-parallel DPSVI.update needs ONE sum over the ranks per step
// (svi.py:343-346 is the only cross-example operation) -- for the VAE of examples/vae.py that is P + 2 = 688 886 floats, 2.76 MB.
// xGMI is point to point: every GPU has a link to each of its 7 peers.  A ring all-reduce takes 2 (n - 1) dependent hops and loads one
// link at a time; on a full mesh the same sum is TWO hops with every link busy:
//
//   reduce-scatter  rank r owns chunk r of the vector (ceil(n / world) elements).  Every rank stores its partial values of chunk o
//                   straight into rank o's inbox (slot [parity][sender]) -- 1 / world of the vector per link;
//   owner sum       the owner adds the world's partials of its chunk in RANK ORDER (one rank, one fixed order: every rank later
//                   receives bit for bit the same sum -- replicas stay identical without a broadcast of parameters);
//   all-gather      the owner stores the sums of its chunk into every peer's gather inbox -- again 1 / world of the vector per link.
//
// Arrival is signalled by the data: every float travels as ONE aligned 8-byte word {fp32 bits | 32-bit epoch tag} (performed whole;
// the low-latency protocol of the step exchange, d3p_logreg_chain.h).  Twice the bytes of a flag-after-data protocol, but no release
// fence -- on this chip a system-scope release is a write-back of the XCD's whole L2 -- and no second trip.  Slots are double-buffered
// by the parity of the epoch: a rank can only send epoch e + 2 after it finished epoch e + 1, which needs every peer's sends of
// e + 1, which those peers issue after they finished READING epoch e.
// All waits are bounded and raise the status word (a stopped collective leaves the vector undefined and is reported).
// One launch of grid-stride workgroups that must ALL be resident (a workgroup that has sent its share waits for the peers' words, some of
// which only come once the peers have received the words of this launch's other workgroups): 512 of 256 threads, two per CU, on a GPU
// of its own.  Ranks that SHARE a GPU (tests, rehearsals) must leave room for each other's kernels -- a compute kernel whose workgroup
// takes a CU's whole register file can never start on a GPU whose every CU holds a waiting workgroup of this launch:
// d3p_fmesh_set_grid (d3p_fmesh_connect_local sets 48).
#include "d3p_fmesh.h"
#include "d3p_ipc_arena.h"

#include <array>
#include <map>
#include <mutex>
#include <new>
#include <vector>

namespace d3p {

// bounded wait for word `word` of the own inbox to carry `tag`; false: the bound ran out or the collective was stopped
[[maybe_unused]] __device__ __forceinline__ bool fm_wait(const char* base, size_t word, uint32_t tag, uint32_t* status, float* out)
{
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(base) + word;
    for (uint32_t spins = 0; spins < D3P_FMESH_WAIT_ROUNDS; ++spins) {
        const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((uint32_t)(w >> 32) == tag) { *out = __uint_as_float((uint32_t)w); return true; }
        if ((spins & 255u) == 255u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    }
    return false;
}

__global__ void __launch_bounds__(256) k_fmesh_allreduce(FMeshArgs a)
{
    // Every loop keeps U independent memory operations of a thread in flight: a thread that loads, stores and polls one word at a time is
    // bound by the latency of each (46 us for 688 886 floats with NO peer at all; 8 us like this).
    constexpr int U = 4;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    char* const mine = a.peer[a.rank];
    const size_t sc_par = (size_t)a.parity * a.world * a.chunk;   // first word of this parity's slots in a region
    // ---- reduce-scatter: my partial values of every other rank's chunk, into that rank's scatter slot [parity][my rank]
    for (uint64_t i0 = tid; i0 < a.n; i0 += U * nthreads) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            v[u] = i < a.n ? a.buf[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            if (i >= a.n) continue;
            const int o = (int)(i / a.chunk);
            if (o != a.rank) fm_store(a.peer[o], sc_par + (size_t)a.rank * a.chunk + (i - (uint64_t)o * a.chunk), v[u], a.tag);
        }
    }
    // ---- owner sum of my chunk in rank order, then all-gather: the sums into every peer's gather slot [parity][my rank]
    const uint64_t lo = (uint64_t)a.rank * a.chunk, hi = lo + a.chunk < a.n ? lo + a.chunk : a.n;
    bool ok = true;
    for (uint64_t i = lo + tid; i < hi && ok; i += nthreads) {
        const uint64_t j = i - lo;
        // the world's partials of element j: all peers' words requested together, again and again until every one carries the tag
        unsigned long long w[D3P_FMESH_MAX_WORLD];
        bool all = false;
        for (uint32_t spins = 0; spins < D3P_FMESH_WAIT_ROUNDS && !all; ++spins) {
#pragma unroll
            for (int r = 0; r < D3P_FMESH_MAX_WORLD; ++r)
                if (r < a.world && r != a.rank)
                    w[r] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(mine) + sc_par + (size_t)r * a.chunk + j, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM);
            all = true;
#pragma unroll
            for (int r = 0; r < D3P_FMESH_MAX_WORLD; ++r)
                if (r < a.world && r != a.rank) all = all && (uint32_t)(w[r] >> 32) == a.tag;
            if (!all && (spins & 255u) == 255u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
        if (!all) { ok = false; break; }
        float s = 0.f;   // in RANK order: r = 0, 1, ... (the own partial in its place)
#pragma unroll
        for (int r = 0; r < D3P_FMESH_MAX_WORLD; ++r)
            if (r < a.world) {
                const float v = r == a.rank ? a.buf[i] : __uint_as_float((uint32_t)w[r]);
                s = r == 0 ? v : s + v;
            }
        a.buf[i] = s;
        for (int p = 0; p < a.world; ++p)
            if (p != a.rank) fm_store(a.peer[p] + a.gather_off, sc_par + (size_t)a.rank * a.chunk + j, s, a.tag);
    }
    // ---- gather: the other chunks' sums from my gather slots, U words requested together
    const unsigned long long* const gin = reinterpret_cast<const unsigned long long*>(mine + a.gather_off) + sc_par;
    for (uint64_t i0 = tid; i0 < a.n && ok; i0 += U * nthreads) {
        size_t word[U];
        bool need[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * nthreads;
            const int o = i < a.n ? (int)(i / a.chunk) : a.rank;
            need[u] = i < a.n && o != a.rank;
            word[u] = need[u] ? (size_t)o * a.chunk + (i - (uint64_t)o * a.chunk) : 0;
        }
        unsigned long long w[U];
        bool all = false;
        for (uint32_t spins = 0; spins < D3P_FMESH_WAIT_ROUNDS && !all; ++spins) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (need[u]) w[u] = __hip_atomic_load(gin + word[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            all = true;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (need[u]) all = all && (uint32_t)(w[u] >> 32) == a.tag;
            if (!all && (spins & 255u) == 255u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
        if (!all) { ok = false; break; }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (need[u]) a.buf[i0 + (uint64_t)u * nthreads] = __uint_as_float((uint32_t)w[u]);
    }
    if (!ok) {   // first code stays: 1 = a word of a peer did not come
        uint32_t expect = 0u;
        (void)__hip_atomic_compare_exchange_strong(a.status, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int fmesh_enqueue_allreduce(hipStream_t s, void* fmesh, float* buf, uint64_t n)
{
    FMesh* x = (FMesh*)fmesh;
    D3P_REQUIRE(x && buf, "d3p_fmesh_allreduce: null pointer");
    D3P_REQUIRE(n == x->n, "d3p_fmesh_allreduce: the mesh was created for another vector length");
    for (int p = 0; p < x->world; ++p) D3P_REQUIRE(x->peer[p], "d3p_fmesh_allreduce: the peers' inboxes are not mapped (d3p_fmesh_connect)");
    FMeshArgs a;
    fmesh_next_args(x, buf, &a);
    hipLaunchKernelGGL(k_fmesh_allreduce, dim3((unsigned)x->wgs), dim3(256), 0, s, a);
    return check_launch("k_fmesh_allreduce");
}

}  // namespace d3p


// ---- the process's hipIpc arena (d3p_ipc_arena.h)
namespace {
struct Arena {
    std::mutex mu;
    bool tried = false;
    char* base = nullptr;
    size_t bytes = 0;
    hipIpcMemHandle_t handle;
    std::vector<std::pair<size_t, size_t>> live;                 // (offset, bytes) of the ranges in use, in address order
    std::map<std::array<uint8_t, 64>, char*> peers;              // arenas of other processes mapped here, by their hipIpc handle
};
Arena& arena() { static Arena a; return a; }

// (under the arena's lock) the allocation + its ONE export; false: no arena (switched off, or the export failed)
bool arena_ready(Arena& A)
{
    if (A.tried) return A.base != nullptr;
    A.tried = true;
    size_t mb = 256;
    if (const char* e = getenv("D3P_IPC_ARENA_MB")) mb = (size_t)strtoull(e, nullptr, 10);
    if (mb == 0) return false;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, mb << 20, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipIpcGetMemHandle(&A.handle, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return false; }
    A.base = (char*)p;
    A.bytes = mb << 20;
    return true;
}
}  // namespace

namespace d3p {

int ipc_range_create(size_t bytes, IpcRange* out, uint8_t handle_out[D3P_IPC_HANDLE_BYTES], const char* who)
{
    bytes = (bytes + 255) & ~(size_t)255;
    memset(handle_out, 0, D3P_IPC_HANDLE_BYTES);
    Arena& A = arena();
    {
        std::lock_guard<std::mutex> lk(A.mu);
        if (arena_ready(A)) {   // first fit between the live ranges
            size_t off = 0;
            size_t at = 0;
            bool found = false;
            for (; at <= A.live.size(); ++at) {
                const size_t end = at < A.live.size() ? A.live[at].first : A.bytes;
                if (end >= off && end - off >= bytes) { found = true; break; }
                if (at < A.live.size()) off = A.live[at].first + A.live[at].second;
            }
            if (found) {
                A.live.insert(A.live.begin() + (long)at, std::make_pair(off, bytes));
                out->ptr = A.base + off;
                out->bytes = bytes;
                out->in_arena = true;
                memcpy(handle_out, &A.handle, sizeof(A.handle));
                const uint64_t o = off, marker = 1;
                memcpy(handle_out + 64, &o, 8);
                memcpy(handle_out + 72, &marker, 8);
            }
        }
    }
    if (out->in_arena) {
        const hipError_t e = hipMemset(out->ptr, 0, bytes);
        if (e != hipSuccess) { ipc_range_destroy(out); return fail(D3P_E_HIP, "%s: hipMemset of the %zu-byte inbox: %s", who, bytes, hipGetErrorString(e)); }
        return D3P_OK;
    }
    // no room in the arena (or no arena): an allocation and an export of its own
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return fail(D3P_E_HIP, "%s: hipExtMallocWithFlags(%zu bytes): %s", who, bytes, hipGetErrorString(e));
    const char* what = "hipMemset";
    e = hipMemset(p, 0, bytes);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) { what = "hipIpcGetMemHandle"; e = hipIpcGetMemHandle(&h, p); }
    if (e != hipSuccess) { (void)hipFree(p); return fail(D3P_E_HIP, "%s: %s of the %zu-byte inbox: %s", who, what, bytes, hipGetErrorString(e)); }
    memcpy(handle_out, &h, sizeof(h));
    out->ptr = (char*)p;
    out->bytes = bytes;
    out->in_arena = false;
    return D3P_OK;
}

void ipc_range_destroy(IpcRange* r)
{
    if (!r->ptr) return;
    if (r->in_arena) {
        Arena& A = arena();
        std::lock_guard<std::mutex> lk(A.mu);
        const size_t off = (size_t)(r->ptr - A.base);
        for (size_t i = 0; i < A.live.size(); ++i)
            if (A.live[i].first == off) { A.live.erase(A.live.begin() + (long)i); break; }
    } else {
        (void)hipFree(r->ptr);
        (void)hipGetLastError();
    }
    r->ptr = nullptr;
}

int ipc_peer_open(const uint8_t handle[D3P_IPC_HANDLE_BYTES], char** ptr_out, bool* opened_out, const char* who, int peer_rank)
{
    uint64_t off = 0, marker = 0;
    memcpy(&off, handle + 64, 8);
    memcpy(&marker, handle + 72, 8);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    if (marker == 1) {   // a range of the peer's arena: the arena is mapped once per process and kept
        Arena& A = arena();
        std::lock_guard<std::mutex> lk(A.mu);
        std::array<uint8_t, 64> key;
        memcpy(key.data(), handle, 64);
        auto it = A.peers.find(key);
        if (it == A.peers.end()) {
            void* q = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return fail(D3P_E_HIP, "%s: hipIpcOpenMemHandle(arena of rank %d): %s", who, peer_rank, hipGetErrorString(e));
            it = A.peers.emplace(key, (char*)q).first;
        }
        *ptr_out = it->second + off;
        *opened_out = false;
        return D3P_OK;
    }
    void* q = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(D3P_E_HIP, "%s: hipIpcOpenMemHandle(rank %d): %s", who, peer_rank, hipGetErrorString(e));
    *ptr_out = (char*)q;
    *opened_out = true;
    return D3P_OK;
}

void ipc_peer_close(char* ptr, bool opened)
{
    if (opened && ptr) { (void)hipIpcCloseMemHandle(ptr); (void)hipGetLastError(); }
}

}  // namespace d3p

using namespace d3p;

extern "C" {

int d3p_fmesh_create(int32_t world, int32_t rank, uint64_t n_floats, void** fmesh_out, uint8_t* handle_out, size_t handle_bytes)
{
    D3P_REQUIRE(fmesh_out && handle_out && handle_bytes >= D3P_IPC_HANDLE_BYTES, "d3p_fmesh_create: null pointer or handle buffer < 80 bytes");
    D3P_REQUIRE(world >= 1 && world <= D3P_FMESH_MAX_WORLD && rank >= 0 && rank < world && n_floats >= 1, "d3p_fmesh_create: bad arguments");
    FMesh* x = new (std::nothrow) FMesh();
    if (!x) return fail(D3P_E_HIP, "d3p_fmesh_create: out of host memory");
    x->world = world;
    x->rank = rank;
    x->n = n_floats;
    x->chunk = (n_floats + (uint64_t)world - 1) / (uint64_t)world;
    x->epoch = 0;
    x->wgs = D3P_FMESH_WGS;
    x->inbox_bytes = 2 * fmesh_region_words(world, x->chunk) * sizeof(unsigned long long) + 64;
    // the inbox: a zeroed range of the process's hipIpc arena (d3p_ipc_arena.h), or an allocation of its own when it does not fit
    IpcRange r;
    if (int rc = ipc_range_create(x->inbox_bytes, &r, handle_out, "d3p_fmesh_create")) { delete x; return rc; }   // (tag 0 is never waited for: epochs count from 1)
    x->inbox = r.ptr;
    x->inbox_bytes = r.bytes;
    x->in_arena = r.in_arena;
    for (int i = 0; i < D3P_FMESH_MAX_WORLD; ++i) { x->peer[i] = nullptr; x->opened[i] = false; }
    x->peer[rank] = x->inbox;
    *fmesh_out = x;
    return D3P_OK;
}

int d3p_fmesh_connect(void* fmesh, const uint8_t* handles, size_t handle_stride)
{
    D3P_REQUIRE(fmesh && handles && handle_stride >= D3P_IPC_HANDLE_BYTES, "d3p_fmesh_connect: bad arguments (handles are 80 bytes)");
    FMesh* x = (FMesh*)fmesh;
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) continue;
        if (int rc = ipc_peer_open(handles + (size_t)p * handle_stride, &x->peer[p], &x->opened[p], "d3p_fmesh_connect", p)) return rc;
    }
    return D3P_OK;
}

int d3p_fmesh_connect_local(void* fmesh, void* const* peers, int32_t world)
{
    D3P_REQUIRE(fmesh && peers, "d3p_fmesh_connect_local: null pointer");
    FMesh* x = (FMesh*)fmesh;
    D3P_REQUIRE(world == x->world, "d3p_fmesh_connect_local: group size differs from the one the mesh was created for");
    for (int p = 0; p < world; ++p) {
        const FMesh* q = (const FMesh*)peers[p];
        D3P_REQUIRE(q && q->rank == p && q->n == x->n && q->world == world, "d3p_fmesh_connect_local: peers must be the group's meshes in rank order");
        x->peer[p] = q->inbox;
    }
    x->wgs = 48;   // the group's ranks share this GPU: see the header
    return D3P_OK;
}

int d3p_fmesh_set_grid(void* fmesh, int32_t workgroups)
{
    D3P_REQUIRE(fmesh && workgroups >= 1 && workgroups <= 1024, "d3p_fmesh_set_grid: 1 <= workgroups <= 1024");
    ((FMesh*)fmesh)->wgs = workgroups;
    return D3P_OK;
}

int d3p_fmesh_allreduce(void* stream, void* fmesh, float* buf_dev, uint64_t n_floats)
{
    return fmesh_enqueue_allreduce((hipStream_t)stream, fmesh, buf_dev, n_floats);
}

int d3p_fmesh_status(void* stream, void* fmesh, int32_t* stopped_out)
{
    D3P_REQUIRE(fmesh && stopped_out, "d3p_fmesh_status: null pointer");
    FMesh* x = (FMesh*)fmesh;
    D3P_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    uint32_t w = 0;
    D3P_HIP_TRY(hipMemcpy(&w, x->inbox + 2 * fmesh_region_words(x->world, x->chunk) * sizeof(unsigned long long), sizeof(w), hipMemcpyDeviceToHost));
    *stopped_out = (int32_t)w;
    return D3P_OK;
}

// Unmap the peers' inboxes (the first half of a teardown: a rank must not FREE its inbox while a peer still has it mapped -- with
// dmabuf IPC the exporter's next allocation + hipIpcGetMemHandle then fails with "invalid argument": every rank disconnects, the
// ranks meet in a barrier, then every rank destroys).  Idempotent; the mesh cannot run a collective afterwards.
int d3p_fmesh_disconnect(void* fmesh)
{
    if (!fmesh) return D3P_OK;
    FMesh* x = (FMesh*)fmesh;
    for (int p = 0; p < x->world; ++p)
        if (p != x->rank && x->peer[p]) {
            ipc_peer_close(x->peer[p], x->opened[p]);   // (a range of a peer's arena stays mapped: nothing to undo)
            x->opened[p] = false;
            x->peer[p] = nullptr;
        }
    return D3P_OK;
}

int d3p_fmesh_destroy(void* fmesh)
{
    if (!fmesh) return D3P_OK;
    FMesh* x = (FMesh*)fmesh;
    for (int p = 0; p < x->world; ++p)
        if (p != x->rank) ipc_peer_close(x->peer[p], x->opened[p]);
    IpcRange r;
    r.ptr = x->inbox; r.bytes = x->inbox_bytes; r.in_arena = x->in_arena;
    ipc_range_destroy(&r);
    delete x;
    return D3P_OK;
}

}  // extern "C"
