// cm_peer.hip -- one-shot peer all-reduce of the [gradient | statistics] buffer over hipIpc-mapped mailboxes.
//
// The exchange step of an env-sharded run is ONE latency-bound message per network and optimiser step (33 KB actor, 116 KB critic at
// config 3; it stands where cleanmarl/mappo_multienvs.py:578-594 steps the optimisers on the full batch).  Through RCCL such a message
// costs >= 27 us on the stream (measured floor of a one-rank all-reduce, tools/probes/rccl_one_rank_latency.py) and the three actor
// messages of an iteration are exposed: the next actor pass needs the step.  Here every rank owns a MAILBOX in fine-grained device
// memory, mapped into its peers with hipIpc: slots [2][world][n] + one tag word per slot.
//   push     one launch: the rank copies its reduced buffer into slot [seq & 1][rank] of EVERY mailbox (its own included; peer writes
//            travel over xGMI), makes them visible system-wide and then publishes the tag word {seq} of that slot in every mailbox;
//   step     the optimiser-step launch of cm_optim.hip with the mailbox's `world` slots as its partial rows: it first waits for the
//            `world` tag words of its own mailbox (bounded by WALL time: a wait that runs out skips the step and raises a host-visible
//            status word -- see cm_optim.hip), then folds the slots IN RANK ORDER (every rank the same order: bit-identical
//            parameters on all ranks), scales by grad_scale / N, takes the norm and applies the update.
// Two slot sets alternate by seq parity: a peer can be at most one step ahead (its push of step s + 2 needs this rank's push of step
// s + 1, issued after this rank's step s has read its slots).  No collective library call, no host involvement on the data path.
#include "cm_common.h"

namespace {

constexpr size_t PEER_HDR = 4096;  // tag words u64 [2][world] at 0, the pushing rank's block counters u32 [world] at 2048

__host__ __device__ inline size_t peer_npad(int64_t n) { return (size_t)((n + 63) / 64 * 64); }

struct PushArgs {
    const float* buf; long n; size_t npad; int rank, world; unsigned seq;
    char* mbox[16];   // every rank's mailbox as mapped in THIS process (mbox[rank] = own)
};

__global__ __launch_bounds__(256) void k_peer_push(const PushArgs a) {
    const int p = blockIdx.y;                       // destination rank
    float* slot = reinterpret_cast<float*>(a.mbox[p] + PEER_HDR) + ((size_t)(a.seq & 1) * a.world + a.rank) * a.npad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) slot[i] = a.buf[i];
    __threadfence_system();                         // this block's part of the slot is visible to every agent ...
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* cnt = reinterpret_cast<unsigned*>(a.mbox[a.rank] + 2048) + p;
        if (atomicAdd(cnt, 1u) == gridDim.x - 1) {  // ... and the last block of this destination publishes the slot
            *cnt = 0u;
            unsigned long long* tag = reinterpret_cast<unsigned long long*>(a.mbox[p]) + (size_t)(a.seq & 1) * a.world + a.rank;
            __hip_atomic_store(tag, (unsigned long long)a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

extern "C" size_t cm_peer_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }
extern "C" size_t cm_peer_mailbox_bytes(int world, int64_t n_floats) {
    return PEER_HDR + 2 * (size_t)world * peer_npad(n_floats) * sizeof(float);
}

extern "C" int cm_peer_mailbox_alloc(size_t bytes, void** mailbox, void* handle_out) {
    CM_REQUIRE(mailbox && handle_out && bytes >= PEER_HDR, "cm_peer_mailbox_alloc: bad arguments");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);  // peer writes must be visible to a RUNNING kernel of the owner
    if (e != hipSuccess) CM_FAIL(-2, "cm_peer_mailbox_alloc: hipExtMallocWithFlags(%zu, fine-grained): %s", bytes, hipGetErrorString(e));
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), p);
    if (e != hipSuccess) { (void)hipFree(p); CM_FAIL(-2, "cm_peer_mailbox_alloc: %s", hipGetErrorString(e)); }
    *mailbox = p;
    return 0;
}
extern "C" int cm_peer_mailbox_open(const void* handle, void** mailbox) {
    CM_REQUIRE(handle && mailbox, "cm_peer_mailbox_open: NULL argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(mailbox, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) CM_FAIL(-2, "cm_peer_mailbox_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int cm_peer_mailbox_close(void* mailbox) {
    const hipError_t e = hipIpcCloseMemHandle(mailbox);
    if (e != hipSuccess) CM_FAIL(-2, "cm_peer_mailbox_close: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int cm_peer_mailbox_free(void* mailbox) {
    const hipError_t e = hipFree(mailbox);
    if (e != hipSuccess) CM_FAIL(-2, "cm_peer_mailbox_free: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int cm_peer_push(const float* buf, int64_t n_floats, int rank, int world, void* const* mailboxes, uint32_t seq, cm_stream_t stream) {
    CM_REQUIRE(buf && mailboxes && n_floats > 0 && world >= 1 && world <= 16 && rank >= 0 && rank < world && seq != 0,
               "cm_peer_push: bad arguments (n=%ld rank=%d world=%d seq=%u; at most 16 ranks, seq starts at 1)", (long)n_floats, rank, world, seq);
    PushArgs a = {};
    a.buf = buf; a.n = n_floats; a.npad = peer_npad(n_floats); a.rank = rank; a.world = world; a.seq = seq;
    for (int p = 0; p < world; ++p) {
        CM_REQUIRE(mailboxes[p], "cm_peer_push: mailbox of rank %d is NULL", p);
        a.mbox[p] = (char*)mailboxes[p];
    }
    const int nb = (int)((n_floats + 4095) / 4096);  // 16 floats per thread
    hipLaunchKernelGGL(k_peer_push, dim3(nb < 64 ? nb : 64, world), dim3(256), 0, (hipStream_t)stream, a);
    CM_CHECK_LAUNCH("cm_peer_push");
    return 0;
}

// the step launch: partial rows = the slots of this rank's own mailbox (cm_optim.hip waits for their tag words first)
// timeout_s: wall-time bound of that wait (<= 0: 30 s).  status: optional word in host-visible (page-locked) or device memory; a launch
// whose wait ran out SKIPS the step (parameters / moments / gradient buffer untouched, logged norm NaN) and writes seq there.
extern "C" int cm_optimizer_step_peer(float* grad_and_stats, int64_t n_params, void* own_mailbox, int world, uint32_t seq,
                                      const cm_opt_step_t* opt, double timeout_s, uint32_t* status, cm_stream_t stream) {
    CM_REQUIRE(grad_and_stats && own_mailbox && world >= 1 && world <= 16 && seq != 0, "cm_optimizer_step_peer: bad arguments");
    const size_t npad = peer_npad(n_params + CM_NUM_STATS);
    const float* slots = reinterpret_cast<const float*>((char*)own_mailbox + PEER_HDR) + (size_t)(seq & 1) * world * npad;
    const unsigned long long* tags = reinterpret_cast<const unsigned long long*>(own_mailbox) + (size_t)(seq & 1) * world;
    return cm_launch_reduce_step(slots, world, (int)npad, nullptr, 0, 0, 0, n_params, grad_and_stats, opt, (hipStream_t)stream,
                                 "cm_optimizer_step_peer", tags, seq, timeout_s, status);
}
