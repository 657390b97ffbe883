// notify.hpp -- completion notice of a ONE-WORKGROUP launch through a word in host-visible memory.
//
// The reference-shaped single-frame entries (capi.hip: decode_one -> host_pipeline's direct small-call path) are one launch and one
// wait.  hipStreamSynchronize after a one-wave kernel costs 12.5 us on this platform; the same kernel storing a ticket into pinned
// mapped memory when it is done, with the host spinning on that word, 8.2 us (tools/ubench/launch_floor.hip,
// profiles/r06_kbench/launch_floor.txt).  So: the caller posts a request (where, which ticket) in a thread-local slot before it calls
// the launcher; a launcher whose launch is the call's ONLY kernel and has ONE workgroup takes it and launches the kernel's notifying
// twin with (flag, ticket); that kernel ends with notify_done(); the caller sees `taken` and spins instead of synchronising.  Launches that do not
// qualify leave the request alone and the caller synchronises as before.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace ldpc {

struct NotifyRequest { uint32_t *flag = nullptr; uint32_t ticket = 0; bool taken = false; };
inline thread_local NotifyRequest g_notify;

// launcher side: true with the (flag, ticket) for a kernel that is the call's only launch and has one workgroup
inline bool take_notify(size_t grid, uint32_t *&flag, uint32_t &ticket)
{
    NotifyRequest &r = g_notify;
    if (grid != 1 || r.flag == nullptr || r.taken) return false;
    flag = r.flag; ticket = r.ticket; r.taken = true;
    return true;
}

// kernel side, as the kernel's last statement, reached by every thread of the (one) workgroup: every thread's results are on their
// way to system scope before the barrier, thread 0's release store of the ticket follows them
__device__ __forceinline__ void notify_done(uint32_t *flag, uint32_t ticket)
{
    if (flag != nullptr) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace ldpc
