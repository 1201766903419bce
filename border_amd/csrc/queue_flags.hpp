// Ordering two HIP streams of ONE agent by device flags instead of events (the SAC step's side queue, sac.hip).
//
// hipEventRecord / hipStreamWaitEvent put a barrier packet into both queues; on this platform each costs its queue a 5-7 us
// bubble (LAB.md section 5), which is a whole stage of a launch-bound step.  Here the producer queue runs a one-wave kernel
// that stores the step's epoch to a flag word, and the consumer queue runs a one-wave kernel that returns once the flag has
// reached the epoch:
//   k_flag_set   a normal dispatch (barrier bit set): it starts after everything queued before it on its stream has completed
//                and released its writes, so the store IS "my predecessors are complete";
//   k_flag_wait  spins (s_sleep) on an agent-scope acquire load; the kernel queued behind it starts after it has returned and
//                acquires at its own start.
// Every wait of the SAC step is for work submitted EARLIER than the waiting kernel (the previous update's actor step, this
// update's prologue), so streams that alias one hardware queue (GPU_MAX_HW_QUEUES) run the same packets in submission order
// and no wait can be for something queued behind it: slower, never a deadlock.  A producer that never arrives trips the time
// limit instead of hanging the queue; the agent's error word reports it at the next synchronisation (and see k_flag_wait).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace bdr {

static __global__ __launch_bounds__(64) void k_flag_set(unsigned* flag, unsigned epoch)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// limit in ticks of the 100 MHz wall clock; on a timeout *err_word = err_code (when given) and the kernel returns.  The error word is
// also the agent's POISON: while it is up every later wait returns at once (no second time limit), the kernels behind a failed wait
// run unordered, and the kernels that write parameters, moments and targets skip their update (agent_base.hpp, dense.hpp) - the state
// stays that of the last good update until the host has seen the error and cleared the word.
static __global__ __launch_bounds__(64) void k_flag_wait(const unsigned* flag, unsigned epoch, unsigned long long limit, unsigned* err_word, unsigned err_code)
{
    if (threadIdx.x != 0) return;
    if (err_word && __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // poisoned
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {   // wrap-safe
        __builtin_amdgcn_s_sleep(4);
        if (err_word && __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // another wait failed first
        if (wall_clock64() - t0 > limit) {
            if (err_word) __hip_atomic_store(err_word, err_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

constexpr unsigned long long FLAG_WAIT_LIMIT = 1000000000ull;   // 10 s

inline int32_t flag_set(hipStream_t st, unsigned* flag, unsigned epoch)
{
    hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, st, flag, epoch);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}
inline int32_t flag_wait(hipStream_t st, const unsigned* flag, unsigned epoch, unsigned* err_word, unsigned err_code, unsigned long long limit = FLAG_WAIT_LIMIT)
{
    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, st, flag, epoch, limit, err_word, err_code);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}

// Do `waiter` and `producer` sit on different hardware queues?  A wait with a 100 ms limit on `waiter`, THEN the flag store on
// `producer`: on one in-order queue the store is behind the wait and the wait times out.  scratch: two zeroed device words.
// Only a performance question for the SAC step (see above) - with aliased queues the side queue buys nothing, and the agent
// keeps the one-queue sequence.
inline int32_t flag_queues_independent(hipStream_t waiter, hipStream_t producer, unsigned* scratch, bool* ok)
{
    BDR_HIP(hipMemsetAsync(scratch, 0, 2 * sizeof(unsigned), waiter));
    BDR_HIP(hipStreamSynchronize(waiter));
    BDR_TRY(flag_wait(waiter, scratch, 1u, scratch + 1, 1u, 10000000ull));
    BDR_TRY(flag_set(producer, scratch, 1u));
    BDR_HIP(hipStreamSynchronize(waiter));
    BDR_HIP(hipStreamSynchronize(producer));
    unsigned err = 0;
    BDR_HIP(hipMemcpy(&err, scratch + 1, sizeof err, hipMemcpyDeviceToHost));
    *ok = err == 0;
    return BDR_OK;
}

}  // namespace bdr
