// kernels_spec.h — the marginalization run AHEAD of the loop's end (round 6).
//
// The reference marginalizes the state the solve has left (estimator.cpp:825-833: ceres::Solve, double2vector, then
// MarginalizationInfo of the solved state, :833-939).  That state is known long before the trust-region loop knows it is done: a
// rejected step leaves x untouched (Ceres' HandleUnsuccessfulStep), the tolerance tests end the loop on the last ACCEPTED point, and
// the passes behind the last acceptance only confirm it.  A single small window leaves 250 of the chip's 256 CUs idle, so the gauge
// fix + frame-0 sweep + k_marg_solve (~0.19 ms, a third of the call) of every newly accepted state are started beside the loop:
//
//   stream 0 (the loop, one graph as before)      k_lin publishes "state a is complete in x[cur]" (SpecCtl::word) in every pass;
//                                                 the gated gauge fix at the loop's end settles who owns the prior of the final state
//   stream 1 (workers: R rounds of four launches) k_spec_begin waits for a state newer than the last one it worked on, copies it
//                                                 into the SHADOW slot (a second Slot behind the context's last one whose input
//                                                 pointers lead back into slot 0 — the marginalization's kernels run on it unchanged),
//                                                 re-anchors it like double2vector; k_lin / k_sum (MODE_MARG, gated) and k_marg_solve
//                                                 follow.  k_marg_solve gives up between its phases when a newer state is out,
//                                                 and at its end waits for the loop to close: if the state it worked on is the
//                                                 final one, the prior goes into slot 0 (and its mailbox) — the same bits the
//                                                 serial tail would have produced, from the same kernels on the same numbers.
//
// Ownership of the prior of "a accepted steps" is one word, SpecCtl::own[a]:
//   worker:  copy the state, see `word` unchanged and `fin` still open, THEN  CAS(FREE -> SIDE)   (a state whose copy may be
//            torn by the in-place gauge fix is never claimed: the loop raises FIN_CLOSING before it touches x[cur])
//   loop:    at its end  CAS(FREE -> MAIN): it runs the serial tail itself; finds SIDE: CAS(SIDE -> COMMIT) and skips its tail —
//            the worker must deliver now; finds ABANDON: MAIN.
//   worker:  waited too long for the loop to close: CAS(SIDE -> ABANDON); if that fails the loop has committed it: deliver.
// No party waits for the other without a bound except the host for a COMMITted worker, which holds a finished prior.
// Every wait of a worker is bounded by a wall-clock limit (a stream that shares its hardware queue with the loop's would
// otherwise never see the word it waits for).
#pragma once
#include "dev_math.h"
#include "dev_types.h"

constexpr long long SPEC_WAIT_TICKS = 300000;  // 3 ms of the 100 MHz wall clock: the longest a worker waits for the loop

DEV int spec_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
// for the waiting loops: a device-scope load that invalidates nothing (an acquire per turn of a loop empties the non-local lines of the
// L2 the spinning wave shares with an eighth of the chip, turn after turn); spec_acquire() once when the wait is over
DEV int spec_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void spec_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
DEV void spec_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
DEV int spec_cas(int *p, int expect, int v) {
  __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  return expect;  // what was there
}
DEV int spec_ep(int w) { return (int)((unsigned)w >> 16); }
DEV int spec_acc(int w) { return (w & 0xffff) >> 1; }
DEV int spec_fin_state(int f) { return f & 3; }
DEV int spec_fin_acc(int f) { return (f & 0xffff) >> 2; }

// ---- the loop's side (stream 0)
// k_setup, one thread: a new call of this slot.  `word` is withdrawn BEFORE `fin` re-opens (a worker reads fin, then word).
DEV void spec_arm(Slot *S) {
  const int ep = (S->spec.ep % 0x7fff) + 1;
  spec_st(&S->spec.word, 0);
  for (int k = 0; k < SPEC_OWN; k++) S->spec.own[k] = SPEC_FREE;
  S->spec.ep = ep;
  S->spec.ticket = S->mail ? __hip_atomic_load((const int *)S->mail + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
  spec_st(&S->spec.fin, 0);
}
// k_lin, one thread of every pass, as soon as the accepted state the pass linearizes at is complete in x[cur] / lam[cur]: at once
// when it is the header's or candidate 0 (written where it lies by the pass before), behind the copy when it was a speculative
// candidate (copy_accepted, the last workgroup).  The pass's own k_step — the first kernel to overwrite x[cur ^ 1] — comes later.
DEV void spec_publish(Slot *S, int num_succ, int cur) {
  // (ticket 0: the host started no workers for this call; whatever rounds are still around from the call before must find nothing)
  if (num_succ < SPEC_OWN && S->spec.ticket != 0) spec_st(&S->spec.word, S->spec.ep << 16 | num_succ << 1 | cur);
}
// the gated gauge fix, one thread, BEFORE the first in-place store (the caller puts a workgroup barrier behind it)
DEV void spec_closing(Slot *S) {
  __hip_atomic_store(&S->spec.fin, S->spec.ep << 16 | FIN_CLOSING, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence();
}
// ... and behind it (the state is out): who forms the prior.  Returns true when a worker owns it — tail_state 3, the gated
// kernels that follow return.
DEV bool spec_settle(Slot *S) {
  const int a = S->tr.num_succ;
  bool side = false;
  if (a < SPEC_OWN) {
    const int was = spec_cas(&S->spec.own[a], SPEC_FREE, SPEC_MAIN);
    if (was == SPEC_SIDE) side = spec_cas(&S->spec.own[a], SPEC_SIDE, SPEC_COMMIT) == SPEC_SIDE;
    if (was != SPEC_FREE && !side) spec_st(&S->spec.own[a], SPEC_MAIN);
  }
  if (side) S->tail_state = 3, S->iters_done = S->tr.iteration;  // (what k_marg_solve leaves for the host's copy of {tail_state .. chain_err})
  spec_st(&S->spec.fin, S->spec.ep << 16 | (a < SPEC_OWN ? a : 0) << 2 | (side ? FIN_SIDE : FIN_MAIN));
  return side;
}

// ---- a worker's side (stream 1).  S: the shadow slot, S0: the slot being solved.
// Has the state this round works on been overtaken?  One thread.
DEV bool spec_stale(const Slot *S0, int my_word) {
  const int f = spec_ld(&S0->spec.fin), w = spec_ld(&S0->spec.word);
  if (spec_ep(w) != spec_ep(my_word)) return true;  // the slot has gone on to another call
  if (w != my_word) return true;                    // a newer accepted state is out
  const int st = spec_ep(f) == spec_ep(my_word) ? spec_fin_state(f) : FIN_OPEN;
  return st >= FIN_MAIN && spec_fin_acc(f) != spec_acc(my_word);  // the loop closed on another state (its last pass accepted a step)
}

// Polling without a wait: a thread that is idle in the phase issues the two loads at one point of the kernel (relaxed, device
// scope: nothing is read through them) and looks at the values a phase later, in front of a barrier the kernel has anyway.
struct SpecPoll {
  const Slot *S0;
  int my_word;   // 0: not a worker's launch, nothing is polled
  int *flag;     // LDS: set once the state has been overtaken; every thread reads it behind a barrier
  int f, w;      // the poller's loads in flight
};
DEV void spec_poll_issue(SpecPoll &p) {
  p.f = __hip_atomic_load(&p.S0->spec.fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  p.w = __hip_atomic_load(&p.S0->spec.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEV void spec_poll_take(SpecPoll &p) {
  const int st = spec_ep(p.f) == spec_ep(p.my_word) ? spec_fin_state(p.f) : FIN_OPEN;
  if (p.w != p.my_word || (st >= FIN_MAIN && spec_fin_acc(p.f) != spec_acc(p.my_word))) *p.flag = 1;
}
