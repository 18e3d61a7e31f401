"""Would one closed-form lead PER SERIES (instead of the call's shortest lead) pay on BASELINE config 5?
("estimate first".)  CPU only.

The LEAD kernels sweep the tail [T - tail, T) of every series of a call with ONE kernel instantiation,
i.e. one chunk length L = ceil(tail / lanes per cell); the call's tail is max(T - shortest lead, 80)
rounded up to a multiple of 16 (ldsr_api.hip lead_tail).  A lane's sweep costs L steps whatever the
number of active lanes, so a per-series lead inside one launch moves steps from the sweeps to the lead
of THAT series but does not shorten any lane's chunk: it saves nothing.  What can save is grouping the
series by tail class into launches of their own chunk length.  This script counts both."""
import numpy as np

T, S, R = 813, 48, 512
n_tail = np.array([30 + (s * 60) // (S - 1) for s in range(S)])          # bench.py build_problem("cfg5")
call_tail = int(-(-max(n_tail.max(), 80) // 16) * 16)
own_tail = np.array([int(-(-max(t, 80) // 16) * 16) for t in n_tail])
LPC = 16                                                                  # four cells per wave (tails <= 256)
L_call = -(-call_tail // LPC)
L_own = -(-own_tail // LPC)
# fp64 operations per cell and EM iteration: masked generic sweeps ~85 per step of the
# tail as seen by a lane (L steps on every lane), lead passes ~16 per step of the lead
ops = lambda L, lead: 85.0 * L * LPC + 16.0 * lead
now = ops(L_call, T - call_tail) * S
grouped = sum(ops(L_own[s], T - own_tail[s]) for s in range(S))
print("config 5: tails %d..%d observed steps; the call sweeps %d steps (L = %d) on every series" % (n_tail.min(), n_tail.max(), call_tail, L_call))
print("per-series tails: %s" % dict(zip(*np.unique(own_tail, return_counts=True))))
print("steps swept: call-minimum lead %d, per-series leads %d (%.1f %% fewer)" % (call_tail * S, own_tail.sum(), 100 * (1 - own_tail.sum() / (call_tail * S))))
print("one launch, per-series lead, common L = %d: the sweeps cost L steps per lane either way -> 0 %% saved" % L_call)
print("series grouped by tail class into launches of their own L: modelled fp64 work %.3g -> %.3g per iteration (%.1f %% less),"
      % (now, grouped, 100 * (1 - grouped / now)))
print("  against one more launch (~10 us of series_prep + dispatch on a 2.7 ms step: +0.4 %) and two partly filled rounds instead of three full ones")
