// celerite_amd/csrc/clr_group_hooks.h -- what csrc/sharded.cpp needs from a single-device plan beyond the C ABI
// (include/celerite_hip.h), so that a batch gives the SAME BITS under any sharding (all state of
// the reference solver is per problem, cholesky.h:703-706).  Internal to libcelerite_hip.so: not exported through the
// public header, no stability promise.
//
// A plan takes a handful of decisions from counts over "its" problems: the prefix plan's time model and the one-launch
// path look at the batch size, the warm-started recurrence is switched on when half of the problems are eligible and
// lengthens its warm-ups when many of them fail their boundary check, level-1 problems are re-planned as a side plan
// whose chunk count follows their number -- or replayed inline when there are too many.  For a plan that is one slice
// of a larger batch these counts are taken over the WHOLE batch: the sharded layer adds them up between the two
// halves of each decision and hands the totals to every shard.
#pragma once

struct clr_batch;

namespace clr_group {

// this plan holds a contiguous slice of a batch of `B_total` problems (<= the plan's own size: it is the whole batch)
void set_batch_context(clr_batch* h, int B_total);

// warm-started recurrence: problems of this plan that could start warm at the series / coefficients in force, and the
// count over the whole batch (activation: at least half of the batch)
long warm_eligible(const clr_batch* h);
void set_warm_eligible_total(clr_batch* h, long eligible_total);

// an evaluation in flight with pending problems (warm path, one-launch path, deferred level-1 problems):
// resolve_begin waits for it and reports the plan's counts, resolve_finish settles the pending problems using the
// counts of the whole batch.  Both are no-ops when nothing is in flight.
bool in_flight(const clr_batch* h);  // (host state only)
int resolve_begin(clr_batch* h, long* pending, long* eligible);
int resolve_finish(clr_batch* h, long pending_total, long eligible_total);

}  // namespace clr_group
