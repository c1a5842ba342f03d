"""Exhaustive model of the barrier that the four workgroups of a row use inside estts_decoder_block_row
(csrc/decoder_row.hip: row_barrier_arrive / row_barrier_wait).  Two counters per row, X for the barrier at hand and O for the
row's other barrier: a workgroup adds 1 to X; the one that draws ticket 3 stores 0 to O and then 5 to X; the others poll X
until it reads 5.  A launch passes barrier 1 (X = A, O = B) and then barrier 2 (X = B, O = A); the next launch starts from
whatever the last one left.  Every interleaving of the workgroups' atomic steps is explored and checked for: no workgroup
passes a barrier before all four have arrived at it, no state in which nobody can move, and the same counter state at the end
of every launch, from the all-zero state of a fresh allocation and from the state a launch leaves behind."""
from collections import deque

N = 4                      # workgroups of a row
RELEASED = N + 1

# program of one workgroup: (barrier index) x (step within the barrier)
#   step 0: ticket = fetch_add(X, 1)
#   step 1: ticket == N - 1 ? store(O, 0) : poll X   (a poll that does not read RELEASED stays at step 1)
#   step 2: (last arriver only) store(X, RELEASED)
#   step 3: passed


def explore(start_counters):
    """All reachable states of one launch; returns the set of final counter states."""
    init = (start_counters, tuple((0, 0, -1) for _ in range(N)))       # per workgroup: (barrier, step, ticket)
    seen, queue, finals = {init}, deque([init]), set()
    while queue:
        counters, wgs = queue.popleft()
        if all(b == 2 for b, _, _ in wgs):
            finals.add(counters)
            continue
        moved = False
        for w, (b, step, ticket) in enumerate(wgs):
            if b == 2:
                continue
            x, o = b, 1 - b                                             # barrier 1 uses counter 0, barrier 2 counter 1
            c = list(counters)
            if step == 0:
                ticket = c[x]
                # a ticket is a count of arrivals: it can never be read from a released or stale counter
                assert 0 <= ticket < N, ('arrival at a counter that is not counting', counters, wgs)
                c[x] += 1
                nxt = (b, 1, ticket)
            elif step == 1:
                if ticket == N - 1:
                    c[o] = 0
                    nxt = (b, 2, ticket)
                elif c[x] == RELEASED:
                    # safety: released only after all N have arrived -- every workgroup is at this barrier or past it
                    assert all(bb > b or (bb == b and ss >= 1) for bb, ss, _ in wgs), ('passed early', counters, wgs)
                    nxt = (b + 1, 0, -1)
                else:
                    continue                                            # a poll that reads something else: no change of state
            else:                                                       # step 2, last arriver
                c[x] = RELEASED
                nxt = (b + 1, 0, -1)
            moved = True
            state = (tuple(c), wgs[:w] + (nxt,) + wgs[w + 1:])
            if state not in seen:
                seen.add(state)
                queue.append(state)
        assert moved, ('nobody can move', counters, wgs)
    return finals, len(seen)


def test_row_barrier_is_safe_live_and_leaves_the_counters_ready():
    finals, n_states = explore((0, 0))                                  # counters as allocated
    assert finals == {(0, RELEASED)} and n_states > 500          # (671 interleaving states)
    again, _ = explore((0, RELEASED))                                   # ... and as a launch leaves them
    assert again == {(0, RELEASED)}


def test_row_barrier_model_catches_a_reset_by_the_wrong_party():
    # the obvious simpler protocol -- the last arriver resets its OWN counter to 0 and the others wait for the count to reach N --
    # loses a waiter that polls after the reset; the model must see that (a check of the checker)
    init = ((0,), tuple((0, -1) for _ in range(N)))
    seen, queue, stuck = {init}, deque([init]), False
    while queue:
        (cnt,), wgs = queue.popleft()
        if all(s == 2 for s, _ in wgs):
            continue
        moved = False
        for w, (step, ticket) in enumerate(wgs):
            c = cnt
            if step == 0:
                nxt, c = (1, cnt), cnt + 1
            elif step == 1 and ticket == N - 1:
                nxt, c = (2, ticket), 0                                  # reset at once
            elif step == 1 and cnt == N:
                nxt = (2, ticket)
            else:
                continue
            moved = True
            state = ((c,), wgs[:w] + (nxt,) + wgs[w + 1:])
            if state not in seen:
                seen.add(state)
                queue.append(state)
        stuck = stuck or not moved
    assert stuck
