"""The slot / flag protocol of csrc/rx_quick_allreduce.hip as a state machine, run under random and adversarial schedules.

The kernel's claim (file header): a slot belongs to a WORKGROUP INDEX, not to a tile -- the workgroup reuses its phase-1 and
phase-2 slots in every peer's region for its next tile WITHOUT any handshake beyond the two flag waits a tile already has,
and flags are "at least" comparisons on a counter that only grows.  Here one workgroup index of W ranks walks T tiles; every
step of a rank (write the W phase-1 sub-slots, raise the W flags, wait, read, write phase 2, raise, wait, read) is atomic,
and a scheduler picks which rank moves.  Checked on every write: the sub-slot's previous payload has been read by its owner;
on every read: the payload is the writer's payload of THIS tile.  Schedules: uniformly random, and "one rank as far ahead as
the protocol lets it" / "one rank starved"."""
import random

import pytest


class Rank:
    def __init__(self, me, world):
        self.me, self.world = me, world
        self.tile = 1            # the colour of the tile in progress (1, 2, ...)
        self.pc = 0              # program counter inside the tile
        # this rank's REGION: slot[phase][src] = (tile, writer) or None, read[phase][src] = last tile read, flag[phase][src]
        self.slot = [[None] * world for _ in range(2)]
        self.read = [[0] * world for _ in range(2)]
        self.flag = [[0] * world for _ in range(2)]
        self.done = False


def runnable(r, ranks):
    if r.done:
        return False
    if r.pc in (2, 6):   # the two waits: "at least" this tile's colour from every source
        ph = 0 if r.pc == 2 else 1
        return all(f >= r.tile for f in r.flag[ph])
    return True


def step(r, ranks, tiles):
    ph = 0 if r.pc < 4 else 1
    if r.pc in (0, 4):      # write my payload of this tile into every rank's sub-slot [me]
        for peer in ranks:
            prev = peer.slot[ph][r.me]
            assert prev is None or peer.read[ph][r.me] >= prev[0], \
                f"rank {r.me} tile {r.tile} phase {ph + 1}: overwrites rank {peer.me}'s unread payload of tile {prev[0]}"
            peer.slot[ph][r.me] = (r.tile, r.me)
    elif r.pc in (1, 5):    # raise my flag in every rank's region (monotonic)
        for peer in ranks:
            assert peer.flag[ph][r.me] < r.tile
            peer.flag[ph][r.me] = r.tile
    elif r.pc in (2, 6):    # the wait (runnable() checked it)
        pass
    else:                   # 3, 7: read my region's W sub-slots
        for src in range(r.world):
            got = r.slot[ph][src]
            assert got == (r.tile, src), f"rank {r.me} tile {r.tile} phase {ph + 1}: read {got} from source {src}"
            r.read[ph][src] = r.tile
    r.pc += 1
    if r.pc == 8:
        r.pc = 0
        r.tile += 1
        if r.tile > tiles:
            r.done = True


def run(world, tiles, pick):
    ranks = [Rank(i, world) for i in range(world)]
    steps = 0
    while not all(r.done for r in ranks):
        ready = [r for r in ranks if runnable(r, ranks)]
        assert ready, "deadlock: nobody can move"
        step(pick(ready, ranks), ranks, tiles)
        steps += 1
    assert steps == world * tiles * 8
    return ranks


@pytest.mark.parametrize("world", [2, 4, 8])
def test_random_schedules_never_overwrite_an_unread_slot(world):
    rng = random.Random(world)
    for trial in range(200 if world < 8 else 60):
        run(world, tiles=6, pick=lambda ready, ranks: rng.choice(ready))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_adversarial_schedules(world):
    # rank 0 runs whenever it can (as far ahead as the flags let it); then the same with rank 0 starved
    run(world, tiles=8, pick=lambda ready, ranks: min(ready, key=lambda r: r.me))
    run(world, tiles=8, pick=lambda ready, ranks: max(ready, key=lambda r: r.me))
    # the most advanced rank always moves first / the least advanced one
    run(world, tiles=8, pick=lambda ready, ranks: max(ready, key=lambda r: (r.tile, r.pc, -r.me)))
    run(world, tiles=8, pick=lambda ready, ranks: min(ready, key=lambda r: (r.tile, r.pc, r.me)))


def test_the_check_catches_a_protocol_without_the_second_wait():
    """Sanity of the model: drop the phase-2 wait + read (a rank goes on to its next tile right after raising the phase-2
    flag) and a fast rank overwrites a slow peer's unread phase-1 payload."""
    world, tiles = 2, 3
    ranks = [Rank(i, world) for i in range(world)]

    def broken_step(r):
        step(r, ranks, tiles)
        if r.pc == 6:          # skip the wait and the read of phase 2
            r.pc = 0
            r.tile += 1
            r.done = r.tile > tiles

    with pytest.raises(AssertionError, match="overwrites|read"):
        for _ in range(200):
            ready = [r for r in ranks if runnable(r, ranks)]
            if not ready:
                break
            broken_step(min(ready, key=lambda r: r.me))


# ---- the decode-sized kernels of csrc/rx_allreduce.hip: staging / result buffers alternate by the call's parity ----------------
class TwoShotRank:
    """One block index of one rank.  Region: stage[2], result[2] (payload = the call number that wrote it), flags ready / done
    per source; reads are recorded per reader so that a writer can check that everybody is done with the old payload."""

    def __init__(self, me, world):
        self.me, self.world = me, world
        self.call, self.pc, self.finished = 1, 0, False
        self.stage, self.result = [0, 0], [0, 0]
        self.ready, self.done = [0] * world, [0] * world
        self.read_stage = [[0, 0] for _ in range(world)]    # [source][buffer] -> last call whose payload I read
        self.read_result = [[0, 0] for _ in range(world)]


def ts_runnable(r):
    if r.finished:
        return False
    if r.pc == 2:
        return all(r.ready[p] >= r.call for p in range(r.world) if p != r.me)
    if r.pc == 5:
        return all(r.done[p] >= r.call for p in range(r.world) if p != r.me)
    return True


def ts_step(r, ranks, calls, one_shot, buffers=2):
    buf = r.call % buffers
    others = [p for p in ranks if p.me != r.me]
    if r.pc == 0:      # stage my input
        old = r.stage[buf]
        for p in others:
            assert old == 0 or p.read_stage[r.me][buf] >= old, f"rank {r.me} call {r.call}: restages over call {old}, unread by rank {p.me}"
        r.stage[buf] = r.call
    elif r.pc == 1:
        for p in others:
            p.ready[r.me] = r.call
    elif r.pc == 3:    # reduce: read every staging buffer, write my result
        for p in others:
            assert p.stage[buf] == r.call, f"rank {r.me} call {r.call}: rank {p.me}'s staging buffer holds call {p.stage[buf]}"
            r.read_stage[p.me][buf] = r.call
        if not one_shot:
            old = r.result[buf]
            for p in others:
                assert old == 0 or p.read_result[r.me][buf] >= old, f"rank {r.me} call {r.call}: result of call {old} unread by rank {p.me}"
            r.result[buf] = r.call
    elif r.pc == 4:
        for p in others:
            p.done[r.me] = r.call
    elif r.pc == 6:    # gather
        for p in others:
            assert p.result[buf] == r.call
            r.read_result[p.me][buf] = r.call
    r.pc += 1
    if r.pc == (4 if one_shot else 7):
        r.pc = 0
        r.call += 1
        r.finished = r.call > calls


@pytest.mark.parametrize("one_shot", [False, True], ids=["two_shot", "one_shot"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_two_shot_buffers_alternate_safely(world, one_shot):
    """rx_allreduce.hip header: 'a buffer is reused at g + 2, after the peer's ready of g + 1 proved it finished g' -- for the
    two-shot form and for the one-shot (deterministic) form, which has no second exchange at all."""
    rng = random.Random(10 * world + one_shot)
    picks = [lambda ready: rng.choice(ready)] * 40 + [lambda ready: min(ready, key=lambda r: r.me), lambda ready: max(ready, key=lambda r: r.me),
                                                      lambda ready: max(ready, key=lambda r: (r.call, r.pc)), lambda ready: min(ready, key=lambda r: (r.call, r.pc))]
    for pick in picks:
        ranks = [TwoShotRank(i, world) for i in range(world)]
        while not all(r.finished for r in ranks):
            ready = [r for r in ranks if ts_runnable(r)]
            assert ready, "deadlock"
            ts_step(pick(ready), ranks, calls=7, one_shot=one_shot)


def test_one_buffer_would_not_be_enough_for_the_one_shot_form():
    """Sanity of the model: with ONE staging buffer (no parity) a fast rank restages while a slow peer still has to read."""
    ranks = [TwoShotRank(i, 2) for i in range(2)]
    with pytest.raises(AssertionError, match="restages|holds"):
        while not all(r.finished for r in ranks):
            ready = [r for r in ranks if ts_runnable(r)]
            assert ready, "deadlock"
            ts_step(min(ready, key=lambda r: r.me), ranks, calls=4, one_shot=True, buffers=1)   # (rank 0 runs whenever it can)
