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
