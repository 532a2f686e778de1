"""rsm_gather_clouds' protocol over a MOCK transport (CPU, host memory): world 2 / 3 / 8 ranks as threads of one process,
every rank driving the real C-ABI entry point (`rsm_comm_create_transport` + `rsm_gather_clouds`).  The mock keeps
RCCL's matching rule -- point-to-point operations between two ranks pair up IN POSTING ORDER, a size mismatch is an
error -- so a sender that walks its pairs in another order than the root's receives is caught here, without a
multi-GPU node.  Replaces the global accumulation of CCloudOptimization.cpp:61,123 fed by the pair loop
CStereoMatching.cpp:17-33."""
import collections
import ctypes as C
import threading

import numpy as np
import pytest

from reconstruction_amd import _lib

REC = 16


class GatherOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("peer", C.c_int), ("pair", C.c_int), ("local_index", C.c_int),
                ("offset", C.c_int64), ("count", C.c_int64)]


ALLRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int)
VOIDFN = C.CFUNCTYPE(C.c_int, C.c_void_p)
SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int)
RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int)
COPY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)


class Transport(C.Structure):
    _fields_ = [("self", C.c_void_p), ("allreduce_sum_i64", ALLRED), ("group_begin", VOIDFN), ("send", SEND),
                ("recv", RECV), ("copy", COPY), ("group_end", VOIDFN)]


class Fabric:
    """Shared state of the in-process ranks."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.lock = threading.Condition()
        self.queues = collections.defaultdict(collections.deque)   # (src, dst) -> messages in posting order
        self.mismatch = []


class MockRank:
    def __init__(self, fabric, rank):
        self.f, self.rank, self.pending = fabric, rank, []
        self.tp = Transport(None, ALLRED(self.allreduce), VOIDFN(self.group_begin), SEND(self.send), RECV(self.recv),
                            COPY(self.copy), VOIDFN(self.group_end))

    def allreduce(self, _self, buf, n):
        a = np.ctypeslib.as_array(buf, shape=(n,))
        self.f.slots[self.rank] = a.copy()
        self.f.barrier.wait(timeout=30)
        total = sum(self.f.slots)
        self.f.barrier.wait(timeout=30)
        a[:] = total
        return 0

    def group_begin(self, _self):
        self.pending = []
        return 0

    def send(self, _self, buf, nbytes, peer):
        with self.f.lock:
            self.f.queues[(self.rank, peer)].append(C.string_at(buf, nbytes))
            self.f.lock.notify_all()
        return 0

    def recv(self, _self, buf, nbytes, peer):
        self.pending.append((buf, nbytes, peer))
        return 0

    def copy(self, _self, dst, src, nbytes):
        C.memmove(dst, src, nbytes)
        return 0

    def group_end(self, _self):
        for buf, nbytes, peer in self.pending:      # receives complete in posting order per peer
            with self.f.lock:
                q = self.f.queues[(peer, self.rank)]
                if not self.f.lock.wait_for(lambda: len(q) > 0, timeout=30):
                    return 1
                msg = q.popleft()
            if len(msg) != nbytes:
                self.f.mismatch.append((peer, self.rank, len(msg), nbytes))
                return 1
            C.memmove(buf, msg, nbytes)
        self.pending = []
        return 0


def _cloud(pair, n):
    rng = np.random.default_rng(1000 + pair)
    rec = rng.integers(0, 256, (n, REC), dtype=np.uint8)
    rec[:, :4] = np.frombuffer(np.int32(pair).tobytes(), np.uint8)   # the pair id is readable in every record
    return rec


def run_world(world, owners, counts, order=None, root=0, capacity=None, n_pairs=None, bad_rank=None):
    """owners[p] = rank (or list of ranks, or None) holding pair p; order[rank] = the order a rank lists its pairs in."""
    lib = _lib.load()
    P = len(owners) if n_pairs is None else n_pairs
    fabric = Fabric(world)
    total = sum(c for o, c in zip(owners, counts) if o is not None)
    cap = total if capacity is None else capacity
    out = np.zeros((max(cap, 1), REC), np.uint8)
    offs = (C.c_int64 * (P + 1))()
    status, errors = [None] * world, [None] * world

    def rank_main(rank):
        mock = MockRank(fabric, rank)
        h = C.c_void_p()
        assert lib.rsm_comm_create_transport(C.byref(h), C.byref(mock.tp), rank, world) == 0
        mine = [p for p, o in enumerate(owners) if o is not None and (rank in o if isinstance(o, (list, tuple)) else o == rank)]
        if order and rank in order:
            mine = list(order[rank])
        clouds = [_cloud(p, counts[p]) for p in mine]
        n = len(mine)
        ids = (C.c_int * max(n, 1))(*mine)
        if bad_rank == rank:
            ids[0] = P + 5                                     # an out-of-range pair id on this rank only
        ptrs = (C.c_void_p * max(n, 1))(*[c.ctypes.data if len(c) else None for c in clouds])
        cnts = (C.c_int64 * max(n, 1))(*[len(c) for c in clouds])
        status[rank] = lib.rsm_gather_clouds(h, root, n, ids, ptrs, cnts, P,
                                             C.c_void_p(out.ctypes.data) if rank == root else None,
                                             C.c_int64(cap if rank == root else 0), offs if rank == root else None)
        errors[rank] = (lib.rsm_comm_last_error(h) or b"").decode()
        lib.rsm_comm_destroy(h)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
        assert not t.is_alive(), "a rank is still waiting: the gather hung"
    return status, errors, out, list(offs), fabric


@pytest.mark.parametrize("world", [2, 3, 8])
def test_out_of_order_pair_ids_empty_clouds_and_unowned_pairs(world):
    rng = np.random.default_rng(world)
    P = 3 * world + 1
    owners = [int(rng.integers(0, world)) for _ in range(P)]
    owners[2] = None                                           # nobody holds pair 2: an empty slot
    counts = [int(rng.integers(1, 40)) for _ in range(P)]
    counts[1] = 0                                              # an empty cloud
    counts[2] = 0
    order = {}
    for r in range(world):                                     # every rank lists its pairs in a shuffled (work-queue) order
        mine = [p for p, o in enumerate(owners) if o == r]
        order[r] = list(rng.permutation(mine)) if mine else []
    root = world - 1 if world == 3 else 0
    status, errors, out, offs, fabric = run_world(world, owners, counts, order=order, root=root)
    assert status == [0] * world, errors
    assert not fabric.mismatch
    assert offs[0] == 0 and offs[-1] == sum(counts)
    for p in range(P):
        got = out[offs[p]:offs[p + 1]]
        want = _cloud(p, counts[p]) if owners[p] is not None else np.zeros((0, REC), np.uint8)
        assert np.array_equal(got, want), "pair %d landed in the wrong slot" % p
    assert all(len(q) == 0 for q in fabric.queues.values())    # nothing sent that nobody received


@pytest.mark.parametrize("world,claimants", [(2, (0, 1)), (3, (0, 1)), (8, (0, 1)), (8, (3, 6)), (3, (2, 2))])
def test_a_pair_claimed_twice_fails_on_every_rank_without_a_hang(world, claimants):
    # ranks 0 and 1 both claiming a pair used to decode as "owner = rank 2" (sum of rank + 1); (2, 2) = one rank
    # listing the same pair twice
    P = 5
    owners = [0, list(claimants), world - 1, 0, 1 % world]
    counts = [7, 9, 3, 0, 11]
    if claimants[0] == claimants[1]:
        order = {claimants[0]: [p for p, o in enumerate(owners) if o == claimants[0]] + [1, 1]}
        owners[1] = None
    else:
        order = None
    status, errors, out, offs, fabric = run_world(world, owners, counts, order=order, capacity=200)
    assert status == [_lib.RSM_E_INVALID] * world
    assert all("two ranks" in e for e in errors), errors
    assert not out.any() and all(len(q) == 0 for q in fabric.queues.values())    # no payload moved


@pytest.mark.parametrize("world", [2, 8])
def test_a_bad_argument_on_one_rank_fails_every_rank_together(world):
    owners = [r % world for r in range(world + 2)]
    counts = [5] * len(owners)
    status, errors, out, _, fabric = run_world(world, owners, counts, bad_rank=world - 1)
    assert status == [_lib.RSM_E_INVALID] * world
    assert "this rank" not in errors[0] and "bad argument" in errors[0] and "bad argument" in errors[world - 1]
    assert not out.any() and all(len(q) == 0 for q in fabric.queues.values())


def test_root_capacity_too_small_fails_every_rank_before_the_payload():
    owners, counts = [0, 1, 2, 1], [10, 20, 30, 40]
    status, errors, out, _, fabric = run_world(3, owners, counts, capacity=99)
    assert status == [_lib.RSM_E_INVALID] * 3 and all("capacity" in e for e in errors)
    assert all(len(q) == 0 for q in fabric.queues.values())
    status, _, out, offs, _ = run_world(3, owners, counts, capacity=100)
    assert status == [0, 0, 0] and offs == [0, 10, 30, 60, 100]


def test_plan_orders_both_sides_by_pair_id():
    """rsm_gather_meta_fill + rsm_gather_plan alone (no transport): for random assignments the k-th send of a peer and
    the k-th receive the root posts for that peer are the same pair with the same size."""
    lib = _lib.load()
    rng = np.random.default_rng(7)
    for trial in range(50):
        world = int(rng.integers(1, 9))
        P = int(rng.integers(0, 30))
        root = int(rng.integers(0, world))
        owners = rng.integers(-1, world, P)
        counts = rng.integers(0, 100, P)
        words = 3 * P + 2
        metas, locals_ = [], []
        for r in range(world):
            mine = [int(p) for p in rng.permutation(np.nonzero(owners == r)[0])]
            ids = (C.c_int * max(len(mine), 1))(*mine)
            cnts = (C.c_int64 * max(len(mine), 1))(*[int(counts[p]) for p in mine])
            meta = (C.c_int64 * words)()
            assert lib.rsm_gather_meta_fill(r, world, root, len(mine), ids, cnts, P, C.c_int64(10 ** 9), meta) == 0
            metas.append(np.array(meta[:]))
            locals_.append((mine, ids, cnts))
        summed = (C.c_int64 * words)(*[int(v) for v in sum(metas)])
        sends, recvs = collections.defaultdict(list), collections.defaultdict(list)
        for r in range(world):
            mine, ids, cnts = locals_[r]
            ops = (GatherOp * (P + len(mine) + 1))()
            n_ops = C.c_int()
            offs = (C.c_int64 * (P + 1))()
            assert lib.rsm_gather_plan(r, world, root, len(mine), ids, cnts, P, summed, offs, ops, len(ops), C.byref(n_ops)) == 0
            for o in ops[:n_ops.value]:
                assert o.count > 0 and o.offset == offs[o.pair]
                if o.kind == 0:
                    assert r != root and o.peer == root and mine[o.local_index] == o.pair
                    sends[r].append((o.pair, o.count))
                elif o.kind == 1:
                    assert r == root and owners[o.pair] == o.peer != root
                    recvs[o.peer].append((o.pair, o.count))
                else:
                    assert r == root and mine[o.local_index] == o.pair
        assert dict(sends) == dict(recvs)
        for r, s in sends.items():
            assert [p for p, _ in s] == sorted(p for p, _ in s)


def test_gather_counts_over_the_mock():
    lib = _lib.load()
    world, P = 3, 5
    owners, counts = [2, 0, 1, None, 0], [4, 0, 9, 0, 2]
    fabric = Fabric(world)
    got, status = [None] * world, [None] * world

    def rank_main(rank):
        mock = MockRank(fabric, rank)
        h = C.c_void_p()
        assert lib.rsm_comm_create_transport(C.byref(h), C.byref(mock.tp), rank, world) == 0
        mine = [p for p, o in enumerate(owners) if o == rank][::-1]
        ids = (C.c_int * max(len(mine), 1))(*mine)
        cnts = (C.c_int64 * max(len(mine), 1))(*[counts[p] for p in mine])
        out = (C.c_int64 * P)()
        status[rank] = lib.rsm_gather_counts(h, len(mine), ids, cnts, P, out)
        got[rank] = list(out)
        lib.rsm_comm_destroy(h)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=60) for t in threads]
    assert status == [0, 0, 0] and got == [counts] * world
