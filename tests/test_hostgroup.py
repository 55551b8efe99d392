"""The torch-free host process group (pyflwdir_amd/hostgroup.py): three real processes over TCP."""
import multiprocessing as mp
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from pyflwdir_amd.hostgroup import HostGroup

    g = HostGroup(rank, world, "127.0.0.1", port, timeout=60)
    parts = g.allgather(bytes([rank]) * (rank + 1))
    b = g.bcast(b"unique-id" if rank == 0 else b"", 0)
    mx = g.allreduce(float(rank) + 0.5, "max")
    sm = g.allreduce(rank + 1, "sum")
    mn = g.allreduce(rank + 1, "min")
    big = g.allgather(bytes([65 + rank]) * 1_500_000)  # a boundary record of 90000 columns is 1.4 MB
    g.barrier()
    g.close()
    q.put((rank, parts, b, mx, sm, mn, [len(x) for x in big], [x[:1] for x in big]))


def test_three_process_tcp_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, parts, b, mx, sm, mn, lens, heads in res:
        assert parts == [bytes([r]) * (r + 1) for r in range(world)]
        assert b == b"unique-id" and mx == world - 0.5 and sm == 6 and mn == 1
        assert lens == [1_500_000] * world and heads == [b"A", b"B", b"C"]


def test_single_rank_group_needs_no_socket():
    sys.path.insert(0, ROOT)
    from pyflwdir_amd.hostgroup import HostGroup

    g = HostGroup(0, 1)
    assert g.allgather(b"x") == [b"x"] and g.bcast(b"y") == b"y" and g.allreduce(3, "min") == 3
    g.barrier()
