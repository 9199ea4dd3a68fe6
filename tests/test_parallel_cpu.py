"""N>1 path on CPU: world_size-2 gloo processes exercise the stream sharding + fixed-stride result gather that
bench.py / a multi-GPU eval uses over RCCL."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from unicorn_amd.parallel import ROW, gather_byte_strings, gather_result_rows, shard_streams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    streams = shard_streams(5, world, rank)                       # 5 streams over 2 ranks: ragged shards
    rows = []
    for s in streams:
        for f in range(3 + s):                                    # ragged number of frames per stream
            rows.append([s, f, 1, 10 * s, f, 10 * s + 5, f + 5, 0.5 + 0.01 * f])
    rows = torch.tensor(rows, dtype=torch.float32).reshape(-1, ROW)
    out = gather_result_rows(rows)
    empty = gather_result_rows(torch.zeros((0, ROW)) if rank == 1 else rows)     # a rank with no rows
    # variable-length payloads (mask RLE strings): ragged item counts AND ragged item lengths, one rank may have none
    mine = [bytes([65 + rank]) * (3 + 5 * i + rank) for i in range(4 - 3 * rank)]
    strs = gather_byte_strings(mine)
    none = gather_byte_strings([] if rank == 0 else mine)
    q.put((rank, streams, out.clone(), empty.shape[0], rows.shape[0], strs, none))
    dist.barrier()
    dist.destroy_process_group()


def test_stream_sharding_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]          # stream i -> rank i % world
    a, b = res[0][2], res[1][2]
    assert torch.equal(a, b)                                       # every rank sees the same table
    assert a.shape == (sum(3 + s for s in range(5)), 8)
    key = a[:, 0] * 100 + a[:, 1]
    assert torch.equal(key, torch.sort(key)[0])                    # ordered by (stream, frame)
    assert res[0][3] == res[0][4] and res[1][3] == res[0][4]       # ragged: one rank contributed zero rows
    exp = [[b"A" * (3 + 5 * i) for i in range(4)], [b"B" * 4]]
    assert res[0][5] == exp and res[1][5] == exp                   # byte strings: same nested list on every rank, rank order
    assert res[0][6] == [[], [b"B" * 4]] and res[1][6] == [[], [b"B" * 4]]


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from unicorn_amd.parallel import gather_result_rows
    r = torch.tensor([[1, 2, 0, 0, 0, 1, 1, .5], [0, 1, 0, 0, 0, 1, 1, .5]])
    out = gather_result_rows(r)
    assert out[0, 0] == 0 and out[1, 0] == 1
