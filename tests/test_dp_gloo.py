"""CPU, world_size 2 over gloo: the data-parallel pieces of the step -- batch sharding, the flat-arena
gradient all-reduce with the 1/world factor folded into grad_scale, and metric gathering -- give the
single-process result.  (Gradients come from the oracle here: tests may use it as the checker.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import tier_a as O
        from ultrafnd_git_amd.arena import FlatArena
        from ultrafnd_git_amd.dp import GradReducer, gather_rows, shard_indices
        torch.set_num_threads(2)
        B = 8
        fus, clf = O.seeded_params(1234)
        full = O.seeded_batch(21, B)
        idx = shard_indices(B, world, rank)
        assert idx.numel() == B // world
        shard = {k: v[idx] for k, v in full.items()}
        _, _, gf, gc = O.loss_and_grads(fus, clf, shard)
        grads = {**{"fusion." + k: g for k, g in gf.items()}, **{"clf." + k: g for k, g in gc.items()}}
        keys = [k for k, g in grads.items() if g is not None]
        arena = FlatArena([[(k, tuple(grads[k].shape))] for k in keys], [], torch.device("cpu"))
        arena.ensure_grad()
        for k in keys:
            arena.grad_view(k).copy_(grads[k])
        red = GradReducer(arena.grad)
        assert red.world == world and abs(red.grad_scale - 1.0 / world) < 1e-12
        red.start(); red.finish()
        mean = arena.grad * red.grad_scale
        if rank == 0:
            _, _, gf0, gc0 = O.loss_and_grads(fus, clf, full)
            ref = {**{"fusion." + k: g for k, g in gf0.items()}, **{"clf." + k: g for k, g in gc0.items()}}
            worst = 0.0
            for k in keys:
                o, shape = arena.offsets[k]
                got = mean[o:o + ref[k].numel()].view(shape)
                worst = max(worst, (got - ref[k]).abs().max().item() / max(1e-6, ref[k].abs().max().item()))
            rows = gather_rows(torch.arange(4.0).view(4, 1) + 10 * rank)
            q.put(("ok", worst, rows.flatten().tolist()))
        else:
            gather_rows(torch.arange(4.0).view(4, 1) + 10 * rank)
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, worst, rows = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert status == "ok", worst
    assert worst <= 2e-5, worst          # mean of the two shard-mean gradients == full-batch mean gradient
    assert rows == [0.0, 1.0, 2.0, 3.0, 10.0, 11.0, 12.0, 13.0]


def test_shard_indices_partition_and_padding():
    from ultrafnd_git_amd.dp import shard_indices
    for n, world in ((10, 2), (11, 4), (3, 8), (32, 8)):
        per = [shard_indices(n, world, r) for r in range(world)]
        assert len({p.numel() for p in per}) == 1 and per[0].numel() == -(-n // world)
        allidx = torch.cat(per)
        assert set(allidx.tolist()) == set(range(n))          # every sample seen, wrap-around padding only
    perm = torch.randperm(9, generator=torch.Generator().manual_seed(0))
    a, b = shard_indices(9, 2, 0, perm), shard_indices(9, 2, 1, perm)
    assert a.tolist() == perm.tolist()[0::2] + [] and b.tolist() == (perm.tolist() + perm.tolist()[:1])[1::2]
