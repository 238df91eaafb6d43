"""Data-parallel layer on CPU: world_size 2 over gloo (the N>1 path of bench.py / the trainers, minus RCCL).
Kernels are the oracle-backed test stand-ins (tests/cpu_kernels.py); what is under test is the host logic of
surfacenetworks_amd.dp: sharding, flat gradient bucket, SUM all-reduce with global-batch loss normalisation,
parameter broadcast — and the property SURVEY.md §8e asks for: sum of shard gradients == full-batch gradient."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SN_DP_FORCE_CPU="1")
    torch.set_num_threads(2)
    import cpu_kernels

    cpu_kernels.install()
    from helpers import deterministic_init
    from surfacenetworks_amd import arap, dp

    r, lr, w, dev = dp.init_distributed("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    ds = arap.ClothSequences([(6, 5), (7, 7), (5, 6), (8, 5)], frames=44, op_frames=2, seed=5, device="cpu", model="dir")
    G = 4                                                     # global batch: one sample of every sequence
    seq, off = np.arange(G), np.zeros(G, dtype=np.int64)

    def make_model(seed):
        m = deterministic_init(arap.DirModel(), seed)
        m.eval()                                              # BN frozen: shards are then exactly additive
        return m

    # rank-dependent init, then broadcast from rank 0 -> identical replicas
    model = make_model(3 + rank)
    dp.broadcast_parameters(model, 0)
    ref = make_model(3)
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k

    bucket = dp.FlatGradBucket(model.parameters())
    assert bucket.nbytes == 4 * 1018872 and bucket.check_views()
    mine = dp.shard_round_robin(G, rank, world)
    batch = ds.sample_batch(len(mine), None, seq_ids=seq[mine], offsets=off[mine])
    opt = arap.make_optimizer(model)
    # variant 1: gradients stored (not accumulated) by the backward, packed into the bucket and all-reduced by sync()
    bucket.detach_grads()
    l1, _ = arap.forward_loss(model, batch, G)
    l1.backward()
    assert all(p.grad is not None and p.grad.untyped_storage().data_ptr() != bucket.flat.untyped_storage().data_ptr()
               for p in model.parameters())
    bucket.sync()
    assert bucket.check_views()
    g_sync = bucket.flat.clone()
    # variant 2: zero the bucket, accumulate into its slices, all-reduce in place
    loss = arap.train_step(model, opt, batch, global_batch=G, grad_sync=bucket.all_reduce)
    assert bucket.check_views()                               # zero_grad(set_to_none=False) keeps the views alive
    g_dp = bucket.flat.clone()
    assert torch.allclose(g_sync, g_dp, rtol=1e-6, atol=1e-9)

    # single-process full batch on the same data
    full = ds.sample_batch(G, None, seq_ids=seq, offsets=off)
    l_full, _ = arap.forward_loss(ref, full, G)
    l_full.backward()
    g_full = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    err = ((g_dp - g_full).norm() / g_full.norm()).item()
    # losses: sum over ranks of shard losses == full loss
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    # replicas stay identical after the optimizer step
    flat_p = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    other = flat_p.clone()
    dist.broadcast(other, 0)
    q.put((rank, err, abs(lsum.item() - l_full.item()) / abs(l_full.item()), torch.equal(other, flat_p), len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_gradients_sum_to_full_batch_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    res = sorted(q.get(timeout=10) for _ in range(2))
    for rank, err, lerr, same, n in res:
        assert n == 2
        assert err < 1e-5, f"rank {rank}: all-reduced shard gradients differ from the full-batch gradient by {err:.2e}"
        assert lerr < 1e-6 and same


def _worker_syncbn(rank, world, port, q):
    """Train-mode BatchNorm with synchronised statistics: two replicas of half the batch == one process on the full batch."""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SN_DP_FORCE_CPU="1")
    torch.set_num_threads(2)
    import cpu_kernels

    cpu_kernels.install()
    from helpers import deterministic_init
    from surfacenetworks_amd import arap, dp

    dp.init_distributed("gloo")
    # equal-size meshes with >= 32 vertices so that the half-width global-average stages run too
    ds = arap.ClothSequences([(6, 6)] * 4, frames=44, op_frames=2, seed=9, device="cpu", model="dir")
    G = 4
    seq, off = np.arange(G), np.zeros(G, dtype=np.int64)
    model = deterministic_init(arap.DirModel(), 3).train()
    ref = deterministic_init(arap.DirModel(), 3).train()
    bucket = dp.FlatGradBucket(model.parameters())
    mine = dp.shard_round_robin(G, rank, world)
    batch = ds.sample_batch(len(mine), None, seq_ids=seq[mine], offsets=off[mine])
    dp.sync_batchnorm(True)
    loss, _ = arap.forward_loss(model, batch, G)
    loss.backward()
    bucket.all_reduce()
    g_dp = bucket.flat.clone()
    dp.sync_batchnorm(False)
    full = ds.sample_batch(G, None, seq_ids=seq, offsets=off)
    l_full, _ = arap.forward_loss(ref, full, G)
    l_full.backward()
    g_full = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    err = ((g_dp - g_full).norm() / g_full.norm()).item()
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    rv = max(((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
             for (ka, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()) if "running" in ka)
    q.put((rank, err, abs(lsum.item() - l_full.item()) / abs(l_full.item()), rv))
    dist.barrier()
    dist.destroy_process_group()


def test_synchronised_batchnorm_world2_equals_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_syncbn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    for rank, err, lerr, rv in sorted(q.get(timeout=10) for _ in range(2)):
        assert err < 1e-5, f"rank {rank}: sync-BN shard gradients differ from the full-batch gradient by {err:.2e}"
        assert lerr < 1e-6 and rv < 1e-5


def test_sharding_helpers():
    from surfacenetworks_amd import dp

    parts = [dp.shard_round_robin(11, r, 4) for r in range(4)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(11))
    assert parts[1].tolist() == [1, 5, 9]
    w = np.array([20000, 1000, 1500, 19000, 7000, 7200, 300, 12000], dtype=float)
    bins = [dp.shard_balanced(w, r, 3) for r in range(3)]
    assert sorted(np.concatenate(bins).tolist()) == list(range(8))
    loads = [w[b].sum() for b in bins]
    assert max(loads) <= 4 / 3 * w.sum() / 3 + 1            # LPT bound
    assert all(np.array_equal(dp.shard_balanced(w, r, 3), bins[r]) for r in range(3))     # deterministic


def test_flat_bucket_single_process():
    from surfacenetworks_amd import dp

    lin = torch.nn.Linear(5, 3)
    b = dp.FlatGradBucket(lin.parameters())
    assert b.flat.numel() == 18 and b.all_reduce() is None
    lin(torch.ones(2, 5)).sum().backward()
    assert torch.equal(b.flat[:15].view(3, 5), lin.weight.grad) and b.flat[15:].tolist() == [2.0, 2.0, 2.0]
    torch.optim.SGD(lin.parameters(), 0.1).zero_grad(set_to_none=False)
    assert b.check_views() and float(b.flat.abs().sum()) == 0.0


def _worker_pairs(rank, world, port, q):
    """BASELINE config 4: dense correspondence, one pair per rank and step, gradients all-reduced."""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SN_DP_FORCE_CPU="1")
    torch.set_num_threads(2)
    import cpu_kernels

    cpu_kernels.install()
    from helpers import deterministic_init
    from surfacenetworks_amd import dense_correspondence as dc, dp

    dp.init_distributed("gloo")
    ds = dc.TorusBodies(3, n=7, m=9, pad_to=64, seed=5, device="cpu")          # every rank holds the (tiny) dataset
    model = deterministic_init(dc.SiameseModel("lap", 3), 11 + rank).eval()      # BN frozen: pairs are exactly additive
    dp.broadcast_parameters(model, 0)
    ref = deterministic_init(dc.SiameseModel("lap", 3), 11).eval()
    bucket = dp.FlatGradBucket(model.parameters())
    opt = dc.make_optimizer(model)
    pairs = [(0, 1), (1, 2)]
    ia, ib = pairs[rank]
    loss = dc.train_step(model, opt, ds, ia, ib, grad_sync=bucket.sync, global_pairs=world, zero_grads=bucket.detach_grads)
    g_dp = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    total = sum(dc.forward_pair_loss(ref, ds, a, b) for a, b in pairs) / len(pairs)
    total.backward()
    g_full = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    err = ((g_dp - g_full).norm() / g_full.norm()).item()
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    flat_p = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    other = flat_p.clone()
    dist.broadcast(other, 0)
    q.put((rank, err, abs(lsum.item() - total.item()) / abs(total.item()), torch.equal(other, flat_p)))
    dist.barrier()
    dist.destroy_process_group()


def test_correspondence_pairs_sharded_over_two_ranks():
    """Sum over ranks of the per-pair gradients (loss / number of pairs) == gradient of the mean loss over both pairs; the
    replicas stay identical after the optimizer step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pairs, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    for rank, err, lerr, same in sorted(q.get(timeout=10) for _ in range(2)):
        assert err < 1e-5, f"rank {rank}: all-reduced pair gradients differ from the two-pair gradient by {err:.2e}"
        assert lerr < 1e-6 and same


def _worker8(rank, world, port, q):
    """World size 8 on a RAGGED pool sharded by shard_balanced (size-balanced bins by entry count, what DESIGN §7 prescribes
    for ragged batches): the sum of the eight shard gradients is the full-batch gradient."""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SN_DP_FORCE_CPU="1")
    torch.set_num_threads(1)
    import cpu_kernels

    cpu_kernels.install()
    from helpers import deterministic_init
    from surfacenetworks_amd import arap, dp

    r, lr, w, dev = dp.init_distributed("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    grids = [(4, 4), (7, 6), (5, 4), (9, 5), (4, 5), (6, 6), (8, 4), (5, 5), (10, 6), (4, 6), (7, 4)]
    ds = arap.ClothSequences(grids, frames=44, op_frames=2, seed=7, device="cpu", model="dir", permute="both")
    G = len(grids)
    seq, off = np.arange(G), np.zeros(G, dtype=np.int64)
    weights = ds.pool_Di._fwd["cnt"][::ds.op_frames]                      # entries of every mesh's operator
    mine = dp.shard_balanced(weights, rank, world)
    owners = [dp.shard_balanced(weights, r_, world) for r_ in range(world)]
    assert sorted(np.concatenate(owners).tolist()) == list(range(G)) and all(len(o) >= 1 for o in owners)
    loads = np.array([weights[o].sum() for o in owners])
    assert loads.max() <= loads.mean() + weights.max()                    # LPT bound
    model = deterministic_init(arap.DirModel(), 3 + rank).eval()
    dp.broadcast_parameters(model, 0)
    bucket = dp.FlatGradBucket(model.parameters())
    opt = arap.make_optimizer(model)
    batch = ds.sample_batch(len(mine), None, seq_ids=seq[mine], offsets=off[mine])
    loss = arap.train_step(model, opt, batch, global_batch=G, grad_sync=bucket.sync, zero_grads=bucket.detach_grads)
    g_dp = bucket.flat.clone()
    err = lerr = 0.0
    if rank == 0:
        ref = deterministic_init(arap.DirModel(), 3).eval()
        l_full, _ = arap.forward_loss(ref, ds.sample_batch(G, None, seq_ids=seq, offsets=off), G)
        l_full.backward()
        g_full = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        err = ((g_dp - g_full).norm() / g_full.norm()).item()
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    if rank == 0:
        lerr = abs(lsum.item() - l_full.item()) / abs(l_full.item())
    flat_p = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    other = flat_p.clone()
    dist.broadcast(other, 0)
    q.put((rank, err, lerr, bool(torch.equal(other, flat_p)), len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_gradients_sum_to_full_batch_world8_balanced_ragged():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    res = sorted(q.get(timeout=10) for _ in range(8))
    assert sum(r[4] for r in res) == 11
    assert res[0][1] < 2e-5 and res[0][2] < 1e-5, res[0]
    assert all(r[3] for r in res)


def test_rank_affinity_without_a_numa_node(tmp_path):
    """bench.cpus_for_rank / gpu_numa_node: a device whose sysfs numa_node is -1 or missing, a node without a cpulist, more
    ranks than CPUs — every rank still gets a non-empty CPU set, disjoint where the CPUs suffice."""
    import bench

    cpus = list(range(16))
    shares = [bench.cpus_for_rank(-1, cpus, r, 8)[0] for r in range(8)]
    assert all(len(s) == 2 for s in shares) and sorted(sum(shares, [])) == cpus
    assert "no numa node" in bench.cpus_for_rank(-1, cpus, 3, 8)[1]
    few = [bench.cpus_for_rank(-1, [0, 1, 2, 3], r, 8)[0] for r in range(8)]
    assert all(len(s) >= 1 for s in few) and few[0] == [0] and few[7] == [0, 1, 2, 3]
    # a node whose cpulist exists: its CPUs (intersected with the allowed set); one whose cpulist is missing: dealt out
    node_dir = tmp_path / "devices/system/node/node1"
    node_dir.mkdir(parents=True)
    (node_dir / "cpulist").write_text("4-7,12-13\n")
    got, how = bench.cpus_for_rank(1, cpus, 5, 8, sysfs=str(tmp_path))
    assert got == [4, 5, 6, 7, 12, 13] and how.startswith("numa node 1")
    assert bench.cpus_for_rank(1, [0, 1], 0, 2, sysfs=str(tmp_path))[0] == [0]         # node CPUs not allowed here: dealt out
    assert bench.cpus_for_rank(2, cpus, 1, 8, sysfs=str(tmp_path))[0] == [2, 3]
    assert bench.gpu_numa_node(0, sysfs=str(tmp_path)) == -1                            # no such device entry / no HIP device
