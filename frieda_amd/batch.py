"""Batch data-parallelism over independent blobs: one process per GPU, blob i -> rank i mod world, no
data-path collective; the only exchange is an all_gather of the 32-byte commitment roots (RCCL over xGMI with the nccl
backend; gloo on CPU in the tests).

`commit_fn(blob) -> bytes[32]` is injected so that the sharding / gather logic is testable without a GPU; the default is the
HIP path of this package (there is no CPU fallback): a rank's shard goes through the C ABI's multi entry on its one device
(`frieda_commit_many` / `frieda_prove_many`: runs of equal-length blobs through the batched kernels, cut into calls by the library's
batch policy, two calls in flight, host blobs uploaded ahead of their kernels).
"""
import torch
import torch.distributed as dist


_multi = {}


def _multi_for(device):
    """The C ABI's multi-GPU handle (frieda_multi_create) over this process's one device: blobs of mixed lengths go through
    frieda_commit_many / frieda_prove_many, which keeps two proofs in flight per device."""
    from . import api

    if device not in _multi:
        _multi[device] = api.MultiContext([device])
    return _multi[device]


def release_workspaces():
    """The handles cached above keep their device workspaces between calls (up to two batch budgets per device: that is what makes a
    second call cheap).  A process that is done with batches for a while hands the memory back with this; the next call allocates again."""
    for mc in _multi.values():
        mc.release_workspace()


def close():
    """Destroys the cached handles (workspaces, streams, RCCL communicators).  Registered with atexit."""
    while _multi:
        _, mc = _multi.popitem()
        mc.close()


import atexit  # noqa: E402

atexit.register(close)


def commit_many_on_node(blobs, log_blowup_factor, devices=None):
    """ONE process, every GPU of the node, through the C ABI (frieda_commit_many): blob i -> devices[i mod n]; the roots are gathered
    with ncclAllGather on a single-process RCCL communicator.  The alternative to one process per GPU + torch.distributed below."""
    from . import api

    devs = list(range(torch.cuda.device_count())) if devices is None else list(devices)
    mc = api.MultiContext(devs)
    try:
        return mc.commit_many(blobs, log_blowup_factor)
    finally:
        mc.close()


def prove_many_on_node(blobs, seeds, pcs_config, devices=None):
    """frieda_prove_many over the GPUs of the node from one process -> [(commitment, proof)] in blob order."""
    from . import api

    devs = list(range(torch.cuda.device_count())) if devices is None else list(devices)
    mc = api.MultiContext(devs)
    try:
        return mc.prove_many(blobs, seeds, pcs_config)
    finally:
        mc.close()


def _local_commit_many(ctx, blobs, log_blowup_factor):
    # more than one blob: the C ABI's multi entry on this one device (frieda_commit_many: runs of equal-length blobs go through the
    # batched kernels, cut into calls by the library's batch policy, two calls in flight, uploads ahead of their kernels)
    if len(blobs) > 1:
        return _multi_for(ctx.device).commit_many(blobs, log_blowup_factor)
    return [ctx.commit(b, log_blowup_factor) for b in blobs]


def _local_prove_many(ctx, blobs, seeds, pcs_config):
    if len(blobs) > 1:  # frieda_prove_many: batch policy for runs of equal lengths, single proofs otherwise, two calls in flight
        return _multi_for(ctx.device).prove_many(blobs, seeds, pcs_config)
    return [ctx.commit_and_generate_proof(b, None if seeds is None else seeds[i], pcs_config) for i, b in enumerate(blobs)]


def shard_indices(n_blobs, rank, world):
    """Blob indices owned by `rank`: round-robin (blob i -> rank i mod world)."""
    return list(range(rank, n_blobs, world))


def commit_batch(blobs, log_blowup_factor, commit_fn=None, device=None):
    """Every rank passes the same list of blobs (or at least its own entries); returns the list of all roots, in blob order,
    on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(blobs)
    mine = shard_indices(n, rank, world)
    if commit_fn is None:
        from . import api

        mine_roots = iter(_local_commit_many(api.default_context(), [blobs[i] for i in mine], log_blowup_factor))
        commit_fn = lambda b: next(mine_roots)  # noqa: E731
    per_rank = (n + world - 1) // world
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")
    local = torch.zeros(per_rank * 32, dtype=torch.uint8, device=device)
    for slot, i in enumerate(mine):
        root = commit_fn(blobs[i])
        assert len(root) == 32
        local[32 * slot : 32 * slot + 32] = torch.frombuffer(bytearray(root), dtype=torch.uint8).to(device)
    if world > 1:
        gathered = torch.zeros(world * per_rank * 32, dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(gathered, local)
    else:
        gathered = local
    flat = gathered.cpu().numpy().tobytes()
    roots = [None] * n
    for r in range(world):
        for slot, i in enumerate(shard_indices(n, r, world)):
            off = (r * per_rank + slot) * 32
            roots[i] = flat[off : off + 32]
    return roots


def gather_rank_roots(local_roots, device=None):
    """bench.py's exchange: every rank holds the K roots of the K blobs it processed (K * 32 bytes, bytes-like or a uint8 tensor
    already on `device`); one all_gather hands every rank every rank's roots.  Returns a uint8 tensor [world, K * 32] on `device`."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")
    if isinstance(local_roots, torch.Tensor):
        local = local_roots.to(device)
    else:
        local = torch.frombuffer(bytearray(local_roots), dtype=torch.uint8).to(device)
    assert local.numel() % 32 == 0
    if world > 1:
        gathered = torch.zeros(world * local.numel(), dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(gathered, local)
    else:
        gathered = local.clone()
    return gathered.view(world, local.numel())


def prove_batch(blobs, seeds, pcs_config, prove_fn=None, device=None):
    """Sharded `commit_and_generate_proof`: rank r proves blobs r, r + world, ...; every rank gets all commitment roots (one
    all_gather of 32 bytes per blob slot) and the proofs of its own shard as {blob index: proof}.

    `prove_fn(blob, seed) -> (bytes[32], proof)` is injectable for the CPU test; the default is the HIP path (batched kernels
    for equal-length shards)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(blobs)
    mine = shard_indices(n, rank, world)
    results = {}
    if prove_fn is None:
        from . import api

        my_seeds = None if seeds is None else [seeds[i] for i in mine]
        results = dict(zip(mine, _local_prove_many(api.default_context(), [blobs[i] for i in mine], my_seeds, pcs_config)))
    else:
        for i in mine:
            results[i] = prove_fn(blobs[i], seeds[i] if seeds is not None else None)
    roots = commit_batch(blobs, pcs_config.fri_config.log_blowup_factor if hasattr(pcs_config, "fri_config") else 4,
                         commit_fn=lambda b, _it=iter([results[i][0] for i in mine]): next(_it), device=device)
    return roots, {i: results[i][1] for i in mine}
