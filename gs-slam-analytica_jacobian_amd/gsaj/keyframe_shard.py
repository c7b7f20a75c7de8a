"""Multi-GPU sharding of the mapping window: independent keyframes of one shared Gaussian map.

The reference is single-GPU; its mapping step renders every keyframe of the window, sums the
losses and back-propagates once, so per-Gaussian gradients ACCUMULATE over keyframes while each
keyframe's dL/dtau stays its own (utils/slam_backend.py:168-232, SURVEY 8e partitioning A).
On an 8-GPU MI355X node: replicate the Gaussians, deal keyframes round-robin over the ranks
(one process per GPU), run forward + analytical backward locally, then
  * ONE all-reduce(sum) of the flat per-Gaussian gradient bucket (the kernels write straight into
    it -- no packing pass); RCCL over xGMI, backend "nccl" (= RCCL on ROCm), "gloo" in CPU tests;
  * ONE all-gather of the [k_local, 6] pose gradients (each keyframe's dL/dtau belongs to it).
No collective sits inside the per-frame data path.
"""
import torch
import torch.distributed as dist

# per-Gaussian gradient fields of the bucket, in order, as (name, floats per Gaussian)
def bucket_layout(M, has_scales=True):
    fields = [("mean3D", 3), ("sh", 3 * M), ("opacity", 1)]
    if has_scales:
        fields += [("scale", 3), ("rot", 4)]
    else:
        fields += [("cov3D", 6)]
    return fields


def bucket_views(bucket, P, M, has_scales=True, n_keyframes=0):
    """Split a flat tensor into per-field [P, w] views (field-major, so each field is one contiguous
    region the kernels can write).  With n_keyframes > 0 the bucket ends with a [n_keyframes, 6] block
    "tau_all": every rank writes the dL/dtau of its own keyframes there and leaves the other rows
    zero, so the SAME sum all-reduce also all-gathers the pose gradients (no second collective)."""
    out, off = {}, 0
    for name, w in bucket_layout(M, has_scales):
        out[name] = bucket[off:off + P * w].view(P, w)
        off += P * w
    if n_keyframes:
        out["tau_all"] = bucket[off:off + 6 * n_keyframes].view(n_keyframes, 6)
        off += 6 * n_keyframes
    assert off == bucket.numel()
    return out


def bucket_numel(P, M, has_scales=True, n_keyframes=0):
    return P * sum(w for _, w in bucket_layout(M, has_scales)) + 6 * n_keyframes


def shard_keyframes(n_keyframes, world_size, rank):
    """Round-robin ownership: keyframe k belongs to rank k % world_size."""
    return list(range(rank, n_keyframes, world_size))


def allreduce_gaussian_grads(bucket, group=None, async_op=False):
    """Sum the per-Gaussian gradient bucket over all ranks in place (one collective).
    With async_op=True the work handle is returned: call .wait() before the bucket is read or
    overwritten, so the collective of one step overlaps the kernels of the next."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return work if async_op else bucket
    return None if async_op else bucket


def exchange_gaussian_grads(bucket, form="all_reduce", group=None):
    """Blocking sum of the bucket over the ranks, in one of two forms with the same result:
      "all_reduce"                  one all-reduce (what the training step uses, asynchronously);
      "reduce_scatter+all_gather"   each rank first receives the sum of ITS 1/world slice, then the slices are gathered --
                                    the two halves of a ring all-reduce as separate collectives (on point-to-point
                                    xGMI both halves can keep all seven links of a GPU busy).  The bucket is padded to a
                                    multiple of the world size in a scratch tensor.  Backends without reduce-scatter (gloo)
                                    raise RuntimeError."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bucket
    if form == "all_reduce":
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
        return bucket
    if form != "reduce_scatter+all_gather":
        raise ValueError("unknown form %r" % (form,))
    world = dist.get_world_size(group)
    n = bucket.numel()
    per = (n + world - 1) // world
    full = bucket if per * world == n else torch.cat([bucket, bucket.new_zeros(per * world - n)])
    shard = torch.empty(per, dtype=bucket.dtype, device=bucket.device)
    dist.reduce_scatter_tensor(shard, full, op=dist.ReduceOp.SUM, group=group)
    dist.all_gather_into_tensor(full, shard, group=group)
    if full is not bucket:
        bucket.copy_(full[:n])
    return bucket


def gather_pose_grads(tau_local, n_keyframes, group=None):
    """tau_local: [k_local, 6] in the order of shard_keyframes(); returns [n_keyframes, 6] with
    row k = dL/dtau of keyframe k on every rank.  Ranks may own different numbers of keyframes:
    rows are padded to the maximum and scattered back by owner."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tau_local.clone()
    world = dist.get_world_size(group)
    k_max = (n_keyframes + world - 1) // world
    pad = torch.zeros((k_max, 6), dtype=tau_local.dtype, device=tau_local.device)
    pad[: tau_local.shape[0]] = tau_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.zeros((n_keyframes, 6), dtype=tau_local.dtype, device=tau_local.device)
    for r in range(world):
        ks = shard_keyframes(n_keyframes, world, r)
        out[ks] = parts[r][: len(ks)]
    return out
