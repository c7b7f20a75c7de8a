"""Multi-GPU sharding of ONE frame by tile bands (SURVEY 8e partitioning B): the tracking loop on several GPUs.

The reference tracks on one device: ~100 sequential iterations of render -> L1 loss -> backward -> Adam step on the camera
pose (utils/slam_frontend.py:135-193), every iteration a function of the previous one -- there are no independent frames to
deal out.  What does split is the frame: every gradient of the path is a sum over pixels, so R ranks that each render a
band of tile rows (gsaj_set_tile_band: the tile rectangles of cuda_rasterizer/auxiliary.h:46-58 clipped to the band) get
the band's share of dL/dtau, and the shares add up to the whole-frame value.  Per iteration and rank:

    forward(band) -> loss seeds -> backward(pose_only)          no collective inside the data path
    all-reduce(sum) of 12 floats: dL/dtau (6), loss, L_rgb, L_depth, dL/da, dL/db, aborted     RCCL ("nccl") / gloo in CPU tests
    pose Adam step + update_pose on every rank                   same input, same arithmetic -> the replicas stay identical

The tracking loss needs no masking for this: its colour term is weighted by the rendered opacity and its depth term is
gated on opacity > 0.95 (utils/slam_utils.py:56-88), both zero outside the band, and the means are over the whole image's
pixel count.  Work that does not shrink with the band: the projection of all P Gaussians (k_preprocess, 18 us of a 210 us
cfg2 frame).  Band boundaries cost nothing extra: a Gaussian straddling two bands is binned by both ranks, exactly the
instances a single GPU would have binned for those tiles.
"""
import torch
import torch.distributed as dist

TILE = 16
# dL/dtau (6) | loss, L_rgb, L_depth, dL/da, dL/db (= the out_scalars[5] of gsaj_loss_seeds, in its order) | aborted: > 0 if a rank's
# share of this iteration was aborted on the device (its terms are stale): summed like the rest, so EVERY rank sees it in the same
# iteration and skips the pose step (gsaj_pose_adam_step's `skip` word)
REDUCED_FLOATS = 12
TAU, LOSS_TERMS, EXPOSURE_GRADS, ABORTED = slice(0, 6), slice(6, 9), slice(9, 11), slice(11, 12)


def tile_rows(H):
    return (int(H) + TILE - 1) // TILE


def uniform_bands(H, world_size):
    """[begin, end) tile rows of every rank: contiguous, ceil(rows / world) rows each, the last ranks possibly empty
    ([rows, rows): nothing to render -- gsaj_set_tile_band takes an empty band at a non-zero row)."""
    rows, world = tile_rows(H), int(world_size)
    if world <= 0:
        raise ValueError("world_size must be positive")
    per = (rows + world - 1) // world
    return [(min(rows, r * per), min(rows, (r + 1) * per)) for r in range(world)]


def balanced_bands(row_weight, world_size):
    """Contiguous bands of about equal weight.  row_weight[y]: cost of tile row y, e.g. row_work() of a whole-frame probe
    (the compositors' cost is the number of Gaussian-pixel interactions, far from uniform over the image).  The cut after
    rank r is the row boundary whose running weight is closest to (r + 1) / world of the total (ties: the earlier row);
    cuts never move backwards, so bands may be empty when one row outweighs a rank's share.  All-zero weights fall back to
    uniform_bands.  Pure integer / float host arithmetic on the same input: every rank computes the same bands."""
    w = [float(x) for x in row_weight]
    rows, world = len(w), int(world_size)
    if world <= 0:
        raise ValueError("world_size must be positive")
    if any(x < 0 for x in w):
        raise ValueError("row weights must be non-negative")
    prefix = [0.0]
    for x in w:
        prefix.append(prefix[-1] + x)
    total = prefix[-1]
    if total <= 0.0:
        return uniform_bands(rows * TILE, world)
    cuts = [0]
    for r in range(world - 1):
        target = total * (r + 1) / world
        e = cuts[-1]
        while e < rows and abs(prefix[e + 1] - target) < abs(prefix[e] - target):
            e += 1
        cuts.append(e)
    cuts.append(rows)
    # an empty band at row 0 cannot be expressed by gsaj_set_tile_band: give rank 0 the first row
    if world > 1 and cuts[1] == 0 and rows > 0:
        cuts = [0] + [max(c, 1) for c in cuts[1:]]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def row_work(n_contrib, H=None):
    """Per tile row: number of Gaussian-pixel interactions (sum of n_contrib [H, W] over the row's pixels) of a whole-frame
    forward -- the weight balanced_bands() wants (FrameContext.interactions() is its grand total)."""
    n = n_contrib.to(torch.int64)
    H = n.shape[0] if H is None else int(H)
    per_pixel_row = n.sum(dim=1)
    rows = tile_rows(H)
    pad = rows * TILE - per_pixel_row.numel()
    if pad:
        per_pixel_row = torch.cat([per_pixel_row, per_pixel_row.new_zeros(pad)])
    return per_pixel_row.view(rows, TILE).sum(dim=1).tolist()


def pack_pose_terms(dL_dtau_sum, loss_scalars=None, out=None):
    """-> the [12] tensor one all-reduce ships: dL/dtau | loss, L_rgb, L_depth, dL/da, dL/db | aborted (left 0 here).
    loss_scalars: out_scalars[5] of gsaj_loss_seeds = {loss, L_rgb, L_depth, dL/da, dL/db} (None: zeros).  The layout is the two
    kernels' own output layouts back to back, so a caller can also hand them views of ONE buffer and skip this copy
    (gsaj.tracking.DeviceTracker does)."""
    out = torch.zeros(REDUCED_FLOATS, dtype=torch.float32, device=dL_dtau_sum.device) if out is None else out
    out[0:6] = dL_dtau_sum
    if loss_scalars is not None:
        out[6:11] = loss_scalars[0:5]
        out[11:].zero_()
    else:
        out[6:].zero_()
    return out


def allreduce_pose_terms(packed, group=None, async_op=False):
    """Sum the packed band shares over the ranks in place.  Afterwards every rank holds the whole-frame dL/dtau =
    packed[TAU], dL/dexposure = packed[EXPOSURE_GRADS] (the two inputs of PoseTracker.step) and the loss terms packed[LOSS_TERMS]."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return work if async_op else packed
    return None if async_op else packed


def gather_band_image(image, band, group=None):
    """Assemble the whole image [C, H, W] from every rank's band (for display / keyframe bookkeeping; tracking itself never
    needs it).  Pixels outside a rank's band hold the background, so the bands are cut out and concatenated: one all-gather
    of equal-sized (padded) row blocks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return image.clone()
    world = dist.get_world_size(group)
    C, H, W = image.shape
    mine = torch.tensor([band[0], band[1]], dtype=torch.int64, device=image.device)
    all_bands = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(all_bands, mine, group=group)
    all_bands = [tuple(int(v) for v in b.tolist()) for b in all_bands]
    tallest = max(e - b for b, e in all_bands) * TILE
    block = image.new_zeros((C, max(tallest, 1), W))
    y0, y1 = band[0] * TILE, min(H, band[1] * TILE)
    if y1 > y0:
        block[:, : y1 - y0] = image[:, y0:y1]
    parts = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(parts, block, group=group)
    out = image.clone()
    for (b, e), part in zip(all_bands, parts):
        y0, y1 = b * TILE, min(H, e * TILE)
        if y1 > y0:
            out[:, y0:y1] = part[:, : y1 - y0]
    return out
