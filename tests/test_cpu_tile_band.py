"""CPU: host logic of the tile-band sharding of one frame (gsaj.tile_band_shard; SURVEY 8e partitioning B) -- how the tile
rows are dealt out, and the collectives that add the band shares / assemble the image, over gloo with world_size 2.
The kernels' side (gsaj_set_tile_band) is tests/test_gpu_tile_band.py."""
import os
import subprocess
import sys

import pytest
import torch

from gsaj import tile_band_shard as tbs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("H,world", [(480, 1), (480, 2), (480, 4), (480, 8), (680, 8), (720, 8), (40, 8), (16, 3), (1, 2)])
def test_uniform_bands_partition_the_tile_rows(H, world):
    bands = tbs.uniform_bands(H, world)
    rows = tbs.tile_rows(H)
    assert len(bands) == world and bands[0][0] == 0 and bands[-1][1] == rows
    for (b0, e0), (b1, e1) in zip(bands, bands[1:]):
        assert e0 == b1 and b0 <= e0
    assert bands[0][1] > 0  # rank 0 is never the empty band [0, 0), which gsaj_set_tile_band cannot express
    sizes = [e - b for b, e in bands]
    assert sorted(sizes, reverse=True) == sizes  # contiguous ceil-split: only trailing bands shrink


def test_balanced_bands_follow_the_weights():
    assert tbs.balanced_bands([1] * 30, 4) == [(0, 7), (7, 15), (15, 22), (22, 30)]
    # heavy rows in the middle (the usual frame: the object sits in the centre of the image)
    w = [1, 1, 2, 10, 40, 40, 10, 2, 1, 1]
    bands = tbs.balanced_bands(w, 2)
    assert bands == [(0, 5), (5, 10)]
    bands = tbs.balanced_bands(w, 4)
    share = [sum(w[b:e]) for b, e in bands]
    assert bands[0][0] == 0 and bands[-1][1] == len(w) and all(e0 == b1 for (_, e0), (b1, _) in zip(bands, bands[1:]))
    assert max(share) <= sum(w) / 4 + max(w)
    # one row outweighs everything: the other ranks get what is left, possibly nothing -- but rank 0 never [0, 0)
    bands = tbs.balanced_bands([0, 0, 0, 9], 2)
    assert bands[0][1] >= 1 and bands[-1][1] == 4
    assert tbs.balanced_bands([0.0] * 30, 4) == tbs.uniform_bands(480, 4)
    with pytest.raises(ValueError):
        tbs.balanced_bands([1, -1], 2)
    with pytest.raises(ValueError):
        tbs.uniform_bands(480, 0)


def test_row_work_sums_pixels_per_tile_row():
    n = torch.zeros((40, 8), dtype=torch.int32)  # H = 40: tile rows of 16, 16 and 8 pixel rows
    n[0:16] = 1
    n[16:32] = 2
    n[32:40] = 5
    assert tbs.row_work(n) == [16 * 8 * 1, 16 * 8 * 2, 8 * 8 * 5]


def test_pack_pose_terms_layout():
    tau = torch.arange(6, dtype=torch.float32)
    scalars = torch.tensor([10.0, 11.0, 12.0, 13.0, 14.0])  # loss, L_rgb, L_depth, dL/da, dL/db
    p = tbs.pack_pose_terms(tau, scalars)
    assert p.tolist() == [0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 0]   # (the 12th float: "a rank's share was aborted", set by the tracker)
    assert tbs.pack_pose_terms(tau).tolist() == [0, 1, 2, 3, 4, 5, 0, 0, 0, 0, 0, 0]
    assert tbs.REDUCED_FLOATS == 12 and p[tbs.ABORTED].tolist() == [0.0]
    assert tbs.allreduce_pose_terms(p) is p  # no process group: identity


_WORKER = r'''
import sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gsaj import tile_band_shard as tbs
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
H, W = 72, 20                                   # 5 tile rows, the last one 8 pixels tall
bands = tbs.balanced_bands([1, 1, 1, 1, 4], 2)  # uneven: rank 0 gets 4 rows, rank 1 the heavy last one
assert bands == [(0, 4), (4, 5)], bands
band = bands[rank]
# stand-in for a banded render: the whole frame is the row index, each rank only has its band, background elsewhere
truth = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(3, H, W).contiguous()
mine = torch.full((3, H, W), -1.0)
y0, y1 = band[0] * 16, min(H, band[1] * 16)
mine[:, y0:y1] = truth[:, y0:y1]
full = tbs.gather_band_image(mine, band)
assert torch.equal(full, truth), (rank, (full - truth).abs().max())
# band shares of dL/dtau, exposure gradients and loss terms add up on every rank
tau = torch.arange(6, dtype=torch.float32) * (rank + 1)
scal = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0]) * (10 ** rank)
packed = tbs.allreduce_pose_terms(tbs.pack_pose_terms(tau, scal))
assert packed[0:6].tolist() == [0.0, 3.0, 6.0, 9.0, 12.0, 15.0]
assert packed[tbs.EXPOSURE_GRADS].tolist() == [44.0, 55.0] and packed[tbs.LOSS_TERMS].tolist() == [11.0, 22.0, 33.0]
work = tbs.allreduce_pose_terms(tbs.pack_pose_terms(tau), async_op=True)
work.wait()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_tile_band_collectives_two_processes_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(31500 + os.getpid() % 2000)
    pkg = os.path.join(ROOT, "gs-slam-analytica_jacobian_amd")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), pkg, port, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
