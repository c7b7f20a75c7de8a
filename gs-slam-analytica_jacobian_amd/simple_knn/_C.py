"""`from simple_knn._C import distCUDA2` (reference gaussian_splatting/scene/gaussian_model.py:18, used at :246-252 to
initialise Gaussian scales) backed by libgsaj_hip.so `gsaj_dist2` (csrc/knn.hip).  Mirrors spatial.cu: takes a [P,3]
float32 device tensor, returns the [P] mean squared distance to the 3 nearest neighbours."""
import torch

from gsaj import _lib


def distCUDA2(points):
    if not torch.is_tensor(points) or points.device.type != "cuda":
        raise _lib.GsajError("distCUDA2 needs a device tensor (there is no CPU path)")
    pts = points.detach().to(torch.float32).contiguous().view(-1, 3)
    lib = _lib.load()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    ws = torch.empty(lib.gsaj_dist2_workspace_bytes(P), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.gsaj_dist2(P, pts.data_ptr(), out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream(pts.device).cuda_stream),
               "gsaj_dist2")
    return out
