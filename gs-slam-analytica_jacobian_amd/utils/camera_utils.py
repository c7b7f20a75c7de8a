"""Camera with the matrix properties `render()` consumes (reference: utils/camera_utils.py:8-109).
Only what the render / pose-update path reads is kept: R, T, pose deltas, exposure a/b,
intrinsics, and the derived world_view_transform (= W2C^T), full_proj_transform (= W2C^T P^T),
camera_center (= inv(W2C^T)[3,:3], a true inverse: W2C may be a similarity transform)."""
import torch
from torch import nn

from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2, getWorld2View2


class Camera(nn.Module):
    def __init__(self, uid, color, depth, gt_T, projection_matrix, fx, fy, cx, cy, fovx, fovy, image_height,
                 image_width, T=None, device="cuda:0"):
        super().__init__()
        self.uid, self.device = uid, device
        T = gt_T if T is None else T
        self.R, self.T = T[:3, :3].to(device), T[:3, 3].to(device)
        self.R_gt, self.T_gt = gt_T[:3, :3], gt_T[:3, 3]
        self.original_image, self.depth, self.grad_mask = color, depth, None
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.FoVx, self.FoVy = fovx, fovy
        self.image_height, self.image_width = image_height, image_width
        self.cam_rot_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=device))
        self.cam_trans_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=device))
        self.exposure_a = nn.Parameter(torch.tensor([0.0], requires_grad=True, device=device))
        self.exposure_b = nn.Parameter(torch.tensor([0.0], requires_grad=True, device=device))
        self.projection_matrix = projection_matrix.to(device=device)

    @staticmethod
    def init_from_gui(uid, T, FoVx, FoVy, fx, fy, cx, cy, H, W, device="cuda:0"):
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
        return Camera(uid, None, None, T, proj, fx, fy, cx, cy, FoVx, FoVy, H, W, device=device)

    @staticmethod
    def from_synthetic(cam, uid=0, color=None, depth=None, device="cuda:0"):
        """From a gsaj.synthetic.make_camera() dict."""
        T = torch.tensor(cam["w2c"], dtype=torch.float32)
        proj = torch.tensor(cam["projmatrix_raw"], dtype=torch.float32)
        return Camera(uid, color, depth, T, proj, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["FoVx"], cam["FoVy"],
                      cam["H"], cam["W"], device=device)

    @property
    def world_view_transform(self):
        return getWorld2View2(self.R, self.T).transpose(0, 1)

    @property
    def full_proj_transform(self):
        return self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0)).squeeze(0)

    @property
    def camera_center(self):
        return self.world_view_transform.inverse()[3, :3]

    def update_RT(self, R, t):
        self.R = R.to(device=self.device)
        self.T = t.to(device=self.device)

    def clean(self):
        self.original_image = self.depth = self.grad_mask = None
        self.cam_rot_delta = self.cam_trans_delta = None
        self.exposure_a = self.exposure_b = None
