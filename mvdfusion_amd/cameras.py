"""Host-side camera algebra for the MVD-Fusion hot path (no pytorch3d dependency).

Conventions are PyTorch3D's (the reference builds ``PerspectiveCameras`` without ``image_size`` => NDC,
row vectors):  X_view = X_world R + T;  ndc = (fx X/Z + px, fy Y/Z + py, 1/Z);
camera centre C = -T R^T.  The device kernels (csrc/gridattn.hip) consume the packed 20-float record made
by :func:`pack_cameras`; anything with ``.R .T .focal_length .principal_point`` attributes (e.g. a real
``pytorch3d.renderer.PerspectiveCameras``) is accepted wherever the reference passes camera objects.

Replaces: utils/camera_utils.py:14-31 (_get_camera_slice), :58-115 (_get_relative_camera),
pytorch3d look_at_view_transform as used by dataset/gso_test.py:134-149.
"""
import math
from dataclasses import dataclass

import torch


@dataclass
class Cameras:
    R: torch.Tensor                 # (N,3,3)
    T: torch.Tensor                 # (N,3)
    focal_length: torch.Tensor      # (N,2)
    principal_point: torch.Tensor   # (N,2)
    image_size = None

    def __len__(self):
        return self.R.shape[0]

    @property
    def device(self):
        return self.R.device

    def to(self, device):
        return Cameras(self.R.to(device), self.T.to(device), self.focal_length.to(device),
                       self.principal_point.to(device))

    def get_camera_center(self):
        return -torch.einsum("ni,nji->nj", self.T, self.R)

    def __getitem__(self, idx):
        return get_camera_slice(self, idx)


def _as_cameras(c):
    if isinstance(c, Cameras):
        return c
    f = torch.as_tensor(c.focal_length, dtype=torch.float32)
    p = torch.as_tensor(c.principal_point, dtype=torch.float32)
    n = c.R.shape[0]
    if f.dim() == 1:
        f = f.reshape(-1, 1).expand(-1, 2)
    return Cameras(c.R.float(), c.T.float(), f.expand(n, 2).contiguous(), p.expand(n, 2).contiguous())


def get_camera_slice(cams, indices):
    """utils/camera_utils.py:14-31."""
    cams = _as_cameras(cams)
    idx = torch.as_tensor(indices, dtype=torch.long).reshape(-1)
    return Cameras(cams.R[idx], cams.T[idx], cams.focal_length[idx], cams.principal_point[idx])


def get_relative_camera(cams, query_idx):
    """utils/camera_utils.py:58-115 with center_at_origin=False: the query (input) view's rotation
    becomes identity, translations are kept:  R'_i = R_q^T R_i,  T'_i = T_i."""
    cams = _as_cameras(cams)
    q = torch.as_tensor(query_idx, dtype=torch.long).reshape(-1)[:1]
    Rq = cams.R[q]                                       # (1,3,3)
    Rrel = torch.matmul(Rq.transpose(1, 2), cams.R)      # inverse of Rotate(R_q) composed with [[R_i,0],[T_i,1]]
    return Cameras(Rrel, cams.T.clone(), cams.focal_length.clone(), cams.principal_point.clone())


def look_at_view_transform(dist, elev_deg, azim_deg, up=(0.0, 1.0, 0.0)):
    """PyTorch3D look_at_view_transform(dist, elev, azim, degrees=True, up): returns R (N,3,3), T (N,3)."""
    elev = torch.as_tensor(elev_deg, dtype=torch.float32).reshape(-1) * (math.pi / 180.0)
    azim = torch.as_tensor(azim_deg, dtype=torch.float32).reshape(-1) * (math.pi / 180.0)
    n = max(elev.numel(), azim.numel())
    elev, azim = elev.expand(n), azim.expand(n)
    d = torch.as_tensor(dist, dtype=torch.float32).reshape(-1).expand(n)
    C = torch.stack([d * torch.cos(elev) * torch.sin(azim), d * torch.sin(elev),
                     d * torch.cos(elev) * torch.cos(azim)], dim=1)
    upv = torch.tensor(up, dtype=torch.float32)[None].expand(n, 3)
    z = torch.nn.functional.normalize(-C, eps=1e-5)
    x = torch.nn.functional.normalize(torch.cross(upv, z, dim=1), eps=1e-5)
    y = torch.nn.functional.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    R = torch.stack([x, y, z], dim=2)                    # axes as columns
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T


CAM_RECORD = 20  # floats per camera record handed to the device kernels


def pack_cameras(cams):
    """(N,20) fp32: R row-major (9), T (3), f (2), p (2), C (3), pad (1)."""
    cams = _as_cameras(cams)
    n = len(cams)
    C = cams.get_camera_center()
    rec = torch.zeros(n, CAM_RECORD, dtype=torch.float32, device=cams.R.device)
    rec[:, 0:9] = cams.R.reshape(n, 9)
    rec[:, 9:12] = cams.T
    rec[:, 12:14] = cams.focal_length
    rec[:, 14:16] = cams.principal_point
    rec[:, 16:19] = C
    return rec
