"""Container-only import shims so that ``/root/reference`` (pure Python) can be imported here.

TEST INFRASTRUCTURE.  Never shipped to / used on the GPU box (the reference tree does not exist there).

The reference imports three third-party packages that are absent from its tree and from this image:

* ``pytorch3d`` (version unpinned, ENVIRONMENT.md:42; era 0.7.x) -- camera algebra only.
  Restated below from the published PyTorch3D semantics (row-vector convention):
    X_view = X_world @ R + T                     (get_world_to_view_transform = [[R,0],[T,1]])
    ndc    = (fx*X/Z + px, fy*Y/Z + py, 1/Z)     (PerspectiveCameras, NDC space, K = [[fx,0,px,0],[0,fy,py,0],[0,0,0,1],[0,0,1,0]])
    unproject(x, y, depth) = inverse of the above with Z = depth
    camera centre C = -T @ R^T
  PyTorch3D composes 4x4 matrices and divides by the homogeneous coordinate; the stand-in does the same
  (including ``torch.inverse`` of the projection matrix) so that rounding order follows the library.
  Call sites in the reference: utils/ray_utils.py:192, mvdfusion/view_attn_efficient2.py:275,293,303,321,334,350,
  utils/camera_utils.py:23-30,80-103, mvdfusion/viewfusion_zero_depth_rgb.py:228-234, dataset/gso_test.py:134-149.
* ``timm`` (unpinned, requirements.txt:24): ``Attention`` and ``Mlp`` of vision_transformer
  (mvdfusion/view_attn_efficient2.py:6,52,57) -- standard MHA / 2-layer MLP, restated below.
* ``omegaconf`` -- no arithmetic; type stubs only.

Facade-only extras (pytorch_lightning, clip, kornia) get inert stubs.
"""
import importlib.machinery
import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# pytorch3d stand-in
# --------------------------------------------------------------------------------------------
class Transform3d:
    """Row-vector 4x4 transform (points_h @ M).  Mirrors pytorch3d.transforms.Transform3d for the
    handful of methods the reference uses (inverse / compose / get_matrix / transform_points).
    Like the library it keeps composed factors as a list: ``get_matrix`` multiplies them left to
    right and ``inverse`` inverts every factor in reversed order."""

    def __init__(self, matrix=None, factors=None):
        self._factors = list(factors) if factors is not None else [("generic", matrix)]

    def get_matrix(self):
        m = self._factors[0][1]
        for _, o in self._factors[1:]:
            m = _broadcast_bmm(m, o)
        return m

    @staticmethod
    def _invert_factor(kind, m):
        if kind == "rotate":            # analytic: transpose
            return ("rotate", m.transpose(1, 2).contiguous())
        if kind == "translate":         # analytic: negate the translation row
            inv = m.clone()
            inv[:, 3, :3] = -m[:, 3, :3]
            return ("translate", inv)
        return ("generic", torch.inverse(m))

    def inverse(self):
        return Transform3d(factors=[self._invert_factor(k, m) for k, m in reversed(self._factors)])

    def compose(self, *others):
        f = list(self._factors)
        for o in others:
            f.extend(o._factors)
        return Transform3d(factors=f)

    def transform_points(self, points, eps=None):
        pts = points
        if pts.dim() == 2:
            pts = pts[None]
        ones = torch.ones(*pts.shape[:-1], 1, dtype=pts.dtype, device=pts.device)
        ph = torch.cat([pts, ones], dim=-1)
        out = _broadcast_bmm(ph, self.get_matrix())
        denom = out[..., 3:]
        if eps is not None:
            sign = denom.sign() + (denom == 0.0).type_as(denom)
            denom = sign * torch.clamp(denom.abs(), eps)
        out = out[..., :3] / denom
        if points.dim() == 2:
            out = out.reshape(points.shape)
        return out


def _broadcast_bmm(a, b):
    if a.dim() == 2:
        a = a[None]
    if len(a) != len(b):
        if len(a) == 1:
            a = a.expand(len(b), -1, -1)
        elif len(b) == 1:
            b = b.expand(len(a), -1, -1)
        else:
            raise ValueError("batch mismatch")
    return a.bmm(b)


def _rotate_translate(R, T):
    """pytorch3d get_world_to_view_transform = Rotate(R).compose(Translate(T)) = [[R,0],[T,1]]."""
    N = R.shape[0]
    eye = torch.eye(4, dtype=R.dtype, device=R.device)[None].repeat(N, 1, 1)
    r = eye.clone()
    r[:, :3, :3] = R
    t = eye.clone()
    t[:, 3, :3] = T
    return Transform3d(factors=[("rotate", r), ("translate", t)])


class CamerasBase:
    pass


class PerspectiveCameras(CamerasBase):
    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), R=None, T=None, K=None,
                 device="cpu", in_ndc=True, image_size=None):
        dtype = torch.float32
        if R is None:
            R = torch.eye(3)[None]
        if T is None:
            T = torch.zeros(1, 3)
        R = torch.as_tensor(R, dtype=dtype).to(device)
        T = torch.as_tensor(T, dtype=dtype).to(device)
        if R.dim() == 2:
            R = R[None]
        if T.dim() == 1:
            T = T[None]
        n = max(R.shape[0], T.shape[0])
        if R.shape[0] == 1 and n > 1:
            R = R.expand(n, -1, -1)
        if T.shape[0] == 1 and n > 1:
            T = T.expand(n, -1)
        fl = torch.as_tensor(focal_length, dtype=dtype).to(device)
        if fl.dim() == 0:
            fl = fl.reshape(1, 1).expand(1, 2)
        elif fl.dim() == 1:
            fl = fl.reshape(-1, 1).expand(-1, 2)
        pp = torch.as_tensor(principal_point, dtype=dtype).to(device)
        if pp.dim() == 1:
            pp = pp[None]
        n = max(n, fl.shape[0], pp.shape[0])
        self.R = (R.expand(n, -1, -1) if R.shape[0] == 1 else R).clone()
        self.T = (T.expand(n, -1) if T.shape[0] == 1 else T).clone()
        self.focal_length = (fl.expand(n, -1) if fl.shape[0] == 1 else fl).clone()
        self.principal_point = (pp.expand(n, -1) if pp.shape[0] == 1 else pp).clone()
        self.image_size = image_size
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self._in_ndc = in_ndc

    def __len__(self):
        return self.R.shape[0]

    def to(self, device):
        return PerspectiveCameras(R=self.R, T=self.T, focal_length=self.focal_length,
                                  principal_point=self.principal_point, device=device,
                                  image_size=self.image_size)

    def in_ndc(self):
        return self._in_ndc

    def get_world_to_view_transform(self, **kwargs):
        return _rotate_translate(self.R, self.T)

    def get_camera_center(self, **kwargs):
        P = self.get_world_to_view_transform().inverse().get_matrix()
        return P[:, 3, :3]

    def get_projection_transform(self, **kwargs):
        n = len(self)
        K = torch.zeros(n, 4, 4, dtype=self.R.dtype, device=self.R.device)
        K[:, 0, 0] = self.focal_length[:, 0]
        K[:, 1, 1] = self.focal_length[:, 1]
        K[:, 0, 2] = self.principal_point[:, 0]
        K[:, 1, 2] = self.principal_point[:, 1]
        K[:, 3, 2] = 1.0
        K[:, 2, 3] = 1.0
        return Transform3d(K.transpose(1, 2).contiguous())

    def get_full_projection_transform(self, **kwargs):
        return self.get_world_to_view_transform().compose(self.get_projection_transform())

    def transform_points(self, points, eps=None, **kwargs):
        return self.get_full_projection_transform().transform_points(points, eps=eps)

    def transform_points_ndc(self, points, eps=None, **kwargs):
        return self.get_full_projection_transform().transform_points(points, eps=eps)

    def unproject_points(self, xy_depth, world_coordinates=True, from_ndc=False, **kwargs):
        if world_coordinates:
            to_camera = self.get_full_projection_transform()
        else:
            to_camera = self.get_projection_transform()
        unprojection = to_camera.inverse()
        xy_inv_depth = torch.cat((xy_depth[..., :2], 1.0 / xy_depth[..., 2:3]), dim=-1)
        return unprojection.transform_points(xy_inv_depth)


class FoVOrthographicCameras(PerspectiveCameras):
    pass


class RayBundle(tuple):
    """NamedTuple-like (origins, directions, lengths, xys)."""
    __slots__ = ()
    _fields = ("origins", "directions", "lengths", "xys")

    def __new__(cls, origins, directions, lengths, xys):
        return tuple.__new__(cls, (origins, directions, lengths, xys))

    origins = property(lambda self: self[0])
    directions = property(lambda self: self[1])
    lengths = property(lambda self: self[2])
    xys = property(lambda self: self[3])


def ray_bundle_to_ray_points(ray_bundle):
    """pytorch3d.renderer.implicit.utils: o[..., None, :] + len[..., :, None] * d[..., None, :]."""
    return (ray_bundle.origins[..., None, :]
            + ray_bundle.lengths[..., :, None] * ray_bundle.directions[..., None, :])


class GridRaysampler(nn.Module):
    """Constructed by mvdfusion/embedder.py:35-45, never called on the hot path."""

    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs


def camera_position_from_spherical_angles(distance, elevation, azimuth, degrees=True):
    dist = torch.as_tensor(distance, dtype=torch.float32).reshape(-1)
    elev = torch.as_tensor(elevation, dtype=torch.float32).reshape(-1)
    azim = torch.as_tensor(azimuth, dtype=torch.float32).reshape(-1)
    n = max(len(dist), len(elev), len(azim))
    dist, elev, azim = (t.expand(n) if len(t) == 1 else t for t in (dist, elev, azim))
    if degrees:
        elev = math.pi / 180.0 * elev
        azim = math.pi / 180.0 * azim
    x = dist * torch.cos(elev) * torch.sin(azim)
    y = dist * torch.sin(elev)
    z = dist * torch.cos(elev) * torch.cos(azim)
    return torch.stack([x, y, z], dim=1)


def look_at_rotation(camera_position, at=((0, 0, 0),), up=((0, 1, 0),)):
    C = torch.as_tensor(camera_position, dtype=torch.float32)
    at = torch.as_tensor(at, dtype=torch.float32).expand_as(C)
    up = torch.as_tensor(up, dtype=torch.float32).expand_as(C)
    z_axis = F.normalize(at - C, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        replacement = F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5)
        x_axis = torch.where(is_close, replacement, x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    return R.transpose(1, 2)


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, eye=None,
                           at=((0, 0, 0),), up=((0, 1, 0),), device="cpu"):
    if eye is not None:
        C = torch.as_tensor(eye, dtype=torch.float32)
    else:
        C = camera_position_from_spherical_angles(dist, elev, azim, degrees=degrees)
    R = look_at_rotation(C, at, up)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T


def meshgrid_ij(*a):
    return torch.meshgrid(*a, indexing="ij")


# --------------------------------------------------------------------------------------------
# timm stand-in
# --------------------------------------------------------------------------------------------
class Attention(nn.Module):
    """timm.models.vision_transformer.Attention (no qk_norm, no dropout)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, **kwargs):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q = q * self.scale
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, **kw):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


# --------------------------------------------------------------------------------------------
# installation
# --------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# ---------------------------------------------------------------------------------------------
# clip / kornia (used only by external/sd1/ldm/modules/encoders/modules.py:402-441).  Published behaviour:
#   clip.load(name) -> (CLIP model, preprocess);  CLIP.encode_image(x) = self.visual(x.type(self.dtype));  VisionTransformer as in
#   clip/model.py (conv1 no bias, class_embedding, positional_embedding, ln_pre, Transformer of ResidualAttentionBlock
#   [nn.MultiheadAttention, LayerNorm, c_fc -> QuickGELU -> c_proj], ln_post, proj).  kornia.geometry.resize(bicubic) wraps
#   F.interpolate; kornia.enhance.normalize = (x - mean[:, None, None]) / std[:, None, None].
# ---------------------------------------------------------------------------------------------
def _kornia_resize(x, size, interpolation="bilinear", align_corners=None, antialias=False, **kw):
    import torch.nn.functional as F
    assert not antialias
    return F.interpolate(x, size=size, mode=interpolation, align_corners=align_corners)


def _kornia_normalize(x, mean, std):
    return (x - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1)


def _clip_load(name="ViT-L/14", device="cpu", jit=False, **kw):
    import torch
    import torch.nn as nn
    from collections import OrderedDict
    cfg = {"ViT-L/14": (224, 14, 1024, 24, 16, 768, 768), "tiny-test": (224, 14, 128, 2, 2, 64, 64)}[name]
    image, patch, width, layers, heads, out_dim, text_width = cfg

    class QuickGELU(nn.Module):
        def forward(self, x):
            return x * torch.sigmoid(1.702 * x)

    class ResidualAttentionBlock(nn.Module):
        def __init__(self):
            super().__init__()
            self.attn = nn.MultiheadAttention(width, heads)
            self.ln_1 = nn.LayerNorm(width)
            self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, width * 4)), ("gelu", QuickGELU()),
                                                  ("c_proj", nn.Linear(width * 4, width))]))
            self.ln_2 = nn.LayerNorm(width)

        def forward(self, x):                      # x: (L, N, width), sequence first as in CLIP
            y = self.ln_1(x)
            x = x + self.attn(y, y, y, need_weights=False)[0]
            return x + self.mlp(self.ln_2(x))

    class Transformer(nn.Module):
        def __init__(self):
            super().__init__()
            self.resblocks = nn.Sequential(*[ResidualAttentionBlock() for _ in range(layers)])

        def forward(self, x):
            return self.resblocks(x)

    class VisionTransformer(nn.Module):
        def __init__(self):
            super().__init__()
            scale = width ** -0.5
            self.conv1 = nn.Conv2d(3, width, kernel_size=patch, stride=patch, bias=False)
            self.class_embedding = nn.Parameter(scale * torch.randn(width))
            self.positional_embedding = nn.Parameter(scale * torch.randn((image // patch) ** 2 + 1, width))
            self.ln_pre = nn.LayerNorm(width)
            self.transformer = Transformer()
            self.ln_post = nn.LayerNorm(width)
            self.proj = nn.Parameter(scale * torch.randn(width, out_dim))

        def forward(self, x):
            x = self.conv1(x)
            x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
            x = torch.cat([self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype), x], dim=1)
            x = x + self.positional_embedding.to(x.dtype)
            x = self.ln_pre(x)
            x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
            x = self.ln_post(x[:, 0, :])
            return x @ self.proj

    class CLIP(nn.Module):
        def __init__(self):
            super().__init__()
            self.visual = VisionTransformer()
            self.transformer = nn.Identity()         # the text tower; the reference deletes it (encoders/modules.py:417)
            self.token_embedding = nn.Embedding(49408, text_width)
            self.positional_embedding = nn.Parameter(torch.empty(77, text_width))
            self.ln_final = nn.LayerNorm(text_width)
            self.text_projection = nn.Parameter(torch.empty(text_width, out_dim))
            self.logit_scale = nn.Parameter(torch.ones([]))

        @property
        def dtype(self):
            return self.visual.conv1.weight.dtype

        def encode_image(self, image):
            return self.visual(image.type(self.dtype))

    return CLIP(), None


def install(reference_root="/root/reference"):
    """Inject the stand-ins and put the reference on sys.path (idempotent)."""
    sys.dont_write_bytecode = True
    if "pytorch3d" not in sys.modules:
        _mod("pytorch3d")
        _mod("pytorch3d.renderer", PerspectiveCameras=PerspectiveCameras, RayBundle=RayBundle,
             ray_bundle_to_ray_points=ray_bundle_to_ray_points, GridRaysampler=GridRaysampler,
             look_at_view_transform=look_at_view_transform, FoVOrthographicCameras=FoVOrthographicCameras)
        _mod("pytorch3d.renderer.cameras", CamerasBase=CamerasBase, look_at_view_transform=look_at_view_transform,
             PerspectiveCameras=PerspectiveCameras)
        _mod("pytorch3d.renderer.implicit")
        _mod("pytorch3d.renderer.implicit.raysampling", _xy_to_ray_bundle=None)
        _mod("pytorch3d.common")
        _mod("pytorch3d.common.compat", meshgrid_ij=meshgrid_ij)
        _mod("pytorch3d.ops", padded_to_packed=None)
        _mod("pytorch3d.transforms", Transform3d=Transform3d)
    if "timm" not in sys.modules:
        _mod("timm")
        _mod("timm.models")
        _mod("timm.models.vision_transformer", Attention=Attention, Mlp=Mlp)
    if "omegaconf" not in sys.modules:
        class ListConfig(list):
            pass

        class OmegaConf:
            @staticmethod
            def load(path):
                import yaml
                with open(path) as f:
                    return yaml.safe_load(f)
        _mod("omegaconf", OmegaConf=OmegaConf, ListConfig=ListConfig)
        _mod("omegaconf.listconfig", ListConfig=ListConfig)
    if "clip" not in sys.modules:          # OpenAI CLIP is not vendored by the reference: structural restatement (clip/model.py)
        _mod("clip", load=_clip_load)
    if "kornia" not in sys.modules:
        k = _mod("kornia")
        k.geometry = _mod("kornia.geometry", resize=_kornia_resize)
        k.enhance = _mod("kornia.enhance", normalize=_kornia_normalize)
    if "pytorch_lightning" not in sys.modules:
        _mod("pytorch_lightning", LightningModule=nn.Module, seed_everything=lambda s: torch.manual_seed(s))
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
