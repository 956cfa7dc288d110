"""Surfel parameters + optimiser groups: the slice of scene/gaussian_model.py the train step touches
(activations :60-134, Adam groups :181-203, densification statistics :484-486)."""
import torch
import torch.nn as nn

from .synthetic import SurfelScene


def pad_scene(scene: SurfelScene, capacity: int) -> SurfelScene:
    """Append dead slots: opacity logit DEAD_LOGIT (sigmoid == 0), identity rotation, everything else zero."""
    from .densify import DEAD_LOGIT
    n = capacity - scene.xyz.shape[0]

    def pad(t, fill=0.0):
        return torch.cat((t, torch.full((n,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)))
    rot = torch.zeros(n, 4, dtype=scene.rotation.dtype, device=scene.rotation.device)
    rot[:, 0] = 1
    return SurfelScene(pad(scene.xyz), pad(scene.log_scale, -6.0), torch.cat((scene.rotation, rot)), pad(scene.opacity_logit, DEAD_LOGIT),
                       pad(scene.f_dc), pad(scene.f_rest), pad(scene.feature, -1e-2))


class SurfelModel(nn.Module):
    def __init__(self, scene: SurfelScene, sh_degree: int = 3, active_sh_degree: int = 3, packed_sh: bool = False,
                 capacity: int = None, with_motion_mask: bool = False):
        """capacity: number of surfel SLOTS (>= the scene's surfel count); the extra slots start dead and are filled by
        densification without re-allocating anything (dgs_amd/densify.py).  `alive` marks the slots in use.
        packed_sh: keep the SH coefficients as ONE [P,16,3] parameter `_features` (what the rasterizer reads) instead of
        the reference's `_features_dc` / `_features_rest` pair that is concatenated on every render (gaussian_model.py:103-107);
        the two learning rates then become a periodic pattern of the flat Adam kernel.  `_features_dc` / `_features_rest`
        stay readable as views.
        with_motion_mask (gs_with_motion_mask, arguments/__init__.py:72; scene/gaussian_model.py:59-63,94-99,177-179): one more feature
        column, initialised to 0, whose sigmoid scales every surfel's deformation (`motion_mask`) and is rendered by render(render_motion=True)."""
        super().__init__()
        self.with_motion_mask = bool(with_motion_mask)
        if self.with_motion_mask and scene.feature.shape[1] % 2 == 0:      # (hyper coordinates only so far: append the mask column)
            scene = scene._replace(feature=torch.cat((scene.feature, torch.zeros_like(scene.feature[:, :1])), dim=1))
        n_alive = scene.xyz.shape[0]
        if capacity is not None and capacity > n_alive:
            scene = pad_scene(scene, capacity)
        self.max_sh_degree = sh_degree
        self.active_sh_degree = active_sh_degree
        self.packed_sh = packed_sh
        self._xyz = nn.Parameter(scene.xyz.clone())
        if packed_sh:
            self._features = nn.Parameter(torch.cat((scene.f_dc, scene.f_rest), dim=1).contiguous())
        else:
            self._features_dc = nn.Parameter(scene.f_dc.clone())
            self._features_rest = nn.Parameter(scene.f_rest.clone())
        self._scaling = nn.Parameter(scene.log_scale.clone())
        self._rotation = nn.Parameter(scene.rotation.clone())
        self._opacity = nn.Parameter(scene.opacity_logit.clone())
        self.feature = nn.Parameter(scene.feature.clone())
        P = scene.xyz.shape[0]
        self.register_buffer("xyz_gradient_accum", torch.zeros(P, 1), persistent=False)
        self.register_buffer("denom", torch.zeros(P, 1), persistent=False)
        self.register_buffer("max_radii2D", torch.zeros(P, dtype=torch.int32), persistent=False)
        self.register_buffer("alive", torch.arange(P) < n_alive, persistent=False)

    @property
    def num_surfels(self):
        return int(self.alive.sum())

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: torch.exp(self._scaling))
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    @property
    def get_features(self):
        self._settle_sh()
        if self.packed_sh:
            return self._features
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def _settle_sh(self):
        """A data-parallel trainer with the sharded SH update keeps the all-gather of the last update's rows in flight between steps
        (Trainer._gather_sh_start); it registers its wait here so that every reader of the coefficients through the model's accessors
        (render, save) sees complete rows.  Raw access to `_features` needs Trainer.settle_shards()."""
        hook = self.__dict__.get("_before_sh_read")
        if hook is not None:
            hook()

    def __getattr__(self, name):
        if name in ("_features_dc", "_features_rest") and self.__dict__.get("packed_sh"):
            self._settle_sh()
            f = self._parameters["_features"]
            return f[:, :1] if name == "_features_dc" else f[:, 1:]
        return super().__getattr__(name)

    def get_rotation_bias(self, rotation_bias=0.0):
        return torch.nn.functional.normalize(self._rotation + rotation_bias)

    @property
    def motion_mask(self):  # scene/gaussian_model.py:94-99; with_motion_mask=False is the reference's default (arguments/__init__.py:72)
        if self.__dict__.get("with_motion_mask"):
            return torch.sigmoid(self.feature[..., -1:])
        return torch.ones_like(self._xyz[..., :1])

    def optimizer_groups(self, position_lr=0.00016, feature_lr=0.004, opacity_lr=0.05, scaling_lr=0.002, rotation_lr=0.002,
                         spatial_lr_scale=5.0):
        if self.packed_sh:
            # one parameter, two rates: elements (i % 48) < 3 are the DC term (dgs_adam_step_pattern)
            sh = [{'params': [self._features], 'lr': feature_lr, "name": "f_all",
                   'pattern': (3 * self._features.shape[1], 3, feature_lr / 20.0)}]
        else:
            sh = [{'params': [self._features_dc], 'lr': feature_lr, "name": "f_dc"},
                  {'params': [self._features_rest], 'lr': feature_lr / 20.0, "name": "f_rest"}]
        return [
            {'params': [self._xyz], 'lr': position_lr * spatial_lr_scale, "name": "xyz"},
            *sh,
            {'params': [self._opacity], 'lr': opacity_lr, "name": "opacity"},
            {'params': [self._scaling], 'lr': scaling_lr * spatial_lr_scale, "name": "scaling"},
            {'params': [self._rotation], 'lr': rotation_lr, "name": "rotation"},
            {'params': [self.feature], 'lr': feature_lr, 'name': 'feature'},
        ]
