"""Samplers with the class names, arguments and return types of NS/model_components/ray_samplers.py
(SpacedSampler :54-126, UniformSampler :129, UniformLinDispPiecewiseSampler :223-246, PDFSampler :249-369,
ProposalNetworkSampler :510-600), backed by libsnerf's per-ray kernels.

Random draws come from `rand_fn(shape, device)` (default torch.rand on the device); parity tests inject the
reference's draws through it.
"""
from typing import Callable, List, Optional, Tuple

import torch
from torch import nn

from . import ops
from .rays import RayBundle, RaySamples


def _default_rand(shape, device):
    return torch.rand(shape, device=device)


class Sampler(nn.Module):
    def __init__(self, num_samples: Optional[int] = None) -> None:
        super().__init__()
        self.num_samples = num_samples
        self.rand_fn: Callable = _default_rand

    def generate_ray_samples(self, *args, **kwargs) -> RaySamples:
        raise NotImplementedError

    def forward(self, *args, **kwargs) -> RaySamples:
        return self.generate_ray_samples(*args, **kwargs)


def _make_samples(ray_bundle: RayBundle, sbins, ebins, kind: str) -> RaySamples:
    nears, fars = ray_bundle.nears, ray_bundle.fars

    def spacing_to_euclidean_fn(x):
        if kind == "uniform":
            return x * fars + (1 - x) * nears
        fn = lambda v: torch.where(v < 1, v / 2, 1 - 1 / (2 * v))
        inv = lambda v: torch.where(v < 0.5, 2 * v, 1 / (2 - 2 * v))
        return inv(x * fn(fars) + (1 - x) * fn(nears))

    compact = {"origins": ray_bundle.origins, "directions": ray_bundle.directions, "times": ray_bundle.times, "ebins": ebins, "sbins": sbins,
               "nears": nears, "fars": fars, "kind": kind}
    return ray_bundle.get_ray_samples(bin_starts=ebins[..., :-1, None], bin_ends=ebins[..., 1:, None], spacing_starts=sbins[..., :-1, None],
                                      spacing_ends=sbins[..., 1:, None], spacing_to_euclidean_fn=spacing_to_euclidean_fn, _compact=compact)


class SpacedSampler(Sampler):
    KIND = "uniform"

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(num_samples=num_samples)
        self.train_stratified, self.single_jitter = train_stratified, single_jitter

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None) -> RaySamples:
        assert ray_bundle is not None and ray_bundle.nears is not None and ray_bundle.fars is not None
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        R = ray_bundle.origins.shape[0]
        t_rand = None
        if self.train_stratified and self.training:
            t_rand = self.rand_fn((R, 1) if self.single_jitter else (R, num_samples + 1), ray_bundle.origins.device)
        sb, eb = ops.spaced_bins(ray_bundle.nears, ray_bundle.fars, num_samples, t_rand, self.KIND)
        return _make_samples(ray_bundle, sb, eb, self.KIND)


class UniformSampler(SpacedSampler):
    KIND = "uniform"


class UniformLinDispPiecewiseSampler(SpacedSampler):
    KIND = "piecewise"


class PDFSampler(Sampler):
    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = True, histogram_padding: float = 0.01) -> None:
        super().__init__(num_samples=num_samples)
        # include_original (the class default, :290) merges the existing bins into the new ones (:353-354); the proposal sampler -- the
        # only user on the hot path -- passes False (:544).  Here it is a torch cat + sort on top of the kernel's bins.
        self.include_original = include_original
        self.train_stratified, self.single_jitter, self.histogram_padding = train_stratified, single_jitter, histogram_padding
        self.last_inds = None

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, ray_samples: Optional[RaySamples] = None,
                             weights: torch.Tensor = None, num_samples: Optional[int] = None, eps: float = 1e-5,
                             anneal: float = 1.0, return_inds: bool = False) -> RaySamples:
        if ray_samples is None or ray_bundle is None:
            raise ValueError("ray_samples and ray_bundle must be provided")
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        c = ray_samples._compact
        assert c is not None, "PDFSampler needs ray samples produced by this package's samplers"
        R = weights.shape[0]
        rand = None
        if self.train_stratified and self.training:
            rand = self.rand_fn((R, 1) if self.single_jitter else (R, num_samples + 1), weights.device)
        out = ops.pdf_resample(c["sbins"], ray_bundle.nears, ray_bundle.fars, num_samples, weights=weights[..., 0], rand=rand, anneal=anneal,
                               kind=c["kind"], histogram_padding=self.histogram_padding, eps=eps, return_inds=return_inds)
        if return_inds:
            self.last_inds = out[2]
        sb, eb = out[0], out[1]
        if self.include_original:
            sb, _ = torch.sort(torch.cat([c["sbins"], sb], -1), -1)  # ray_samplers.py:353-354
            nears, fars = ray_bundle.nears, ray_bundle.fars
            if c["kind"] == "uniform":
                eb = sb * fars + (1 - sb) * nears
            else:
                fn = lambda v: torch.where(v < 1, v / 2, 1 - 1 / (2 * v))
                inv = lambda v: torch.where(v < 0.5, 2 * v, 1 / (2 - 2 * v))
                eb = inv(sb * fn(fars) + (1 - sb) * fn(nears))
            sb, eb = sb.contiguous(), eb.contiguous()
        return _make_samples(ray_bundle, sb, eb, c["kind"])


class ProposalNetworkSampler(Sampler):
    def __init__(self, num_proposal_samples_per_ray: Tuple[int] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, single_jitter: bool = False, update_sched: Callable = lambda x: 1,
                 initial_sampler: Optional[Sampler] = None) -> None:
        super().__init__()
        self.num_proposal_samples_per_ray, self.num_nerf_samples_per_ray = num_proposal_samples_per_ray, num_nerf_samples_per_ray
        self.num_proposal_network_iterations, self.update_sched = num_proposal_network_iterations, update_sched
        if num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        self.initial_sampler = UniformLinDispPiecewiseSampler(single_jitter=single_jitter) if initial_sampler is None else initial_sampler
        self.pdf_sampler = PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal, self._steps_since_update, self._step = 1.0, 0, 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, density_fns: Optional[List[Callable]] = None):
        assert ray_bundle is not None and density_fns is not None
        weights_list, ray_samples_list = [], []
        n = self.num_proposal_network_iterations
        weights, ray_samples = None, None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        for i_level in range(n + 1):
            is_prop = i_level < n
            num_samples = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num_samples)
            else:
                # weights ** anneal is applied inside the resampling kernel (ray_samplers.py:584)
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, weights, num_samples=num_samples, anneal=self._anneal)
            if is_prop:
                fn = density_fns[i_level]
                if updated:
                    density = _call_density(fn, ray_samples)
                else:
                    with torch.no_grad():
                        density = _call_density(fn, ray_samples)
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        assert ray_samples is not None
        return ray_samples, weights_list, ray_samples_list


def _call_density(fn, ray_samples: RaySamples):
    """density_fns[i](positions) as the reference calls it (ray_samplers.py:589); when the callable is one of this
    package's density fields the sample coordinates are derived in-kernel instead of materialising positions."""
    base = getattr(fn, "func", fn)
    owner = getattr(base, "__self__", None)
    if owner is not None and hasattr(owner, "density_from_ray_samples") and getattr(owner, "spatial_distortion", None) is None:
        return owner.density_from_ray_samples(ray_samples)
    return fn(ray_samples.frustums.get_positions())
