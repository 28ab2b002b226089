"""Batched, device-resident counterpart of the reference simulator ``MultiUbsCoverageEnv``
(/root/reference/envs/mubs_cov/mubs_cov.py:10-345; maps: envs/mubs_cov/maps.py) - SURVEY 8f row f3.

B independent environments advance with ONE kernel launch per step (csrc/env_sim.hip, one wavefront per environment);
the padded observations it emits are exactly what the device-side graph builder (``graph.from_padded_obs``) and the
tensor-native replay consume, so a rollout never leaves the GPU: simulator -> graph -> agent -> actions -> simulator.

    env = BatchedUbsCoverageEnv(MapParams(n_ubs=8, n_gts=50, n_rbs=5, ...), B=4096)
    obs = env.reset(pos_ubs, pos_gts, prior)           # or env.reset() with the built-in uniform placement
    obs, reward, done, info = env.step(actions)        # actions [B, n] int64 on the device
    g = env.graph()                                    # HeteroBatch of the current observations (f1)

Initial positions and the initial GT priority permutation are INPUTS (the reference draws them from Python's / NumPy's
global generators in ``Map.set_positions`` and ``reset``); ``reset()`` without arguments places UBSs and GTs uniformly.
"""
from __future__ import annotations

import ctypes
import dataclasses
import math
from typing import Dict, Optional

import numpy as np
import torch as th

from . import _lib as L
from .graph import from_padded_obs


@dataclasses.dataclass
class MapParams:
    """Attributes a reference ``Map`` splats onto the env (maps.py:7-30) + the env's class constants (mubs_cov.py:13-20)."""
    n_ubs: int
    n_gts: int
    n_rbs: int = 1
    range_pos: float = 500.0
    episode_limit: int = 20
    dt: float = 10.0
    r_cov: float = 100.0
    r_sns: float = math.inf
    r_comm: float = math.inf
    vels: tuple = (10.0,)
    n_dirs: int = 4
    reward_scale_rate: float = 1.0
    fair_service: bool = True
    avoid_collision: bool = True
    h_ubs: float = 100.0
    p_tx: float = 1e-3 * 10 ** (10 / 10)
    n0: float = 1e-3 * 10 ** (-170 / 10)
    bw: float = 180e3
    fc: float = 2.4e9
    scene: str = "dense-urban"
    safe_dist: float = 10.0
    penalty: float = 5.0

    CHAN = {"suburban": (4.88, 0.43, 0.1, 21), "urban": (9.61, 0.16, 1, 20), "dense-urban": (12.08, 0.11, 1.6, 23),
            "high-rise-urban": (27.23, 0.08, 2.3, 34)}                                          # envs/common.py:34-39

    def chan(self):
        return self.CHAN[self.scene]

    def chan_gain(self, d_level: float) -> float:
        """envs/common.py:49-59 in float64 (used once, for max_rate: mubs_cov.py:39-41)."""
        a, b, eta_los, eta_nlos = self.chan()
        p_los = 1 / (1 + a * math.exp(-b * (math.atan(self.h_ubs / (d_level + 1e-5)) - a)))
        d = math.sqrt(d_level ** 2 + self.h_ubs ** 2)
        fspl = (4 * math.pi * self.fc * d / 3e8) ** 2
        pl = p_los * fspl * 10 ** (eta_los / 20) + (1 - p_los) * fspl * 10 ** (eta_nlos / 20)
        return 1 / pl

    @property
    def max_rate(self) -> float:
        snr_max = self.p_tx * self.chan_gain(0.0) / (self.n0 * self.bw)
        return self.bw * math.log2(1 + snr_max) * 1e-6

    def avail_moves(self) -> np.ndarray:
        """mubs_cov.py:60-64: hover + |vels| x n_dirs displacement vectors."""
        amounts = self.dt * np.array(self.vels, dtype=np.float64).reshape(-1, 1)
        ang = 2 * np.pi * np.arange(self.n_dirs) / self.n_dirs
        dirs = np.stack([np.cos(ang), np.sin(ang)]).T
        return np.ascontiguousarray(np.concatenate((np.zeros((1, 2)), np.kron(amounts, dirs))))


class BatchedUbsCoverageEnv:
    def __init__(self, p: MapParams, B: int, device="cuda", max_rate: Optional[float] = None):
        self.p, self.B, self.device = p, B, th.device(device)
        n, M = p.n_ubs, p.n_gts
        self.n_agents, self.n_gts = n, M
        moves = p.avail_moves()
        self.n_actions = moves.shape[0]
        self.episode_limit = p.episode_limit
        a, b, eta_los, eta_nlos = p.chan()
        self.max_rate = p.max_rate if max_rate is None else float(max_rate)
        self._ic = (ctypes.c_int32 * 7)(n, M, p.n_rbs, self.n_actions, p.episode_limit, int(p.fair_service),
                                        int(p.avoid_collision))
        self._fc = (ctypes.c_double * 18)(p.range_pos, p.r_cov, min(p.r_sns, 1e300), min(p.r_comm, 1e300), p.dt, p.h_ubs,
                                          p.p_tx, p.n0, p.bw, p.fc, a, b, eta_los, eta_nlos, p.safe_dist, p.penalty,
                                          p.reward_scale_rate, self.max_rate)
        self.state_dim = L.lib().uavgnn_env_state_dim(n, M, int(p.fair_service))
        dev = self.device
        f32, f64, i32 = (dict(dtype=d, device=dev) for d in (th.float32, th.float64, th.int32))
        self.moves = th.as_tensor(moves, **f64).contiguous()
        self.pos_ubs, self.pos_gts = th.zeros(B, n, 2, **f64), th.zeros(B, M, 2, **f32)
        self.prior, self.avg_rate, self.t = th.zeros(B, M, **i32), th.zeros(B, M, **f32), th.zeros(B, **i32)
        self.run_f32, self.n_colls = th.zeros(B, 4, **f32), th.zeros(B, **f64)
        Sg = 5 if p.fair_service else 4
        self.out = dict(d_u2g=th.zeros(B, n, M, **f32), d_u2u=th.zeros(B, n, n, **f32), gt_ubs=th.zeros(B, M, **i32),
                        gt_rb=th.zeros(B, M, **i32), rate_per_gt=th.zeros(B, M, **f32), rate_per_ubs=th.zeros(B, n, **f64),
                        mask_collision=th.zeros(B, n, **i32), reward=th.zeros(B, n, **f64), done=th.zeros(B, **f32),
                        obs_gt=th.zeros(B, n, M, Sg, **f32), obs_ubs=th.zeros(B, n, max(n - 1, 0), 3, **f32),
                        obs_agent=th.zeros(B, n, 2, **f32), state=th.zeros(B, self.state_dim, **f32))
        self.ep_ret = th.zeros(B, **f64)

    # ---- the one kernel ---------------------------------------------------------------------------------------------
    def _launch(self, actions: Optional[th.Tensor]):
        L.require_gpu(self.pos_ubs, actions)
        o = self.out
        if actions is not None:
            actions = actions.to(th.int64).contiguous()
        L.check(L.lib().uavgnn_env_step(self._ic, self._fc, self.B, L.ptr(actions), self.moves.data_ptr(),
                                        self.pos_ubs.data_ptr(), self.pos_gts.data_ptr(), self.prior.data_ptr(),
                                        self.avg_rate.data_ptr(), self.t.data_ptr(), self.run_f32.data_ptr(),
                                        self.n_colls.data_ptr(), o["d_u2g"].data_ptr(), o["d_u2u"].data_ptr(),
                                        o["gt_ubs"].data_ptr(), o["gt_rb"].data_ptr(), o["rate_per_gt"].data_ptr(),
                                        o["rate_per_ubs"].data_ptr(), o["mask_collision"].data_ptr(),
                                        o["reward"].data_ptr(), o["done"].data_ptr(), o["obs_gt"].data_ptr(),
                                        o["obs_ubs"].data_ptr(), o["obs_agent"].data_ptr(), o["state"].data_ptr(),
                                        L.stream()), "uavgnn_env_step")

    def observations(self) -> Dict[str, th.Tensor]:
        """Padded observation tensors of the current state (mubs_cov.py:215-242) + what the wrapper's comm graph reads."""
        o = self.out
        return dict(gt=o["obs_gt"], ubs=o["obs_ubs"], agent=o["obs_agent"], d_u2u=o["d_u2u"], state=o["state"])

    def graph(self, with_comm: bool = True, static: bool = False):
        """HeteroBatch of the current observations, built on the device (f1)."""
        o = self.out
        return from_padded_obs(o["obs_gt"], o["obs_ubs"], o["obs_agent"], o["d_u2u"] if with_comm else None,
                               r_comm=self.p.r_comm, static=static)

    def reset(self, pos_ubs=None, pos_gts=None, prior=None, generator: Optional[th.Generator] = None):
        """mubs_cov.py:86-102.  pos_ubs [B,n,2], pos_gts [B,M,2], prior [B,M] (a permutation of the GTs per env); missing
        ones are drawn uniformly over the square / as random permutations on the device."""
        B, n, M, dev = self.B, self.n_agents, self.n_gts, self.device
        if pos_ubs is None:
            pos_ubs = th.rand(B, n, 2, device=dev, generator=generator, dtype=th.float64) * self.p.range_pos
        if pos_gts is None:
            pos_gts = th.rand(B, M, 2, device=dev, generator=generator) * self.p.range_pos
        if prior is None:
            prior = th.argsort(th.rand(B, M, device=dev, generator=generator), dim=1)
        self.pos_ubs.copy_(th.as_tensor(pos_ubs, dtype=th.float64))
        self.pos_gts.copy_(th.as_tensor(pos_gts).to(th.float32))
        self.prior.copy_(th.as_tensor(prior).to(th.int32))
        for t_ in (self.avg_rate, self.t, self.run_f32, self.n_colls, self.ep_ret):
            t_.zero_()
        self._launch(None)                                     # UBSs serve the GTs at the initial positions (:98)
        return self.observations()

    def step(self, actions: th.Tensor):
        """mubs_cov.py:104-129.  actions [B, n] (or [B*n]) int64 on the device.  Returns (observations, reward [B,n] f64,
        done [B] f32, info dict of device tensors) - nothing is synchronised with the host."""
        self._launch(actions.view(self.B, self.n_agents))
        o = self.out
        self.ep_ret += o["reward"].mean(1)
        info = dict(EpRet=self.ep_ret, EpLen=self.t, AvgGlobalUtility=self.run_f32[:, 1], FairIdx=self.run_f32[:, 2],
                    TotalThroughput=self.run_f32[:, 0], ProbCollision=self.n_colls / self.t.clamp(min=1),
                    BadMask=o["done"])                       # the only termination is the episode limit (:343-345)
        return self.observations(), o["reward"], o["done"], info
