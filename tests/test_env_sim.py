"""Row f3: the batched device simulator (csrc/env_sim.hip, uav_bs_ctrl_amd/sim.py) against the REFERENCE simulator's own
output (tests/golden/env_mubs_cov.npz: envs/mubs_cov/mubs_cov.py imported unchanged and stepped with seeded actions,
tests/golden/make_golden.py `env`).

Index / decision work is BIT-EXACT: the greedy schedule (serving UBS and resource block per GT), collision masks,
visibility flags, termination.  Floating point (distances, rates, averages, Jain index, utilities, rewards, observation
and state features) to 1e-5 relative.

Priorities.  ``np.argsort(avg_rate)`` (mubs_cov.py:209) is not a stable sort and its tie order depends on NumPy's SIMD
dispatch; most GTs tie at rate 0.  The kernel uses the STABLE order.  Therefore every transition is checked twice:
  * replayed with the reference's own priority vector as input (``prior_used``): pins the whole step;
  * the kernel's next priorities must be a valid argsort of the averages (non-decreasing keys, a permutation, stable
    among equal keys) and must equal the reference's wherever the reference's keys are all distinct;
and a free-running episode (the kernel's own priorities fed forward) is checked for the permutation-invariant
quantities that do not depend on tie order whenever no RB shortage occurred."""
import numpy as np
import pytest
import torch as th

from tests.util import GOLDEN

pytestmark = pytest.mark.gpu
CASES = ["debug", "r800", "8ubs", "8ubs_parked", "8ubs_crowded"]


def _case(case):
    from uav_bs_ctrl_amd.sim import MapParams
    z = np.load(f"{GOLDEN}/env_mubs_cov.npz")
    c = {k.split(":")[-1]: float(z[k]) for k in z.files if k.startswith(f"{case}:const:")}
    moves = z[f"{case}:avail_moves"]
    n_dirs = 4
    dt = c["dt"]
    vels = tuple(sorted({round(float(np.hypot(*mv)) / dt, 9) for mv in moves[1:]}))
    p = MapParams(n_ubs=int(c["n_ubs"]), n_gts=int(c["n_gts"]), n_rbs=int(c["n_rbs"]), range_pos=c["range_pos"],
                  episode_limit=int(c["episode_limit"]), dt=dt, r_cov=c["r_cov"], r_sns=c["r_sns"], r_comm=c["r_comm"],
                  vels=vels, n_dirs=n_dirs, reward_scale_rate=c["reward_scale_rate"])
    assert np.allclose(p.avail_moves(), moves, atol=1e-9)
    assert abs(p.max_rate - c["max_rate"]) < 1e-12 * c["max_rate"]
    steps = int(z[f"{case}:steps"])
    return z, p, steps


def _close(got, ref, what, rel=1e-5):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    tol = rel * max(float(np.abs(ref).max()) if ref.size else 0.0, 1e-30) + rel * np.abs(ref)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{ref.size} off; worst {np.abs(got - ref).max():.3e} (max|ref| {np.abs(ref).max():.3e})"


@pytest.mark.parametrize("case", CASES)
def test_every_transition_replayed_with_the_reference_priorities(case):
    from uav_bs_ctrl_amd.sim import BatchedUbsCoverageEnv
    z, p, steps = _case(case)
    f = lambda t, k: z[f"{case}:t{t}:{k}"]  # noqa: E731
    env = BatchedUbsCoverageEnv(p, 1)
    env.reset(pos_ubs=f(0, "pos_ubs")[None], pos_gts=z[f"{case}:pos_gts"][None], prior=f(0, "prior_used")[None])
    n_dec = 0
    for t in range(steps + 1):
        if t > 0:
            env.prior.copy_(th.as_tensor(f(t, "prior_used")[None]).to(th.int32))     # the reference's tie resolution
            obs, rew, done, info = env.step(th.as_tensor(f(t, "actions")[None]).cuda())
        o = {k: v[0].cpu().numpy() for k, v in env.out.items()}
        # ---- decisions: bit-exact --------------------------------------------------------------------------------
        assert np.array_equal(o["gt_ubs"], f(t, "gt_ubs")), (case, t, "serving UBS per GT")
        assert np.array_equal(o["gt_rb"], f(t, "gt_rb")), (case, t, "resource block per GT")
        assert np.array_equal(o["mask_collision"].astype(bool), f(t, "mask_collision")), (case, t)
        assert np.array_equal(o["obs_gt"][..., 0], f(t, "obs_gt")[..., 0]) and np.array_equal(o["obs_ubs"][..., 0], f(t, "obs_ubs")[..., 0])
        assert float(o["done"]) == float(f(t, "done")) and int(env.t[0]) == t
        n_dec += int((f(t, "gt_ubs") >= 0).sum())
        # ---- floating point: 1e-5 --------------------------------------------------------------------------------
        _close(env.pos_ubs[0].cpu(), f(t, "pos_ubs"), "pos_ubs", 1e-12)
        for k in ("d_u2g", "d_u2u", "rate_per_gt", "rate_per_ubs", "obs_gt", "obs_ubs", "obs_agent", "state"):
            _close(o[k], f(t, k), f"{case} t={t} {k}")
        _close(env.avg_rate[0].cpu(), f(t, "avg_rate"), "avg_rate")
        rf = env.run_f32[0].cpu().numpy()
        _close(rf, [f(t, "total_throughput"), f(t, "avg_global_util"), f(t, "fair_idx"), f(t, "global_util")], "running scalars")
        _close(env.n_colls[0].cpu(), f(t, "n_colls"), "n_colls", 1e-12)
        if t > 0:
            _close(o["reward"], f(t, "reward"), "reward")
        # ---- next priorities: a valid (stable) argsort; equal to the reference's when its keys are distinct ------
        pr, avg = env.prior[0].cpu().numpy(), env.avg_rate[0].cpu().numpy()
        assert sorted(pr.tolist()) == list(range(p.n_gts))
        keys = avg[pr]
        assert (np.diff(keys) >= 0).all()
        same = np.diff(keys) == 0
        assert (np.diff(pr)[same] > 0).all(), "ties must keep GT index order (stable)"
        ref_avg = f(t, "avg_rate")
        if len(np.unique(ref_avg)) == p.n_gts:
            assert np.array_equal(pr, f(t, "prior_next"))
        else:                                   # the reference's order is also a valid argsort of (its) averages
            assert (np.diff(ref_avg[f(t, "prior_next")]) >= 0).all()
    assert case in ("8ubs",) or n_dec > 0


def test_batched_environments_are_independent_and_feed_the_graph_builder():
    """B copies with different inputs in one launch == the same environments stepped one by one; the emitted padded
    observations go straight into the device graph builder (f3 -> f1) and the agent."""
    from uav_bs_ctrl_amd.sim import BatchedUbsCoverageEnv
    z, p, steps = _case("8ubs_crowded")
    f = lambda t, k: z[f"8ubs_crowded:t{t}:{k}"]  # noqa: E731
    B = 5
    gen = th.Generator(device="cuda").manual_seed(0)
    shift = th.rand(B, 1, 2, device="cuda", generator=gen, dtype=th.float64) * 40
    shift[0] = 0
    pos_u = th.as_tensor(f(0, "pos_ubs")).cuda()[None] + shift
    pos_g = th.as_tensor(z["8ubs_crowded:pos_gts"]).cuda()[None].expand(B, -1, -1)
    prior = th.as_tensor(f(0, "prior_used")).cuda()[None].expand(B, -1)
    env = BatchedUbsCoverageEnv(p, B)
    env.reset(pos_u, pos_g, prior)
    singles = []
    for b in range(B):
        e1 = BatchedUbsCoverageEnv(p, 1)
        e1.reset(pos_u[b:b + 1], pos_g[b:b + 1], prior[b:b + 1])
        singles.append(e1)
    for t in range(1, 4):
        a = th.randint(0, env.n_actions, (B, p.n_ubs), device="cuda", generator=gen)
        env.step(a)
        for b, e1 in enumerate(singles):
            e1.step(a[b:b + 1])
            for k in ("gt_ubs", "gt_rb", "rate_per_gt", "reward", "obs_gt", "state"):
                assert th.equal(env.out[k][b], e1.out[k][0]), (t, b, k)
            assert th.equal(env.prior[b], e1.prior[0])
    # environment 0 follows the fixture while its priorities do (free-running: the kernel's own stable priorities)
    g = env.graph()
    assert g.num_nodes("agent") == B * p.n_ubs and g.graph_off.cpu().tolist() == list(range(0, B * p.n_ubs + 1, p.n_ubs))
    xs, off = g.relation_segments("seen")
    vis = env.out["obs_gt"][..., 0].sum(-1).reshape(-1).to(th.int32)
    assert th.equal(off[1:] - off[:-1], vis)
