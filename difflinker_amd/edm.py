"""``EDM`` — the ancestral sampler of DiffLinker, MI355X-native.

Drop-in for the SAMPLING surface of reference ``src/edm.py::EDM`` (constructor edm.py:15-39,
``sample_chain`` :126-176 and the helpers it uses).  With a fully-connected ``Dynamics`` the whole
chain — T reverse steps (:178-208) + final decode (:210-242) — is ONE kernel launch
(``dl_sample_chain_fc``): every molecule keeps its state in LDS of one compute unit for all T+1
denoiser calls.  The noise is drawn up front with the reference's ``torch.randn`` call sequence
(x-part then h-part, 2(T+2) calls on the tensors' device), so the same ``torch.manual_seed`` gives
the same noise stream as the reference run on the same device.

``InpaintingEDM`` (edm.py:466-727, sampling side) is here too: one HIP denoiser call + one fused HIP tail per step.

Parity note: the per-step scalars (gamma lookup, expm1 / softplus / logsigmoid algebra) are evaluated on CPU fp32 tensors
of the reference's ``[B,1]`` shape (``step_coefficients``).  The reference evaluates them on the tensors' device; the
cancellation in sigma^2_{t|s} amplifies a 1-ulp difference of a device's transcendental to ~2e-5, so "same seed, same
samples" is claimed — and tested (golden fixtures) — against the reference run on the CPU.

Out of scope (training only): ``EDM.forward`` and the likelihood/KL terms (edm.py:41-124, 244-326), the learned
``GammaNetwork`` schedule.
"""
import ctypes
import os
import threading
import warnings

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, utils
from .egnn import Dynamics, DynamicsWithPockets
from .noise import PredefinedNoiseSchedule


# measured +2.8 .. 3.2 % on the C2 headline, same box (profiles/r05/ab_split_chain.log); DIFFLINKER_SPLIT_CHAIN=0 turns it off
SPLIT_CHAIN_DEFAULT = os.environ.get('DIFFLINKER_SPLIT_CHAIN', '1') == '1'
SPLIT_STAGES_DEFAULT = int(os.environ.get('DIFFLINKER_SPLIT_STAGES', '2'))
XCD_ORDER_MODE = int(os.environ.get('DIFFLINKER_XCD_ORDER', '2'))
XCD_AWARE_ORDER = XCD_ORDER_MODE != 0


def compute_units(dev):
    """compute units of the device (256 on an MI355X): what one-molecule-per-compute-unit launches fill"""
    return torch.cuda.get_device_properties(dev).multi_processor_count


def _pass_steps(receivers, senders):
    """pair-loop steps of one pass (egnn_fc.hip: slot_plan): `receivers` atoms share 256 slots, a slot walks ceil(senders / g) senders"""
    if receivers <= 0:
        return 0
    g = max(1, min(256 // receivers, senders))
    return -(-senders // g)


def forward_cost(n, n_linker, n_layers, sublayers, team=1):
    """Relative cost of one denoiser call for a molecule of `n` atoms (`n_linker` of them receivers of the coordinate pass) on
    `team` compute units, in units of one pair-loop step: steps of the GCL and coordinate passes plus a per-pass overhead fitted
    to the measured forward times (profiles/r04/forward_time_vs_n.log: 0.68 / 0.86 / 1.02 / 1.14 ms at n = 35 / 40 / 43 / 50 on one
    compute unit, 0.68 ms at n = 50 on a team of two).  Only the RATIOS matter: they place the hand-over step of split_plan."""
    own, own_l = -(-n // team), -(-max(n_linker, 1) // team)
    a_gcl, a_eq = (1.6, 1.1) if team == 1 else (1.7, 1.3)
    return n_layers * (sublayers * (_pass_steps(own, n) + a_gcl) + _pass_steps(own_l, n) + a_eq)


_PLAN_CACHE = {}
_CACHE_LOCK = threading.Lock()           # the plan cache, the side streams and their workspaces are shared by every EDM of the process
_SPLIT_QSCALE = os.environ.get('DIFFLINKER_SPLIT_QSCALE', '1.0')          # (calibration experiments: scripts/r5/ab_split_qscale.sh)


def split_plan(sizes, linkers, n_calls, compute_units, n_layers, sublayers, min_gain=0.02, grid=48, allow_singles=None):
    """The static hand-over inside a ragged batch that holds one compute unit per molecule.  FIRST launch: every molecule on its
    compute unit for about the same time tau - the small molecules complete their chain, every other molecule b stops after
    s_b = floor(tau / cost_b) denoiser calls and leaves its state in HBM.  SECOND phase: the unfinished molecules finish on TEAMS
    OF TWO, which take the compute units the small ones left (2 x teams <= compute units).  tau runs over a grid between the
    cheapest and the dearest molecule's chain; the plan with the smallest predicted makespan wins, provided it beats the single
    launch by `min_gain`.  A function of the sizes alone (cached by them).
    ``allow_singles`` (measured, not adopted: profiles/r05/ab_split_chain.log): when the compute units do not suffice for teams
    everywhere, the molecules with the least work left resume on ONE compute unit each in a launch beside the teams' - a better
    makespan on paper (0.90 instead of 0.94 of the single launch at C2), a worse one on the chip, whose power cap taxes a first
    launch that keeps every compute unit busy to its end.
    Returns (q_end of the first launch [B], molecules for teams, molecules for single compute units) or None."""
    B = len(sizes)
    if B > compute_units or B == 0:
        return None
    allow_singles = bool(allow_singles)
    key = (tuple(sizes), tuple(linkers), n_calls, compute_units, n_layers, sublayers, min_gain, grid, bool(allow_singles), _SPLIT_QSCALE)
    with _CACHE_LOCK:
        if key in _PLAN_CACHE:
            return _PLAN_CACHE[key]
    table = {}
    for n, l in set(zip(sizes, linkers)):
        table[(n, l)] = (forward_cost(n, l, n_layers, sublayers, 1), forward_cost(n, l, n_layers, sublayers, 2))
    c1 = np.array([table[k][0] for k in zip(sizes, linkers)])
    c2 = np.array([table[k][1] for k in zip(sizes, linkers)])
    single = float(c1.max()) * n_calls
    lo = float(c1.min()) * n_calls
    best = None
    for i in range(1, grid):
        tau = lo + (single - lo) * i / grid
        q_end = np.clip((tau / c1).astype(np.int64), 1, n_calls)
        rest = np.nonzero(q_end < n_calls)[0]
        if rest.size == 0:
            continue
        left1 = (n_calls - q_end[rest]) * c1[rest]
        rest = rest[np.lexsort((rest, -left1))]                     # most work left first; ties by index: deterministic
        m = min(rest.size, compute_units - rest.size)               # teams of two: 2 m + (unfinished - m) <= compute units
        if m < 0 or (m < rest.size and not allow_singles):
            continue
        teams, singles = rest[:m], rest[m:]
        second = max(float(((n_calls - q_end[teams]) * c2[teams]).max()) if m else 0.0,
                     float(((n_calls - q_end[singles]) * c1[singles]).max()) if singles.size else 0.0)
        total = float((q_end * c1).max()) + second
        if best is None or total < best[0]:
            best = (total, q_end.tolist(), teams.tolist(), singles.tolist())
    plan = None if best is None or best[0] > (1.0 - min_gain) * single else (best[1], best[2], best[3])
    qscale = float(_SPLIT_QSCALE)
    if plan is not None and qscale != 1.0:
        stop = set(plan[1]) | set(plan[2])
        plan = ([max(1, min(n_calls - 1, int(round(q * qscale)))) if b in stop else q for b, q in enumerate(plan[0])], plan[1], plan[2])
    with _CACHE_LOCK:
        if len(_PLAN_CACHE) > 64:
            _PLAN_CACHE.clear()
        _PLAN_CACHE[key] = plan
    return plan


def split_plan_stages(sizes, linkers, n_calls, compute_units, n_layers, sublayers, max_stages=6, min_gain=0.02):
    """The hand-over of split_plan in MORE than two stages (round 6).  split_plan waits until the
    43..45-atom molecules are done before any compute unit changes hands: the compute units of the 35..40-atom molecules idle for
    up to a third of the first launch.  Here a stage ends when a number of molecules have completed their chain; the compute
    units they leave go, at once, to the unfinished molecules with the most work left, which run as TEAMS OF TWO from then on.
    Every stage is one launch of the molecules still on one compute unit (the current stream) beside one launch of the teams
    (the side stream); a molecule's state crosses a stage boundary through dl_chain_args.z_state.  Within a stage every workgroup
    is busy for about the stage's length: molecule b runs floor(length / cost_b) denoiser calls.
    The number of completions that ends a stage is searched over a small grid; the plan with the smallest predicted makespan and
    at most `max_stages` stages wins, provided it beats the single launch by `min_gain` (cost model: forward_cost).
    A function of the sizes alone (cached).  Returns a list of stages ``(q_end [B] - calls completed when the stage ends,
    teams - molecules it runs on teams of two, singles - on one compute unit)``, or None."""
    B = len(sizes)
    if B > compute_units or B == 0:
        return None
    key = ('stages', tuple(sizes), tuple(linkers), n_calls, compute_units, n_layers, sublayers, max_stages, min_gain)
    with _CACHE_LOCK:
        if key in _PLAN_CACHE:
            return _PLAN_CACHE[key]
    table = {}
    for n, l in set(zip(sizes, linkers)):
        table[(n, l)] = (forward_cost(n, l, n_layers, sublayers, 1), forward_cost(n, l, n_layers, sublayers, 2))
    c1 = np.array([table[k][0] for k in zip(sizes, linkers)])
    c2 = np.array([table[k][1] for k in zip(sizes, linkers)])
    single = float(c1.max()) * n_calls

    def simulate(events):
        left = np.full(B, n_calls, dtype=np.int64)
        mode = np.ones(B, dtype=np.int64)
        done = np.zeros(B, dtype=np.int64)
        total, stages = 0.0, []
        while (left > 0).any():
            act = np.nonzero(left > 0)[0]
            free = compute_units - int(mode[act].sum())
            if stages:                                                  # (the first stage: everybody on one compute unit)
                cand = [b for b in act if mode[b] == 1]
                cand.sort(key=lambda b: (-left[b] * c1[b], b))          # most work left first; ties by index: deterministic
                # teams of two sit in groups of eight molecules (dl_team_max): at most compute_units / 2 of them, rounded down to 8
                room = (compute_units // 2) // 8 * 8 - int((mode[act] == 2).sum())
                for b in cand[:max(0, min(free, room))]:
                    mode[b] = 2
            cost = np.where(mode == 2, c2, c1)
            fin = np.sort(left[act] * cost[act])
            tau = float(fin[min(len(fin) - 1, events - 1)])
            calls = np.minimum(left[act], np.maximum(1, np.floor(tau / cost[act]).astype(np.int64)))
            total += float((calls * cost[act]).max())
            left[act] -= calls
            done[act] += calls
            stages.append((done.tolist(), [int(b) for b in act if mode[b] == 2], [int(b) for b in act if mode[b] == 1]))
            if len(stages) > max_stages:
                return None
        return total, stages

    best = None
    for events in (12, 16, 24, 32, 48, 64, 96, 128):
        got = simulate(events)
        if got is not None and len(got[1]) >= 2 and (best is None or got[0] < best[0]):
            best = got
    plan = None if best is None or best[0] > (1.0 - min_gain) * single else best[1]
    with _CACHE_LOCK:
        if len(_PLAN_CACHE) > 64:
            _PLAN_CACHE.clear()
        _PLAN_CACHE[key] = plan
    return plan


# second stream (and its workspace) of the launch that samples the molecules beyond one per compute unit on teams
# (EDM._sample_chain_fused); per device, shared by every EDM of the process - launches on one stream are ordered
_SIDE_STREAMS = {}
_SIDE_WORKSPACE = {}


class EDM(torch.nn.Module):
    def __init__(
            self,
            dynamics,
            in_node_nf: int,
            n_dims: int,
            timesteps: int = 1000,
            noise_schedule='learned',
            noise_precision=1e-4,
            loss_type='vlb',
            norm_values=(1., 1., 1.),
            norm_biases=(None, 0., 0.),
    ):
        super().__init__()
        if noise_schedule == 'learned':
            raise NotImplementedError("noise_schedule='learned' (GammaNetwork) is training-only and out of scope; "
                                      'released configs use polynomial_2')
        self.gamma = PredefinedNoiseSchedule(noise_schedule, timesteps=timesteps, precision=noise_precision)
        self.dynamics = dynamics
        self.in_node_nf = in_node_nf
        self.n_dims = n_dims
        self.T = timesteps                      # callers overwrite it for --n_steps (generate.py:103-104)
        self.norm_values = norm_values
        self.norm_biases = norm_biases
        # 'torch' (default): the reference's own stream — torch.randn on the device in the reference's call order, so the
        # same torch.manual_seed reproduces the reference's draws.  'philox': counter-based draws generated inside the
        # kernels (no 2(T+2) randn launches, no noise bank in HBM), keyed by (noise_seed, global molecule index, atom,
        # draw) and therefore independent of the batch split; noise_seed advances by one per sampled chain.
        self.noise_source = 'torch'
        self.noise_seed = 0
        # batch size the per-step scalars are evaluated for (None: the batch at hand).  The reference evaluates them on
        # [B,1] tensors and PyTorch's CPU kernels round differently on their vector (B >= 16) and scalar paths, so a
        # shard of a batch pins this to the size of the whole batch to sample exactly what the unsharded call would.
        self.coef_batch = None
        # batch size the team size (compute units per molecule, Dynamics.team = 'auto') is chosen for (None: the batch at
        # hand).  A shard of a batch pins it to the whole batch as well: the order in which an atom's messages are summed
        # depends on the team size, so bitwise-identical samples for any split need one team size for all of them.
        self.team_batch = None
        # a batch of up to 1.25x the number of compute units: the molecules beyond one per compute unit are sampled by teams in a
        # second, concurrent launch (see _sample_chain_fused) instead of waiting for a second round.  Their messages are then
        # summed in the team's order: False keeps every molecule on one compute unit (bitwise the numbers of any other split).
        self.overflow_teams = True
        # a ragged batch that fills the chip (one molecule per compute unit): the chain runs in TWO launches - every molecule on its
        # compute unit until the small ones are done, then the big ones finish on teams of two that take the freed compute units
        # (see _sample_chain_fused / split_plan).  The steps a molecule runs on a team are summed in the team's order (fp32
        # rounding); the plan is a function of the batch's sizes alone, so a batch is sampled bit for bit the same every time.
        # False: one launch, every molecule on one compute unit for the whole chain.
        self.split_chain = SPLIT_CHAIN_DEFAULT
        # how many launches deep the hand-over may go (split_plan_stages; 2 = the two launches of round 5: split_plan)
        self.split_stages = SPLIT_STAGES_DEFAULT
        self.split_singles = os.environ.get('DIFFLINKER_SPLIT_SINGLES', '0') == '1'     # (measured, not adopted: see split_plan)

    @staticmethod
    def _side_stream(dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        with _CACHE_LOCK:
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
            return _SIDE_STREAMS[key]

    @staticmethod
    def _side_workspace(key, need, dev):
        """scratch of a launch beside the main one, one buffer per (device, role): grown, never shrunk, reused by every chain"""
        with _CACHE_LOCK:
            ws = _SIDE_WORKSPACE.get(key)
            if ws is None or ws.numel() < need:
                ws = _SIDE_WORKSPACE[key] = torch.empty(need, dtype=torch.uint8, device=dev)
            return ws

    def forward(self, *args, **kwargs):
        raise NotImplementedError('EDM.forward is the training loss (edm.py:41-124): out of scope of the '
                                  'sampling hot path')

    # ---- gamma algebra (host scalars; edm.py:369-403) -----------------------------------------------
    def sigma(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(gamma)), target_tensor)

    def alpha(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(-gamma)), target_tensor)

    def SNR(self, gamma):
        return torch.exp(-gamma)

    def sigma_and_alpha_t_given_s(self, gamma_t, gamma_s, target_tensor):
        sigma2_t_given_s = self.inflate_batch_array(
            -torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t)), target_tensor)
        log_alpha2_t_given_s = F.logsigmoid(-gamma_t) - F.logsigmoid(-gamma_s)
        alpha_t_given_s = self.inflate_batch_array(torch.exp(0.5 * log_alpha2_t_given_s), target_tensor)
        return sigma2_t_given_s, torch.sqrt(sigma2_t_given_s), alpha_t_given_s

    @staticmethod
    def inflate_batch_array(array, target):
        return array.view((array.size(0),) + (1,) * (len(target.size()) - 1))

    @staticmethod
    def numbers_of_nodes(mask):
        return torch.sum(mask.squeeze(2), dim=1)

    # ---- (un)normalisation (edm.py:347-361) ---------------------------------------------------------
    def normalize(self, x, h):
        return x / self.norm_values[0], (h.float() - self.norm_biases[1]) / self.norm_values[1]

    def unnormalize(self, x, h):
        return x * self.norm_values[0], h * self.norm_values[1] + self.norm_biases[1]

    def unnormalize_z(self, z):
        assert z.size(2) == self.n_dims + self.in_node_nf
        x, h = self.unnormalize(z[:, :, :self.n_dims], z[:, :, self.n_dims:])
        return torch.cat([x, h], dim=2)

    # ---- noise (edm.py:328-345) ---------------------------------------------------------------------
    def sample_combined_position_feature_noise(self, n_samples, n_nodes, mask):
        z_x = utils.sample_gaussian_with_mask(size=(n_samples, n_nodes, self.n_dims), device=mask.device,
                                              node_mask=mask)
        z_h = utils.sample_gaussian_with_mask(size=(n_samples, n_nodes, self.in_node_nf), device=mask.device,
                                              node_mask=mask)
        return torch.cat([z_x, z_h], dim=2)

    def sample_normal(self, mu, sigma, node_mask):
        return mu + sigma * self.sample_combined_position_feature_noise(mu.size(0), mu.size(1), node_mask)

    def draw_noise_bank(self, n_samples, n_nodes, device, n_steps=None):
        """All 2(T+2) standard-normal draws of one chain, in the reference's call order and shapes
        (x-part ``[B,N,3]`` then h-part ``[B,N,nf]``; initial z, T steps, decode), unmasked — the kernels
        apply ``linker_mask``.  Returns (noise_x [T+2,B,N,3], noise_h [T+2,B,N,nf])."""
        n_steps = self.T if n_steps is None else n_steps
        noise_x = torch.empty((n_steps + 2, n_samples, n_nodes, self.n_dims), device=device)
        noise_h = torch.empty((n_steps + 2, n_samples, n_nodes, self.in_node_nf), device=device)
        for k in range(n_steps + 2):
            # randn(size, out=slice) consumes the generator exactly like randn(size) does (same element count, same
            # launch) but writes in place: no temporary, no copy kernel (tests/test_gpu_philox.py checks the equality)
            torch.randn((n_samples, n_nodes, self.n_dims), out=noise_x[k])
            torch.randn((n_samples, n_nodes, self.in_node_nf), out=noise_h[k])
        return noise_x, noise_h

    # ---- per-step scalars ---------------------------------------------------------------------------
    def step_coefficients(self, batch_size=1):
        """Host (CPU fp32) restatement of the scalar algebra of every reverse step, in execution order
        s = T-1 ... 0 (edm.py:147-150,180-185,199,202) and of the final decode (:213-216,237-242).

        Evaluated per step on ``[batch_size, 1]`` tensors exactly as the reference does: PyTorch's CPU
        transcendentals round differently on its vectorised and scalar paths, and the cancellation in
        sigma^2_{t|s} amplifies that to ~2e-5, so the shape is part of the parity contract.
        Returns (coefs [T,4] = (t, alpha_ts, c_eps, sigma), (inv_alpha0, sigma0, sigma_x))."""
        if self.coef_batch is not None:
            batch_size = self.coef_batch
        key = (self.T, int(batch_size))
        cached = getattr(self, '_coef_cache', None)
        if cached is not None and cached[0] == key and cached[1] == self.gamma.gamma._version:
            return cached[2]
        gamma = self.gamma.gamma.detach().to('cpu', torch.float32)
        timesteps = self.gamma.timesteps
        lookup = lambda tt: gamma[torch.round(tt * timesteps).long()]
        rows = []
        for s_int in reversed(range(0, self.T)):
            s = torch.full((batch_size, 1), fill_value=s_int)
            t = (s + 1) / self.T
            s = s / self.T
            gamma_s, gamma_t = lookup(s), lookup(t)
            sigma2_ts, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(gamma_t, gamma_s, gamma_t)
            sigma_s = torch.sqrt(torch.sigmoid(gamma_s))
            sigma_t = torch.sqrt(torch.sigmoid(gamma_t))
            c_eps = sigma2_ts / alpha_ts / sigma_t
            sigma = sigma_ts * sigma_s / sigma_t
            rows.append(torch.stack([t[0, 0], alpha_ts[0, 0], c_eps[0, 0], sigma[0, 0]]))
        coefs = torch.stack(rows).to(torch.float32).contiguous()
        gamma_0 = lookup(torch.zeros(batch_size, 1))
        sigma_x = self.SNR(-0.5 * gamma_0)[0, 0]
        inv_alpha0 = (1. / torch.sqrt(torch.sigmoid(-gamma_0)))[0, 0]
        sigma0 = torch.sqrt(torch.sigmoid(gamma_0))[0, 0]
        out = (coefs, (float(inv_alpha0), float(sigma0), float(sigma_x)))
        self._coef_cache = (key, self.gamma.gamma._version, out)
        return out

    # ---- one reverse step / decode, host-driven (edm.py:178-242) --------------------------------------
    def sample_p_zs_given_zt_only_linker(self, s, t, z_t, node_mask, fragment_mask, linker_mask, edge_mask, context):
        """Samples z_s ~ p(z_s | z_t) for the linker atoms (edm.py:178-208): HIP denoiser + fused HIP tail."""
        gamma_s, gamma_t = self.gamma(s), self.gamma(t)
        sigma2_ts, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(gamma_t, gamma_s, z_t)
        sigma_s, sigma_t = self.sigma(gamma_s, target_tensor=z_t), self.sigma(gamma_t, target_tensor=z_t)
        eps_hat = self.dynamics.forward(xh=z_t, t=t, node_mask=node_mask, linker_mask=linker_mask,
                                        context=context, edge_mask=edge_mask)
        c_eps = sigma2_ts / alpha_ts / sigma_t
        sigma = sigma_ts * sigma_s / sigma_t
        bs, n, d = z_t.shape
        noise = torch.cat([torch.randn((bs, n, self.n_dims), device=z_t.device),
                           torch.randn((bs, n, self.in_node_nf), device=z_t.device)], dim=2)
        # the reference broadcasts per-sample scalars that are identical across the batch (edm.py:147-150)
        coef = _lib.DLStepCoef(float(t.reshape(-1)[0]), float(alpha_ts.reshape(-1)[0]), float(c_eps.reshape(-1)[0]),
                               float(sigma.reshape(-1)[0]))
        return self._sampler_step(z_t, eps_hat, noise, fragment_mask, linker_mask, coef)

    def _sampler_step(self, z_t, eps_hat, noise, fragment_mask, linker_mask, coef):
        lib = _lib.load()
        if z_t.device.type != 'cuda':
            raise RuntimeError('difflinker_amd.EDM runs on the GPU only (no CPU fallback)')
        bs, n, d = z_t.shape
        z_t, eps_hat, noise = z_t.float().contiguous(), eps_hat.float().contiguous(), noise.float().contiguous()
        fm = fragment_mask.reshape(bs, n).float().contiguous()
        lm = linker_mask.reshape(bs, n).float().contiguous()
        z_s = torch.empty_like(z_t)
        with torch.cuda.device(z_t.device):
            stream = torch.cuda.current_stream(z_t.device).cuda_stream
            _lib.check(lib.dl_sampler_step(bs, n, d, _lib.ptr(z_t), _lib.ptr(eps_hat), _lib.ptr(noise), _lib.ptr(fm),
                                           _lib.ptr(lm), coef, _lib.ptr(z_s), ctypes.c_void_p(stream)),
                       'dl_sampler_step')
        return z_s

    def compute_x_pred(self, eps_t, z_t, gamma_t):
        sigma_t = self.sigma(gamma_t, target_tensor=eps_t)
        alpha_t = self.alpha(gamma_t, target_tensor=eps_t)
        return 1. / alpha_t * (z_t - sigma_t * eps_t)

    def sample_p_xh_given_z0_only_linker(self, z_0, node_mask, fragment_mask, linker_mask, edge_mask, context):
        """Samples x,h ~ p(x,h | z_0) for the linker atoms (edm.py:210-235), host-driven."""
        zeros = torch.zeros(size=(z_0.size(0), 1), device=z_0.device)
        gamma_0 = self.gamma(zeros)
        sigma_x = self.SNR(-0.5 * gamma_0).unsqueeze(1)
        eps_hat = self.dynamics.forward(t=zeros, xh=z_0, node_mask=node_mask, linker_mask=linker_mask,
                                        edge_mask=edge_mask, context=context)
        eps_hat = eps_hat * linker_mask
        mu_x = self.compute_x_pred(eps_t=eps_hat, z_t=z_0, gamma_t=gamma_0)
        xh = self.sample_normal(mu=mu_x, sigma=sigma_x, node_mask=linker_mask)
        xh = z_0 * fragment_mask + xh * linker_mask
        x, h = self.unnormalize(xh[:, :, :self.n_dims], xh[:, :, self.n_dims:])
        h = F.one_hot(torch.argmax(h, dim=2), self.in_node_nf) * node_mask
        return x, h

    # ---- the chain ------------------------------------------------------------------------------------
    def _fused_ok(self):
        return isinstance(self.dynamics, Dynamics) and not isinstance(self.dynamics, DynamicsWithPockets) \
            and self.dynamics.graph_type == 'FC' and not self.dynamics.centering and not self.dynamics.sin_embedding

    @torch.no_grad()
    def sample_chain(self, x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames=None,
                     noise_bank=None, mol_offset=0):
        """``EDM.sample_chain`` (edm.py:126-176).  Returns ``chain [keep_frames, B, N, 3+nf]`` whose frame 0
        is the final sample ``[x, one_hot(h)]``.

        ``noise_bank`` (optional, not in the reference signature) = ``(noise_x [T+2,B,N,3], noise_h
        [T+2,B,N,nf])`` replaces the internal ``torch.randn`` draws (parity tests share one bank with the
        CPU oracle).  ``mol_offset``: global index of the first molecule of this batch (``noise_source='philox'``
        with a batch sharded over ranks).
        """
        if keep_frames is None:
            keep_frames = self.T
        else:
            assert keep_frames <= self.T
        bs, n = x.size(0), x.size(1)
        if bs == 0:                                 # an empty batch is an empty chain in the reference (every op runs on empty tensors)
            return torch.zeros((keep_frames, 0, n, self.n_dims + self.in_node_nf), dtype=x.dtype, device=x.device)
        if not self._fused_ok():
            philox_draws = None
            if noise_bank is None and self.noise_source == 'philox':
                philox_draws = (int(self.noise_seed) & 0xFFFFFFFFFFFFFFFF, int(mol_offset))   # one draw per step, no bank
                self.noise_seed = int(self.noise_seed) + 1
            # a centering / sin_embedding Dynamics may still pick a team for its forward calls (small batch, 56..110 atoms): should
            # one fail to assemble, the chain is sampled again without teams - from the SAME draws (generator state restored) - and
            # nothing but FoundNaNException reaches the caller
            if not hasattr(self.dynamics, 'without_teams'):
                return self._sample_chain_host_loop(x, h, node_mask, fragment_mask, linker_mask, edge_mask, context,
                                                    keep_frames, noise_bank, philox_draws)
            # team sizes of the denoiser calls: those of the WHOLE batch (a shard pins EDM.team_batch), as on the fused chain
            self.dynamics.team_batch = bs if self.team_batch is None else max(bs, int(self.team_batch))
            rng = None
            if noise_bank is None and philox_draws is None and x.device.type == 'cuda':
                rng = torch.cuda.get_rng_state(x.device)

            def host_loop():
                if rng is not None:
                    torch.cuda.set_rng_state(rng, x.device)
                return self._sample_chain_host_loop(x, h, node_mask, fragment_mask, linker_mask, edge_mask, context,
                                                    keep_frames, noise_bank, philox_draws)
            try:
                return self.dynamics.without_teams(host_loop)
            finally:
                self.dynamics.team_batch = None
        dev = x.device
        if dev.type != 'cuda':
            raise RuntimeError('difflinker_amd.EDM.sample_chain runs on the GPU only (HIP kernels, no CPU fallback)')
        philox = noise_bank is None and self.noise_source == 'philox'
        seed = 0
        if philox:
            seed = int(self.noise_seed) & 0xFFFFFFFFFFFFFFFF
            self.noise_seed = int(self.noise_seed) + 1
        elif noise_bank is None:
            noise_bank = self.draw_noise_bank(bs, n, dev)       # the reference's call sequence, for the whole batch
        else:
            noise_bank = tuple(t.to(dev, torch.float32).contiguous() for t in noise_bank)
            assert tuple(noise_bank[0].shape) == (self.T + 2, bs, n, self.n_dims) and \
                tuple(noise_bank[1].shape) == (self.T + 2, bs, n, self.in_node_nf)
        # the draws are fixed by now: should a team of workgroups fail to assemble (another kernel holding compute units), the
        # same chain is sampled again on one compute unit per molecule - nothing but FoundNaNException reaches the caller
        return self.dynamics.without_teams(lambda: self._sample_chain_by_size(
            x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames, noise_bank, philox, seed, mol_offset))

    def _sample_chain_by_size(self, x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames, noise_bank,
                              philox, seed, mol_offset):
        """Molecules of <= 55 atoms: the fused chain with one compute unit (or a team) per molecule; up to 110: the fused chain
        with a team of at least two, in pieces the chip holds at once; beyond: the HBM-resident kernels under a host-driven
        loop.  Noise rows and per-step scalars are those of the WHOLE batch, so every molecule gets the sample it would get
        from any of the paths alone."""
        bs, n = x.size(0), x.size(1)
        dev = x.device
        dyn = self.dynamics
        small, med, big = dyn.size_classes(node_mask.reshape(bs, n).to(torch.int8))
        if small is None and med is None and big is None:
            return self._sample_chain_fused(x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames,
                                            noise_bank, seed, mol_offset, None)
        em = edge_mask.reshape(bs, n * n) if edge_mask is not None else None

        def part(idx):
            return dict(x=x[idx], h=h[idx], node_mask=node_mask[idx], fragment_mask=fragment_mask[idx], linker_mask=linker_mask[idx],
                        edge_mask=em[idx].reshape(-1, 1) if em is not None else None,
                        context=context[idx] if context is not None else None)
        pinned, pinned_team = self.coef_batch, self.team_batch
        if pinned is None:
            self.coef_batch = bs
        if pinned_team is None:
            # the team sizes of every part follow the size of the WHOLE batch, exactly as in a shard of it
            # (distributed.sample_chain_sharded pins the same number; a pinned EDM.team_batch also keeps the chain in ONE launch
            # - no split_plan, no surplus teams): world = 1 and world > 1 sample the same bits for every molecule that the
            # unsharded call runs in one launch as well (a batch beyond the number of compute units, or split_chain = False)
            self.team_batch = bs
        try:
            chain = torch.zeros((keep_frames, bs, n, self.n_dims + self.in_node_nf), device=dev)
            fused = []
            if small is not None:
                fused.append((small, None))
            if med is not None:
                # team size of the 56..110-atom molecules: that of the WHOLE (possibly sharded) batch when the caller pinned it
                # (EDM.team_batch, distributed.sample_chain_sharded) - an atom's messages are summed in a team-size dependent
                # order, so a sample must not depend on how the batch was split - else of the piece at hand
                def med_team(count):
                    return max(2, dyn.team_for_size(max(count, int(self.team_batch)), dev))
                fused += [(c, med_team(int(c.numel()))) for c in dyn.team_chunks(med, dev)]
            # every part reports NaNs in ITS numbering, for ITS first offending denoiser call; the reference raises at the first
            # call whose output holds a NaN with the molecules that are NaN THERE (egnn.py:441-442) and its callers index the batch
            # with the sets of the exception (lightning.py:353-361): collect the sets in whole-batch numbering with their call
            # index, keep those of the earliest call, raise once.  Every part runs even after one has failed at call 0: the
            # molecules of another part that are NaN at call 0 too belong to the sets the reference reports (the
            # exception is rare and its callers re-sample the batch anyway, generate.py:154-161)
            nan_sets = []

            def run_part(idx, fn):
                try:
                    return fn()
                except utils.FoundNaNException as e:
                    rows = idx.tolist()
                    nan_sets.append((int(getattr(e, 'first_step', 0)),
                                     tuple({rows[k] for k in s_} for s_ in (e.x_h_nan_idx, e.only_x_nan_idx, e.only_h_nan_idx,
                                                                            getattr(e, 'f16_range_idx', ())))))
                    return None
            for idx, team in fused:
                bank = None if philox else (noise_bank[0][:, idx].contiguous(), noise_bank[1][:, idx].contiguous())
                got = run_part(idx, lambda: self._sample_chain_fused(keep_frames=keep_frames, noise_bank=bank, seed=seed,
                                                                     mol_offset=mol_offset, mol_index=idx.to(torch.int32).contiguous(),
                                                                     team=team, **part(idx)))
                if got is not None:
                    chain[:, idx] = got
            if big is not None:
                bank = None if philox else (noise_bank[0][:, big], noise_bank[1][:, big])
                large = run_part(big, lambda: self._sample_chain_host_loop(
                    keep_frames=keep_frames, noise_bank=bank, philox_draws=(seed, int(mol_offset), big, bs) if philox else None,
                    **part(big)))
                if large is not None:
                    chain[:, big] = large.to(chain.dtype)
            if nan_sets:
                first = min(step for step, _ in nan_sets)
                err = utils.FoundNaNException.from_index_sets(*(set().union(*(s_[k] for step, s_ in nan_sets if step == first))
                                                                for k in range(4)))
                err.first_step = first
                raise err
        finally:
            self.coef_batch = pinned
            self.team_batch = pinned_team
        return chain

    def _sample_chain_fused(self, x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames, noise_bank,
                            seed, mol_offset, mol_index, team=None):
        """The whole chain as ONE launch (``dl_sample_chain_fc``).  ``noise_bank`` = device tensors, or None: draws generated
        in the kernel from ``seed`` and the molecules' global indices ``mol_offset + mol_index[b]`` (``mol_index`` None: b)."""
        dev = x.device
        lib = _lib.load()
        bs, n = x.size(0), x.size(1)
        nf, T = self.in_node_nf, self.T
        handle = self.dynamics.hip_model(dev)
        philox = noise_bank is None
        noise_x, noise_h = (None, None) if philox else noise_bank
        coefs, (inv_alpha0, sigma0, sigma_x) = self.step_coefficients(bs)
        coefs = coefs.to(dev)
        f32 = lambda t_, shape: t_.reshape(shape).to(torch.float32).contiguous()
        xs, hs = f32(x, (bs, n, self.n_dims)), f32(h, (bs, n, nf))
        nm = node_mask.reshape(bs, n).to(torch.int8).contiguous()
        fm, lm = f32(fragment_mask, (bs, n)), f32(linker_mask, (bs, n))
        em = edge_mask.reshape(bs, n, n).to(torch.int8).contiguous() if edge_mask is not None else None
        ctx = f32(context, (bs, n, self.dynamics.context_node_nf)) if context is not None else None
        # longest-processing-time order: a molecule holds one compute unit for the whole chain and a workgroup's cost grows
        # with n_b^2, so the big ones are launched first (matters once the batch exceeds the number of compute units)
        n_real = nm.ne(0).sum(1)
        order = torch.argsort(n_real, descending=True, stable=True).to(torch.int32).contiguous()
        if XCD_AWARE_ORDER and bs <= compute_units(dev) and bs >= 16:
            # a batch that fits the chip in one wave of workgroups: workgroup k runs on XCD k % 8 (observed dispatch pattern; only
            # speed depends on it), and the compute units of an XCD share one 4 MB L2 through which each of them streams the same
            # 5.8 MB of weights per forward.  Molecules of SIMILAR SIZE on one XCD stay in phase, and a weight chunk one of them
            # pulled in is still there when the others ask: deal the size-sorted molecules to the XCDs in contiguous runs
            if XCD_ORDER_MODE >= 2:
                # ... R runs per XCD (2: one from the big end and one from the small end of the sorted batch), dealt in a zig-zag:
                # the same load on every XCD
                R = XCD_ORDER_MODE
                ln = -(-bs // (8 * R))
                k = torch.arange(8 * R * ln, device=dev)
                x, j = k % 8, k // 8
                r = j // ln                                   # which of the XCD's runs
                run = torch.where(r % 2 == 0, 8 * r + x, 8 * r + 7 - x)
                idx = run * ln + j % ln
            else:
                per = -(-bs // 8)
                k = torch.arange(8 * per, device=dev)
                idx = (k % 8) * per + k // 8
            order = order[idx[idx < bs]].contiguous()
        chain = torch.zeros((keep_frames, bs, n, self.n_dims + nf), device=dev)
        flags = torch.zeros(bs, dtype=torch.int32, device=dev)
        steps = torch.full((bs,), -1, dtype=torch.int32, device=dev)
        # a batch smaller than the chip: several compute units per molecule (Dynamics.team)
        if team is None:
            team = 1 if self.dynamics._no_teams else \
                self.dynamics.team_for(bs if self.team_batch is None else max(bs, int(self.team_batch)), dev)
        # A batch just above the number of compute units: with one compute unit per molecule for the whole chain the molecules
        # beyond that number start only when others have finished (measured round 4: B = 257 costs 1.3x B = 256 on 256 compute
        # units).  The few molecules over - the smallest, last in the longest-first order - go to TEAMS of compute units in a
        # second launch on another stream instead: its workgroups take the compute units the smallest molecules of the first
        # launch leave after ~0.6 of the chain, and a team of four samples a 35-atom molecule in a quarter of the chain.
        over = 0
        if team == 1 and self.overflow_teams and self.team_batch is None and not self.dynamics._no_teams:
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            if cus < bs <= cus + cus // 4:
                over = bs - cus
                team2 = 4 if over <= cus // 16 else 2
                # (the entry point refuses a team the device cannot hold at once - its round-up to whole groups of eight molecules
                # may not fit when the number of compute units is no multiple of 16: then everybody keeps one compute unit)
                while team2 > 1 and int(lib.dl_team_max(over)) < team2:
                    team2 //= 2
                if team2 < 2:
                    over = 0
            elif cus < bs < 2 * cus and not getattr(EDM, '_warned_off_sweet_spot', False):
                EDM._warned_off_sweet_spot = True              # say so once
                warnings.warn(f'EDM.sample_chain: a batch of {bs} molecules on {cus} compute units runs one molecule per compute unit '
                              f'for the whole chain, so the {bs - cus} molecules beyond {cus} start only when others have finished; '
                              f'batches of k x {cus} molecules (or up to {cus + cus // 4}, whose surplus is sampled by teams beside '
                              f'the rest) use the chip evenly', RuntimeWarning, stacklevel=3)
        ws, ws_bytes = self.dynamics.workspace(bs - over, team, dev)

        # the static hand-over inside a ragged batch that fills the chip (split_plan): two launches
        # (not for a shard of a batch - EDM.team_batch pinned, distributed.sample_chain_sharded: the plan follows the sizes of the
        # molecules at hand, and a molecule's sample must not depend on how its batch was split - SURVEY 8e)
        plan = None
        if team == 1 and over == 0 and self.split_chain and self.team_batch is None and not self.dynamics._no_teams \
                and bs <= compute_units(dev) and T >= 20:
            counts = torch.stack([n_real, lm.ne(0).sum(1)]).cpu()             # ONE device-to-host copy: sizes and linker sizes
            sizes_h, linkers_h = counts[0].tolist(), counts[1].tolist()
            if max(sizes_h) <= int(lib.dl_max_atoms()) and min(sizes_h) > 0:
                sub_ = int(getattr(self.dynamics, 'inv_sublayers', 2))
                if int(self.split_stages) > 2:
                    plan = split_plan_stages(sizes_h, linkers_h, T + 1, compute_units(dev), self.dynamics.n_layers, sub_,
                                             max_stages=int(self.split_stages))
                else:
                    two = split_plan(sizes_h, linkers_h, T + 1, compute_units(dev), self.dynamics.n_layers, sub_,
                                     allow_singles=bool(self.split_singles))
                    # (as a list of stages: everybody on one compute unit up to q_end, then the teams and the single ones to the end)
                    plan = None if two is None else [(two[0], [], list(range(bs))), ([T + 1] * bs, two[1], two[2])]
                if plan is not None and any(t_ and int(lib.dl_team_max(len(t_))) < 2 for _, t_, _ in plan):
                    plan = None                                               # the teams of a stage would not fit at once
        q_end_t = z_state = None
        if plan is not None:
            q_end_t = torch.tensor(plan[0][0], dtype=torch.int32, device=dev)
            z_state = torch.empty((bs, n, self.n_dims + nf), device=dev)

        def chain_args(flags_, steps_, team_, ws_, ws_bytes_, first, count, q_begin=None, q_end=None, skip=None, order_=None):
            return _lib.DLChainArgs(
                q_begin=q_begin.data_ptr() if q_begin is not None else None, q_end=q_end.data_ptr() if q_end is not None else None,
                z_state=z_state.data_ptr() if z_state is not None else None, skip_flags=skip.data_ptr() if skip is not None else None,
                B=bs, N=n, T=T, keep_frames=keep_frames,
                x=xs.data_ptr(), h=hs.data_ptr(), node_mask=nm.data_ptr(), fragment_mask=fm.data_ptr(),
                linker_mask=lm.data_ptr(), edge_mask=em.data_ptr() if em is not None else None,
                context=ctx.data_ptr() if ctx is not None else None,
                noise_x=None if philox else noise_x.data_ptr(), noise_h=None if philox else noise_h.data_ptr(),
                noise_seed=seed, mol_offset=int(mol_offset), team=team_, coefs=coefs.data_ptr(),
                inv_alpha0=inv_alpha0, sigma0=sigma0, sigma_x=sigma_x,
                norm_x=float(self.norm_values[0]), norm_h=float(self.norm_values[1]), bias_h=float(self.norm_biases[1]),
                chain=chain.data_ptr(), nan_flags=flags_.data_ptr(), nan_step=steps_.data_ptr(),
                order=(order if order_ is None else order_).data_ptr(),
                workspace=ws_.data_ptr(), workspace_bytes=ws_bytes_,
                mol_index=mol_index.data_ptr() if mol_index is not None else None, order_first=first, order_count=count)
        args = chain_args(flags, steps, team, ws, ws_bytes, 0, bs - over if over else 0, q_end=q_end_t)
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            if getattr(self, 'profile_events', False):     # bench.py: HIP events right around the launch
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(cur)
            if over:
                # (its own flag arrays: the team entry point clears them on its stream; its own workspace: the launches overlap)
                side = self._side_stream(dev)
                flags2 = torch.zeros(bs, dtype=torch.int32, device=dev)
                steps2 = torch.full((bs,), -1, dtype=torch.int32, device=dev)
                need2 = int(lib.dl_workspace_bytes(over, team2))
                ws2 = self._side_workspace((dev.index, 'teams'), need2, dev)
                args2 = chain_args(flags2, steps2, team2, ws2, need2, bs - over, over)
                ready = torch.cuda.Event()
                ready.record(cur)                          # inputs, coefficients, noise bank: enqueued on the current stream
            _lib.check(lib.dl_sample_chain_fc(handle, ctypes.byref(args), ctypes.c_void_p(cur.cuda_stream)),
                       'dl_sample_chain_fc')
            if over:
                side.wait_event(ready)
                _lib.check(lib.dl_sample_chain_fc(handle, ctypes.byref(args2), ctypes.c_void_p(side.cuda_stream)),
                           'dl_sample_chain_fc (molecules beyond one per compute unit, on teams)')
                done = torch.cuda.Event()
                done.record(side)
                cur.wait_event(done)
                for t_ in (xs, hs, nm, fm, lm, em, ctx, coefs, order, chain, noise_x, noise_h, mol_index):
                    if t_ is not None:
                        t_.record_stream(side)             # the caching allocator must not hand these out while the side launch runs
                flags = flags | flags2
                steps = torch.where(steps2 >= 0, steps2, steps)
            if plan is not None:
                # later stages: the molecules with the most work left resume from z_state on teams of two (a launch on the side stream),
                # the others on one compute unit each (a launch on this stream) - side by side, both behind the stage before; a
                # molecule that ended in an earlier launch (NaN) is skipped
                by_size = lambda idx: torch.tensor(sorted(idx, key=lambda b_: (-sizes_h[b_], b_)), dtype=torch.int32, device=dev)  # noqa: E731
                after_first = torch.cuda.Event(enable_timing=bool(getattr(self, 'profile_events', False)))
                after_first.record(cur)
                self.last_split_event = after_first
                side = self._side_stream(dev)
                q_begin_t = q_end_t
                for si in range(1, len(plan)):
                    q_end_list, teams_, singles_ = plan[si]
                    last = si == len(plan) - 1
                    q_end_s = None if last else torch.tensor(q_end_list, dtype=torch.int32, device=dev)
                    boundary = torch.cuda.Event()
                    boundary.record(cur)               # everything of the stages before is behind this point of the current stream
                    parts = []
                    if teams_:
                        rest, m2 = by_size(teams_), len(teams_)
                        flags2 = torch.zeros(bs, dtype=torch.int32, device=dev)
                        steps2 = torch.full((bs,), -1, dtype=torch.int32, device=dev)
                        need2 = int(lib.dl_workspace_bytes(m2, 2))
                        ws2 = self._side_workspace((dev.index, 'teams'), need2, dev)
                        args2 = chain_args(flags2, steps2, 2, ws2, need2, 0, m2, q_begin=q_begin_t, q_end=q_end_s, skip=flags, order_=rest)
                        side.wait_event(boundary)
                        _lib.check(lib.dl_sample_chain_fc(handle, ctypes.byref(args2), ctypes.c_void_p(side.cuda_stream)),
                                   'dl_sample_chain_fc (a later stage of a split chain: teams of two)')
                        done2 = torch.cuda.Event()
                        done2.record(side)
                        parts.append((flags2, steps2, done2, rest))
                    if singles_:
                        rest1, m1 = by_size(singles_), len(singles_)
                        flags3 = torch.zeros(bs, dtype=torch.int32, device=dev)
                        steps3 = torch.full((bs,), -1, dtype=torch.int32, device=dev)
                        need3 = int(lib.dl_workspace_bytes(m1, 1))
                        ws3 = self._side_workspace((dev.index, 'singles'), need3, dev)
                        args3 = chain_args(flags3, steps3, 1, ws3, need3, 0, m1, q_begin=q_begin_t, q_end=q_end_s, skip=flags, order_=rest1)
                        _lib.check(lib.dl_sample_chain_fc(handle, ctypes.byref(args3), ctypes.c_void_p(cur.cuda_stream)),
                                   'dl_sample_chain_fc (a later stage of a split chain: one compute unit each)')
                        parts.append((flags3, steps3, None, rest1))
                    for f_, s_, done_, rest_ in parts:
                        if done_ is not None:
                            cur.wait_event(done_)
                            for t_ in (xs, hs, nm, fm, lm, em, ctx, coefs, rest_, chain, noise_x, noise_h, mol_index, q_begin_t, q_end_s, z_state, flags):
                                if t_ is not None:
                                    t_.record_stream(side)
                        # (the single-compute-unit kernel initialises its molecules' words itself: 0 / -1; merge only what the phase set)
                        flags = flags | f_
                        steps = torch.where(s_ >= 0, s_, steps)
                    q_begin_t = q_end_s
            if getattr(self, 'profile_events', False):
                ev1.record(cur)
                self.last_kernel_events = (ev0, ev1)
        self._raise_on_chain_flags(flags, steps)
        bias = float(self.norm_biases[1])
        if bias != 0.0 and keep_frames > 1:
            # the kernel writes the rows of real atoms; the reference's intermediate frames are unnormalize_z of the WHOLE z
            # (edm.py:357-361), whose padding rows - zeros - come out as the feature bias (no released configuration has one)
            chain[1:, :, :, self.n_dims:] += bias * (1.0 - node_mask.to(chain.dtype).reshape(bs, n, 1))
        return chain

    def philox_noise_bank(self, n_samples, n_nodes, device, mol_offset=0, seed=None, n_draws=None):
        """The in-kernel stream as a bank ``(noise_x [T+2,B,N,3], noise_h [T+2,B,N,nf])`` (``dl_philox_fill``; ``n_draws``: another
        number of draws); advances ``noise_seed`` unless ``seed`` is given."""
        if torch.device(device).type != 'cuda':
            raise RuntimeError('the Philox noise bank is generated on the GPU (no CPU fallback)')
        if seed is None:
            seed = int(self.noise_seed)
            self.noise_seed = seed + 1
        n_draws = self.T + 2 if n_draws is None else int(n_draws)
        noise_x = torch.empty((n_draws, n_samples, n_nodes, self.n_dims), device=device)
        noise_h = torch.empty((n_draws, n_samples, n_nodes, self.in_node_nf), device=device)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(_lib.load().dl_philox_fill(int(seed) & 0xFFFFFFFFFFFFFFFF, int(mol_offset), None, n_samples, n_nodes,
                                                  self.in_node_nf, 0, n_draws, noise_x.data_ptr(), noise_h.data_ptr(),
                                                  ctypes.c_void_p(stream)), 'dl_philox_fill')
        return noise_x, noise_h

    def _raise_on_chain_flags(self, flags, steps):
        """The reference raises at the first denoiser call whose output holds a NaN (egnn.py:441-442);
        the fused chain reports, per molecule, its first offending call — raise for the earliest one."""
        if bool(flags.any()):
            f, st = flags.cpu(), steps.cpu()
            if bool((f & 8).any()):
                from .egnn import TeamNotAssembled
                raise TeamNotAssembled('a team of workgroups did not assemble in time (another kernel held compute units)')
            if bool((f & 4).any()):
                raise ValueError('molecule with more real atoms than the LDS-resident fully-connected kernels take')
            first = int(st[f != 0].min())
            err = utils.FoundNaNException.from_flags(torch.where((st == first) & (f != 0), f, torch.zeros_like(f)))
            err.first_step = first          # the denoiser call (0 .. T) the exception stands for: a batch sampled in parts keeps the earliest
            raise err

    def _philox_draw(self, seed, mol_offset, k, n_samples, n_nodes, device, mol_index=None):
        """Draw number ``k`` of the in-kernel stream as one ``[B,N,3+nf]`` tensor (``dl_philox_fill`` with one draw);
        ``mol_index`` (device int32 ``[B]``): the rows of these molecules of a larger batch, and nothing else."""
        nx = torch.empty((1, n_samples, n_nodes, self.n_dims), device=device)
        nh = torch.empty((1, n_samples, n_nodes, self.in_node_nf), device=device)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(_lib.load().dl_philox_fill(seed, mol_offset, _lib.ptr(mol_index), n_samples, n_nodes, self.in_node_nf, k, 1,
                                                  nx.data_ptr(), nh.data_ptr(), ctypes.c_void_p(stream)), 'dl_philox_fill')
        return torch.cat([nx[0], nh[0]], dim=2)

    def _sample_chain_host_loop(self, x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames,
                                noise_bank=None, philox_draws=None):
        """Reference-shaped loop for dynamics the fused kernel does not cover (pockets): one HIP denoiser call
        and one fused HIP tail per step (edm.py:126-176).  Noise: the reference's ``torch.randn`` call order,
        or the given bank ``(noise_x [T+2,B,N,3], noise_h [T+2,B,N,nf])``."""
        n_samples, n_nodes = x.size(0), x.size(1)
        dev = x.device
        draw_idx = [0]

        def draw():
            k = draw_idx[0]
            draw_idx[0] += 1
            if noise_bank is not None:
                return torch.cat([noise_bank[0][k].to(dev, torch.float32), noise_bank[1][k].to(dev, torch.float32)], dim=2)
            if philox_draws is not None:
                if len(philox_draws) == 4:       # (seed, mol_offset, rows, size of the whole batch): a non-contiguous part of it
                    rows = philox_draws[2].to(torch.int32).contiguous()
                    return self._philox_draw(philox_draws[0], philox_draws[1], k, n_samples, n_nodes, dev, mol_index=rows)
                return self._philox_draw(philox_draws[0], philox_draws[1], k, n_samples, n_nodes, dev)
            return torch.cat([torch.randn((n_samples, n_nodes, self.n_dims), device=dev),
                              torch.randn((n_samples, n_nodes, self.in_node_nf), device=dev)], dim=2)

        x, h = self.normalize(x, h)
        xh = torch.cat([x, h], dim=2)
        z = xh * fragment_mask + (draw() * linker_mask) * linker_mask
        chain = torch.zeros((keep_frames,) + z.size(), device=dev)
        coefs, (inv_alpha0, sigma0, sigma_x) = self.step_coefficients(n_samples)
        # pocket dynamics: convert / check the masks once per chain and only enqueue kernels per step; the NaN flags of
        # every call are OR-ed on the device and examined once at the end (a NaN never heals along the chain)
        prep = self.dynamics.prepare(node_mask, linker_mask, edge_mask, context) if hasattr(self.dynamics, 'prepare') else None
        seen = torch.zeros(n_samples, dtype=torch.int32, device=dev) if prep is not None else None
        first_bad = torch.full((n_samples,), -1, dtype=torch.int32, device=dev) if prep is not None else None

        def denoise(z_, t_arr_, q_):
            if prep is None:
                return self.dynamics.forward(xh=z_, t=t_arr_, node_mask=node_mask, linker_mask=linker_mask,
                                             context=context, edge_mask=edge_mask)
            out, flags = self.dynamics.launch(prep, t_arr_, z_)
            if q_ == 0 and not isinstance(self.dynamics, DynamicsWithPockets) and not self.dynamics._no_teams:
                # a Dynamics may run this loop on teams of workgroups (small batch, 56..110 atoms): should they fail to assemble
                # (a co-tenant kernel holding compute units), say so after the FIRST launch - every further one would spin to the
                # limit as well - and let without_teams() repeat the chain on one compute unit per molecule
                if bool((flags & 8).any()):
                    from .egnn import TeamNotAssembled
                    raise TeamNotAssembled('a team of workgroups did not assemble in time (another kernel held compute units)')
            newly = (flags != 0) & (seen == 0)
            first_bad.masked_fill_(newly, q_)
            seen.bitwise_or_(flags)
            return out

        t_arr = torch.empty((n_samples, 1), device=dev)
        for q, s in enumerate(reversed(range(0, self.T))):
            t_, a_, c_, sg_ = (float(v) for v in coefs[q])
            t_arr.fill_(t_)
            eps_hat = denoise(z, t_arr, q)
            z = self._sampler_step(z, eps_hat, draw(), fragment_mask, linker_mask, _lib.DLStepCoef(t_, a_, c_, sg_))
            widx = (s * keep_frames) // self.T
            if s == 0 or ((s - 1) * keep_frames) // self.T != widx:       # only the last writer of a frame survives
                chain[widx] = self.unnormalize_z(z)
        # final decode (edm.py:210-235)
        zeros = torch.zeros(size=(n_samples, 1), device=dev)
        eps_hat = denoise(z, zeros, self.T) * linker_mask
        if prep is not None:
            self._raise_on_chain_flags(seen, first_bad)
        mu_x = inv_alpha0 * (z - sigma0 * eps_hat)
        xh = mu_x + sigma_x * (draw() * linker_mask)
        xh = z * fragment_mask + xh * linker_mask
        x, h = self.unnormalize(xh[:, :, :self.n_dims], xh[:, :, self.n_dims:])
        h = F.one_hot(torch.argmax(h, dim=2), self.in_node_nf) * node_mask
        chain[0] = torch.cat([x, h], dim=2)
        return chain


class InpaintingEDM(EDM):
    """``InpaintingEDM`` (edm.py:466-727), sampling side: every atom is denoised (``linker_mask=None``, centred dynamics),
    the linker atoms keep the ``p(z_s | z_t)`` sample, the fragment atoms are re-drawn from ``q(z_s | z_t, x)``, and the
    centre of gravity is projected out after every step.  One HIP denoiser call + one fused HIP tail
    (``dl_inpaint_step``) per step; no released configuration uses it."""

    def inpaint_coefficients(self, batch_size=1):
        """Per-step scalars in execution order: ``(t, alpha_ts, c_eps, sigma, a_q, b_q)`` (edm.py:616-672), evaluated on
        ``[batch_size, 1]`` CPU tensors like ``EDM.step_coefficients``."""
        if self.coef_batch is not None:
            batch_size = self.coef_batch
        gamma = self.gamma.gamma.detach().to('cpu', torch.float32)
        timesteps = self.gamma.timesteps
        lookup = lambda tt: gamma[torch.round(tt * timesteps).long()]       # noqa: E731
        rows = []
        for s_int in reversed(range(0, self.T)):
            s = torch.full((batch_size, 1), fill_value=s_int)
            t = (s + 1) / self.T
            s = s / self.T
            gamma_s, gamma_t = lookup(s), lookup(t)
            sigma2_ts, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(gamma_t, gamma_s, gamma_t)
            sigma_s = torch.sqrt(torch.sigmoid(gamma_s))
            sigma_t = torch.sqrt(torch.sigmoid(gamma_t))
            alpha_s = torch.sqrt(torch.sigmoid(-gamma_s))
            c_eps = sigma2_ts / alpha_ts / sigma_t
            sigma = sigma_ts * sigma_s / sigma_t
            a_q = alpha_ts * (sigma_s ** 2) / (sigma_t ** 2)
            b_q = alpha_s * sigma2_ts / (sigma_t ** 2)
            rows.append([float(v[0, 0]) for v in (t, alpha_ts, c_eps, sigma, a_q, b_q)])
        return rows

    def draw_inpainting_noise_bank(self, n_samples, n_nodes, device):
        """The ``torch.randn`` calls of one chain in the reference's order: initial z, then (p, q) per step, then the p
        and q draws of the decode; each an x-part ``[B,N,3]`` and an h-part ``[B,N,nf]`` (utils.py:158-168,189-192)."""
        n = 1 + 2 * self.T + 2
        noise_x = torch.empty((n, n_samples, n_nodes, self.n_dims), device=device)
        noise_h = torch.empty((n, n_samples, n_nodes, self.in_node_nf), device=device)
        for k in range(n):
            torch.randn((n_samples, n_nodes, self.n_dims), out=noise_x[k])
            torch.randn((n_samples, n_nodes, self.in_node_nf), out=noise_h[k])
        return noise_x, noise_h

    @torch.no_grad()
    def sample_chain(self, x, h, node_mask, edge_mask, fragment_mask, linker_mask, context, keep_frames=None,
                     noise_bank=None, mol_offset=0):
        """``InpaintingEDM.sample_chain`` (edm.py:549-610).  Noise: the reference's ``torch.randn`` sequence, an explicit bank of
        ``1 + 2T + 2`` draws (initial z, then the p and q draw of every step, then the two of the decode), or - round 5 -
        ``noise_source='philox'``: that bank generated on the device from (``noise_seed``, GLOBAL molecule index = ``mol_offset`` +
        row, atom, draw), so a shard of a batch samples what the whole batch would (``distributed.sample_chain_sharded``); the
        ``torch.randn`` stream cannot be sharded (``mol_offset`` raises with it)."""
        philox = noise_bank is None and self.noise_source == 'philox'
        if mol_offset and not philox and noise_bank is None:
            raise NotImplementedError("a shard of a batch (mol_offset) needs noise_source='philox' or the shard's rows of an explicit "
                                      'noise_bank: the torch.randn stream is a property of the whole batch')
        dev = x.device
        if dev.type != 'cuda':
            raise RuntimeError('difflinker_amd.InpaintingEDM.sample_chain runs on the GPU only (HIP kernels, no CPU fallback)')
        if not getattr(self.dynamics, 'centering', False):
            raise ValueError('InpaintingEDM needs a centred denoiser (Dynamics(centering=True), lightning.py:99)')
        lib = _lib.load()
        bs, n = x.size(0), x.size(1)
        nf, T = self.in_node_nf, self.T
        keep_frames = T if keep_frames is None else keep_frames
        assert keep_frames <= T
        if bs == 0:
            return torch.zeros((keep_frames, 0, n, self.n_dims + nf), dtype=x.dtype, device=dev)
        if philox:
            noise_x, noise_h = self.philox_noise_bank(bs, n, dev, mol_offset=mol_offset, n_draws=1 + 2 * T + 2)
        elif noise_bank is None:
            noise_x, noise_h = self.draw_inpainting_noise_bank(bs, n, dev)
        else:
            noise_x, noise_h = (t_.to(dev, torch.float32).contiguous() for t_ in noise_bank)
            assert noise_x.shape[0] == 1 + 2 * T + 2
        # (the draws are fixed: a team of workgroups that cannot assemble means one more run on one compute unit per molecule;
        # team sizes follow the WHOLE batch when this is a shard of one: EDM.team_batch)
        self.dynamics.team_batch = bs if self.team_batch is None else max(bs, int(self.team_batch))
        try:
            return self.dynamics.without_teams(lambda: self._inpaint_chain(x, h, node_mask, edge_mask, fragment_mask, linker_mask,
                                                                           context, keep_frames, noise_x, noise_h))
        finally:
            self.dynamics.team_batch = None

    def _inpaint_chain(self, x, h, node_mask, edge_mask, fragment_mask, linker_mask, context, keep_frames, noise_x, noise_h):
        lib = _lib.load()
        dev = x.device
        bs, n = x.size(0), x.size(1)
        nf, T = self.in_node_nf, self.T
        f32 = lambda t_, shape: t_.reshape(shape).to(torch.float32).contiguous()      # noqa: E731
        nm, fm, lm = f32(node_mask, (bs, n)), f32(fragment_mask, (bs, n)), f32(linker_mask, (bs, n))
        xn, hn = self.normalize(x, h)
        xh = torch.cat([xn, hn], dim=2).float()
        xh_frag = (xh * fm.unsqueeze(-1)).contiguous()
        # initial z: centre-of-gravity-free position noise + feature noise on the node mask (edm.py:559, :715-727)
        nm3 = nm.unsqueeze(-1)
        zx = noise_x[0] * nm3
        zx = zx - (zx.sum(1, keepdim=True) / nm3.sum(1, keepdim=True)) * nm3
        z = torch.cat([zx, noise_h[0] * nm3], dim=2).contiguous()
        chain = torch.zeros((keep_frames,) + z.size(), device=dev)
        rows = self.inpaint_coefficients(bs)
        _, (inv_alpha0, sigma0, sigma_x) = self.step_coefficients(bs)
        prep = self.dynamics.prepare(node_mask, None, edge_mask, context)
        seen = torch.zeros(bs, dtype=torch.int32, device=dev)
        first_bad = torch.full((bs,), -1, dtype=torch.int32, device=dev)
        t_arr = torch.empty((bs, 1), device=dev)
        norm = dict(norm_x=float(self.norm_values[0]), norm_h=float(self.norm_values[1]), bias_h=float(self.norm_biases[1]))

        def tail(z_, eps_, k_p, k_q, coef):
            out = torch.empty_like(z_)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                _lib.check(lib.dl_inpaint_step(bs, n, nf, _lib.ptr(z_), _lib.ptr(eps_), _lib.ptr(xh_frag),
                                               _lib.ptr(noise_x[k_p]), _lib.ptr(noise_h[k_p]), _lib.ptr(noise_x[k_q]),
                                               _lib.ptr(noise_h[k_q]), _lib.ptr(nm), _lib.ptr(fm), _lib.ptr(lm), coef,
                                               _lib.ptr(out), ctypes.c_void_p(stream)), 'dl_inpaint_step')
            return out

        def denoise(z_, q_):
            eps_, flags = self.dynamics.launch(prep, t_arr, z_, center=False)      # dl_inpaint_step centres the velocity
            first_bad.masked_fill_((flags != 0) & (seen == 0), q_)
            seen.bitwise_or_(flags)
            return eps_

        for q, s in enumerate(reversed(range(0, T))):
            t_, a_, c_, sg_, aq_, bq_ = rows[q]
            t_arr.fill_(t_)
            eps_hat = denoise(z, q)
            coef = _lib.DLInpaintCoef(alpha_ts=a_, c_eps=c_, sigma=sg_, a_q=aq_, b_q=bq_, decode=0, inv_alpha0=0.0,
                                      sigma0=0.0, sigma_x=0.0, **norm)
            z = tail(z, eps_hat, 1 + 2 * q, 2 + 2 * q, coef)
            widx = (s * keep_frames) // T
            if s == 0 or ((s - 1) * keep_frames) // T != widx:
                chain[widx] = self.unnormalize_z(z)
        t_arr.fill_(0.0)
        eps_hat = denoise(z, T)
        self._raise_on_chain_flags(seen, first_bad)
        coef = _lib.DLInpaintCoef(alpha_ts=1.0, c_eps=0.0, sigma=0.0, a_q=0.0, b_q=0.0, decode=1, inv_alpha0=inv_alpha0,
                                  sigma0=sigma0, sigma_x=sigma_x, **norm)
        chain[0] = tail(z, eps_hat, 1 + 2 * T, 2 + 2 * T, coef)
        return chain
