"""Oracle restatement of the diffusion sampler (reference ``src/edm.py``, ``src/noise.py``).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  The sampler consumes noise
through an explicit ``noise_fn(size, mask) -> tensor`` so that the oracle, the
unmodified reference (patched ``utils.sample_gaussian_with_mask``) and the HIP
path can be fed the SAME noise bank: draw order is x-part ``[B,N,3]`` then
h-part ``[B,N,nf]``, once for the initial z, once per reverse step, once for the
final decode (edm.py:136,205,228 -> :328-345).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import egnn_oracle


# ----------------------------------------------------------------------------------------------
# noise schedule (noise.py)
# ----------------------------------------------------------------------------------------------
def polynomial_gamma_table(timesteps, precision, power):
    """gamma[t], t = 0..timesteps, for ``polynomial_<power>`` (noise.py:7-36,92-124).

    float64 numpy arithmetic, cast to fp32 at the end exactly like the reference.
    """
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas2 = (1 - np.power(x / steps, power)) ** 2
    # clip_noise_schedule, noise.py:7-19
    a = np.concatenate([np.ones(1), alphas2], axis=0)
    step = np.clip(a[1:] / a[:-1], a_min=0.001, a_max=1.)
    alphas2 = np.cumprod(step, axis=0)
    alphas2 = (1 - 2 * precision) * alphas2 + precision
    sigmas2 = 1 - alphas2
    gamma = -(np.log(alphas2) - np.log(sigmas2))
    return torch.from_numpy(gamma).float()


def gamma_lookup(gamma_table, t, timesteps):
    """``PredefinedNoiseSchedule.forward`` noise.py:126-128."""
    return gamma_table[torch.round(t * timesteps).long()]


# ----------------------------------------------------------------------------------------------
# sampler (edm.py)
# ----------------------------------------------------------------------------------------------
class EDMOracle:
    """Functional twin of ``EDM`` restricted to sampling (edm.py:126-242, 328-416).

    ``denoise(t, xh, node_mask, linker_mask, edge_mask, context)`` is the
    Dynamics forward; ``T`` may be overwritten after construction like
    ``generate.py:103-104`` does, the gamma table keeps its trained length.
    """

    def __init__(self, denoise, in_node_nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2',
                 noise_precision=1e-5, norm_values=(1., 4., 10.), norm_biases=(None, 0., 0.), dtype=torch.float32):
        kind, power = noise_schedule.split('_')
        assert kind == 'polynomial'
        self.gamma_table = polynomial_gamma_table(timesteps, noise_precision, float(power)).to(dtype)
        self.timesteps = timesteps
        self.T = timesteps
        self.denoise = denoise
        self.in_node_nf = in_node_nf
        self.n_dims = n_dims
        self.norm_values = norm_values
        self.norm_biases = norm_biases

    def gamma(self, t):
        return gamma_lookup(self.gamma_table, t, self.timesteps)

    @staticmethod
    def inflate(a, target):                                       # edm.py:409-416
        return a.view((a.size(0),) + (1,) * (target.dim() - 1))

    def sigma(self, g, target):                                   # edm.py:369-371
        return self.inflate(torch.sqrt(torch.sigmoid(g)), target)

    def alpha(self, g, target):                                   # edm.py:373-375
        return self.inflate(torch.sqrt(torch.sigmoid(-g)), target)

    def sigma_and_alpha_t_given_s(self, g_t, g_s, target):         # edm.py:381-403
        sigma2 = self.inflate(-torch.expm1(F.softplus(g_s) - F.softplus(g_t)), target)
        log_a2 = F.logsigmoid(-g_t) - F.logsigmoid(-g_s)
        alpha = self.inflate(torch.exp(0.5 * log_a2), target)
        return sigma2, torch.sqrt(sigma2), alpha

    def combined_noise(self, noise_fn, n_samples, n_nodes, mask):  # edm.py:328-340
        z_x = noise_fn((n_samples, n_nodes, self.n_dims), mask)
        z_h = noise_fn((n_samples, n_nodes, self.in_node_nf), mask)
        return torch.cat([z_x, z_h], dim=2)

    def normalize(self, x, h):                                     # edm.py:347-350
        return x / self.norm_values[0], (h.to(x.dtype) - self.norm_biases[1]) / self.norm_values[1]

    def unnormalize(self, x, h):                                   # edm.py:352-355
        return x * self.norm_values[0], h * self.norm_values[1] + self.norm_biases[1]

    def unnormalize_z(self, z):                                    # edm.py:357-361
        x, h = self.unnormalize(z[:, :, :self.n_dims], z[:, :, self.n_dims:])
        return torch.cat([x, h], dim=2)

    def step(self, s, t, z_t, node_mask, fragment_mask, linker_mask, edge_mask, context, noise_fn):
        """``sample_p_zs_given_zt_only_linker`` edm.py:178-208."""
        g_s, g_t = self.gamma(s), self.gamma(t)
        sigma2_ts, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(g_t, g_s, z_t)
        sigma_s, sigma_t = self.sigma(g_s, z_t), self.sigma(g_t, z_t)
        eps_hat = self.denoise(t, z_t, node_mask, linker_mask, edge_mask, context) * linker_mask
        mu = z_t / alpha_ts - (sigma2_ts / alpha_ts / sigma_t) * eps_hat
        sigma = sigma_ts * sigma_s / sigma_t
        z_s = mu + sigma * self.combined_noise(noise_fn, mu.size(0), mu.size(1), linker_mask)
        return z_t * fragment_mask + z_s * linker_mask

    def decode(self, z_0, node_mask, fragment_mask, linker_mask, edge_mask, context, noise_fn):
        """``sample_p_xh_given_z0_only_linker`` edm.py:210-242."""
        zeros = torch.zeros((z_0.size(0), 1), dtype=z_0.dtype, device=z_0.device)
        g_0 = self.gamma(zeros)
        sigma_x = torch.exp(-(-0.5 * g_0)).unsqueeze(1)            # SNR(-0.5*gamma_0), edm.py:216,377-379
        eps_hat = self.denoise(zeros, z_0, node_mask, linker_mask, edge_mask, context) * linker_mask
        mu_x = 1. / self.alpha(g_0, eps_hat) * (z_0 - self.sigma(g_0, eps_hat) * eps_hat)
        xh = mu_x + sigma_x * self.combined_noise(noise_fn, z_0.size(0), z_0.size(1), linker_mask)
        xh = z_0 * fragment_mask + xh * linker_mask
        x, h = self.unnormalize(xh[:, :, :self.n_dims], xh[:, :, self.n_dims:])
        h = F.one_hot(torch.argmax(h, dim=2), self.in_node_nf) * node_mask
        return x, h

    @torch.no_grad()
    def sample_chain(self, x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, noise_fn,
                     keep_frames=None):
        """``EDM.sample_chain`` edm.py:126-176."""
        n_samples, n_nodes = x.size(0), x.size(1)
        x, h = self.normalize(x, h)
        xh = torch.cat([x, h], dim=2)
        z = self.combined_noise(noise_fn, n_samples, n_nodes, linker_mask)
        z = xh * fragment_mask + z * linker_mask
        keep_frames = self.T if keep_frames is None else keep_frames
        assert keep_frames <= self.T
        chain = torch.zeros((keep_frames,) + z.size(), dtype=z.dtype, device=z.device)
        for s in reversed(range(0, self.T)):
            s_arr = torch.full((n_samples, 1), fill_value=s, device=z.device)
            t_arr = (s_arr + 1) / self.T
            s_arr = s_arr / self.T
            z = self.step(s_arr.to(z.dtype), t_arr.to(z.dtype), z, node_mask, fragment_mask, linker_mask,
                          edge_mask, context, noise_fn)
            chain[(s * keep_frames) // self.T] = self.unnormalize_z(z)
        x, h = self.decode(z, node_mask, fragment_mask, linker_mask, edge_mask, context, noise_fn)
        chain[0] = torch.cat([x, h.to(x.dtype)], dim=2)
        return chain


def remove_mean_with_mask(x, node_mask):
    """utils.py:56-63 (without its masking assert)."""
    n = node_mask.sum(1, keepdims=True)
    return x - (torch.sum(x, dim=1, keepdim=True) / n) * node_mask


class InpaintingEDMOracle(EDMOracle):
    """Functional twin of ``InpaintingEDM`` restricted to sampling (edm.py:549-727): every atom is denoised
    (``linker_mask=None``, centred dynamics), the fragment atoms are re-drawn from ``q(z_s | z_t, x)`` each step and the
    centre of gravity is projected out after every step; the position noise is centre-of-gravity free."""

    def combined_noise(self, noise_fn, n_samples, n_nodes, mask):  # edm.py:715-727
        z_x = remove_mean_with_mask(noise_fn((n_samples, n_nodes, self.n_dims), mask), mask)   # utils.py:158-168
        z_h = noise_fn((n_samples, n_nodes, self.in_node_nf), mask)
        return torch.cat([z_x, z_h], dim=2)

    def p_zs(self, s, t, z_t, node_mask, edge_mask, context, noise_fn):
        """edm.py:616-650."""
        g_s, g_t = self.gamma(s), self.gamma(t)
        sigma2_ts, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(g_t, g_s, z_t)
        sigma_s, sigma_t = self.sigma(g_s, z_t), self.sigma(g_t, z_t)
        eps_hat = self.denoise(t, z_t, node_mask, None, edge_mask, context)
        mu = z_t / alpha_ts - (sigma2_ts / alpha_ts / sigma_t) * eps_hat
        sigma = sigma_ts * sigma_s / sigma_t
        return mu + sigma * self.combined_noise(noise_fn, mu.size(0), mu.size(1), node_mask)

    def q_zs(self, s, t, z_t, x, node_mask, noise_fn):
        """edm.py:652-672."""
        g_s, g_t = self.gamma(s), self.gamma(t)
        sigma2_ts, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(g_t, g_s, z_t)
        sigma_s, sigma_t = self.sigma(g_s, z_t), self.sigma(g_t, z_t)
        alpha_s = self.alpha(g_s, x)
        mu = alpha_ts * (sigma_s ** 2) / (sigma_t ** 2) * z_t + alpha_s * sigma2_ts / (sigma_t ** 2) * x
        sigma = sigma_ts * sigma_s / sigma_t
        return mu + sigma * self.combined_noise(noise_fn, mu.size(0), mu.size(1), node_mask)

    @torch.no_grad()
    def sample_chain(self, x, h, node_mask, edge_mask, fragment_mask, linker_mask, context, noise_fn, keep_frames=None):
        """``InpaintingEDM.sample_chain`` edm.py:549-614."""
        n_samples, n_nodes = x.size(0), x.size(1)
        x, h = self.normalize(x, h)
        xh = torch.cat([x, h], dim=2)
        z = self.combined_noise(noise_fn, n_samples, n_nodes, node_mask)
        keep_frames = self.T if keep_frames is None else keep_frames
        assert keep_frames <= self.T
        chain = torch.zeros((keep_frames,) + z.size(), dtype=z.dtype, device=z.device)
        for s in reversed(range(0, self.T)):
            s_arr = torch.full((n_samples, 1), fill_value=s, device=z.device)
            t_arr = ((s_arr + 1) / self.T).to(z.dtype)
            s_arr = (s_arr / self.T).to(z.dtype)
            z_l = self.p_zs(s_arr, t_arr, z, node_mask, edge_mask, context, noise_fn)
            z_f = self.q_zs(s_arr, t_arr, z, xh * fragment_mask, fragment_mask, noise_fn)
            z = z_l * linker_mask + z_f * fragment_mask
            z = torch.cat([remove_mean_with_mask(z[:, :, :self.n_dims], node_mask), z[:, :, self.n_dims:]], dim=2)
            chain[(s * keep_frames) // self.T] = self.unnormalize_z(z)
        # p(x, h | z_0) for the linker, q(x, h | z_0) for the fragments (edm.py:674-713)
        zeros = torch.zeros((n_samples, 1), dtype=z.dtype, device=z.device)
        g_0 = self.gamma(zeros)
        sigma_x = torch.exp(-(-0.5 * g_0)).unsqueeze(1)
        eps_hat = self.denoise(zeros, z, node_mask, None, edge_mask, context)
        mu_x = 1. / self.alpha(g_0, eps_hat) * (z - self.sigma(g_0, eps_hat) * eps_hat)
        xh_l = mu_x + sigma_x * self.combined_noise(noise_fn, n_samples, n_nodes, node_mask)
        x_l, h_l = self.unnormalize(xh_l[:, :, :self.n_dims], xh_l[:, :, self.n_dims:])
        h_l = F.one_hot(torch.argmax(h_l, dim=2), self.in_node_nf) * node_mask
        alpha_0, sigma_0 = self.alpha(g_0, z), self.sigma(g_0, z)
        eps = self.combined_noise(noise_fn, n_samples, n_nodes, node_mask)
        xh_f = (1 / alpha_0) * z - (sigma_0 / alpha_0) * eps
        x_f, h_f = self.unnormalize(xh_f[:, :, :self.n_dims], xh_f[:, :, self.n_dims:])
        h_f = F.one_hot(torch.argmax(h_f, dim=2), self.in_node_nf) * node_mask
        out_l = torch.cat([x_l, h_l.to(x_l.dtype)], dim=2)
        out_f = torch.cat([x_f, h_f.to(x_f.dtype)], dim=2)
        chain[0] = out_l * linker_mask + out_f * fragment_mask
        return chain


class NoiseBank:
    """Replayable noise: ``2*(T+2)`` draws in reference order (x-part then h-part).

    ``draws[k]`` is the UNMASKED standard-normal tensor of draw k; the mask is
    applied on use like ``utils.sample_gaussian_with_mask`` (utils.py:189-192).
    """

    def __init__(self, draws):
        self.draws = draws
        self.pos = 0

    @classmethod
    def generate(cls, n_steps, batch, n_nodes, n_dims, nf, seed, dtype=torch.float32):
        g = torch.Generator().manual_seed(seed)
        draws = []
        for _ in range(n_steps + 2):
            draws.append(torch.randn((batch, n_nodes, n_dims), generator=g, dtype=torch.float32).to(dtype))
            draws.append(torch.randn((batch, n_nodes, nf), generator=g, dtype=torch.float32).to(dtype))
        return cls(draws)

    def reset(self):
        self.pos = 0

    def __call__(self, size, mask):
        d = self.draws[self.pos]
        assert tuple(d.shape) == tuple(size), (d.shape, size, self.pos)
        self.pos += 1
        return d.to(mask.device) * mask

    def stacked(self):
        """(noise_x [T+2,B,N,3], noise_h [T+2,B,N,nf]) — the layout the HIP chain kernel reads."""
        return torch.stack(self.draws[0::2]), torch.stack(self.draws[1::2])


def make_dynamics_oracle(state_dict, cfg, prefix='dynamics'):
    """Bind a state_dict + config into a ``denoise(...)`` callable."""
    fn = egnn_oracle.dynamics_forward if cfg.graph_type == 'FC' else egnn_oracle.dynamics_forward_pockets

    def denoise(t, xh, node_mask, linker_mask, edge_mask, context):
        return fn(state_dict, cfg, t, xh, node_mask, linker_mask, edge_mask, context, pre=prefix)
    return denoise
