"""Build container only (it imports /root/reference, with the import stubs of tests/golden/make_golden.py for RDKit, pytorch_lightning,
...): the HOST side of the hot path against the unmodified reference on random molecules -
  * the PRODUCT's ``difflinker_amd.datasets.collate``, ``collate_with_fragment_edges``, ``collate_with_fragment_without_pocket_edges``
    and ``create_templates_for_linker_generation`` against ``src/datasets.py:332-512``, tensor for tensor, bit for bit;
  * the oracle's ``ddpm_oracle.sample_chain`` (templates, context with / without anchors, the pockets branch, the centre-of-mass
    mask by dataset type and by ``center_of_mass``) against ``DDPM.sample_chain`` (src/lightning.py:405-463) on a CPU reference
    ``DDPM`` with seeded weights and a shared noise bank.
    PYTHONDONTWRITEBYTECODE=1 python scripts/r5/fuzz_glue_vs_reference.py [--cases 200] [--seed 0]"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location('make_golden', os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
mg._stub_reference_dependencies()
from src import datasets as ref_datasets                # noqa: E402
from src import utils as ref_utils                      # noqa: E402
from src.lightning import DDPM as RefDDPM               # noqa: E402
from difflinker_amd import DDPM as OurDDPM             # noqa: E402

from helpers import seeded_state_dict, GLUE_HPARAMS, max_abs    # noqa: E402
from difflinker_amd import datasets as our_datasets     # noqa: E402
from oracle import ddpm_oracle, edm_oracle, egnn_oracle # noqa: E402
from oracle.egnn_oracle import EGNNConfig               # noqa: E402


def molecules(rng, g, pockets, nf):
    mols = []
    for _ in range(int(rng.integers(1, 6))):
        n_frag, n_link = int(rng.integers(2, 16)), int(rng.integers(1, 8))
        n_pocket = int(rng.integers(3, 25)) if pockets else 0
        n = n_frag + n_pocket + n_link
        types = torch.randint(0, nf, (n,), generator=g)
        role = torch.cat([torch.zeros(n_frag), torch.ones(n_pocket), 2 * torch.ones(n_link)])
        anchors = torch.zeros(n)
        for a in rng.choice(n_frag, size=min(2, n_frag), replace=False):
            anchors[int(a)] = 1
        m = {'uuid': len(mols), 'name': f'm{len(mols)}', 'positions': 2.5 * torch.randn((n, 3), generator=g),
             'one_hot': torch.nn.functional.one_hot(types, nf).float(), 'anchors': anchors,
             'fragment_mask': (role < 2).float(), 'linker_mask': (role == 2).float(), 'num_atoms': n}
        if pockets:
            m['fragment_only_mask'] = (role == 0).float()
            m['pocket_mask'] = (role == 1).float()
        mols.append(m)
    return mols


def equal(x, y):
    if torch.is_tensor(x) or torch.is_tensor(y):
        return torch.is_tensor(x) and torch.is_tensor(y) and x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y)
    if isinstance(x, (list, tuple)):
        return isinstance(y, (list, tuple)) and len(x) == len(y) and all(equal(p, q) for p, q in zip(x, y))
    return x == y


def same(a, b, where, bad):
    """dicts of tensors / lists: same keys, same shapes and dtypes, same bits"""
    if set(a) != set(b):
        bad.append(f'{where}: keys {sorted(set(a) ^ set(b))} differ')
        return
    for k in a:
        if not equal(a[k], b[k]):
            bad.append(f'{where}[{k}] differs')


def schedules_and_step_scalars(n, bad):
    """The product's ``PredefinedNoiseSchedule`` (tables and lookups) and the per-step scalars its sampler kernels are given
    (``EDM.step_coefficients``) against the reference's ``src/noise.py`` and the expressions of ``EDM.sample_p_zs_given_zt_only_linker`` /
    ``sample_p_xh_given_z0_only_linker`` (src/edm.py:178-242) evaluated by the reference's own methods - bit for bit."""
    from src.noise import PredefinedNoiseSchedule as RefSchedule
    from src.edm import EDM as RefEDM
    from difflinker_amd.noise import PredefinedNoiseSchedule as OurSchedule
    from difflinker_amd import EDM as OurEDM, Dynamics as OurDynamics
    rng = np.random.default_rng(77)
    dyn = OurDynamics(3, 8, 1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
    for _ in range(n):
        timesteps = int(rng.integers(2, 1501))
        sched = str(rng.choice(['polynomial_1', 'polynomial_2', 'polynomial_3', 'polynomial_2.5', 'cosine']))
        prec = float(rng.choice([1e-5, 1e-4, 1e-3]))
        r, o = RefSchedule(sched, timesteps=timesteps, precision=prec), OurSchedule(sched, timesteps=timesteps, precision=prec)
        t = torch.rand(5, 1)
        if not (torch.equal(r.gamma.data, o.gamma.data) and torch.equal(r(t), o(t))):
            bad.append(f'noise schedule {sched} timesteps={timesteps} precision={prec}')
        if sched == 'cosine':
            continue
        T = int(rng.integers(1, min(timesteps, 40) + 1))
        kw = dict(in_node_nf=8, n_dims=3, timesteps=timesteps, noise_schedule=sched, noise_precision=prec, loss_type='l2', norm_values=[1, 4, 10])
        ours, ref = OurEDM(dyn, **kw), RefEDM(dynamics=torch.nn.Identity(), **kw)
        ours.T = ref.T = T
        coefs, (inv_a0, s0, sx) = ours.step_coefficients()
        z = torch.zeros(1, 1, 1)
        for q, s_ in enumerate(reversed(range(T))):
            s_arr, t_arr = torch.full((1, 1), s_) / T, (torch.full((1, 1), s_) + 1) / T
            g_s, g_t = ref.gamma(s_arr), ref.gamma(t_arr)
            s2, s_ts, a_ts = ref.sigma_and_alpha_t_given_s(g_t, g_s, z)
            want = torch.stack([t_arr.view(()), a_ts.view(()), (s2 / a_ts / ref.sigma(g_t, z)).view(()),
                                (s_ts * ref.sigma(g_s, z) / ref.sigma(g_t, z)).view(())])
            if not torch.equal(coefs[q], want):
                bad.append(f'step scalars {sched} timesteps={timesteps} precision={prec} T={T} step {q}: {coefs[q].tolist()} != {want.tolist()}')
                break
        g0 = ref.gamma(torch.zeros(1, 1))
        if not (abs(sx - float(torch.exp(0.5 * g0))) < 1e-9 and abs(inv_a0 - float(1. / ref.alpha(g0, z))) < 1e-7 and abs(s0 - float(ref.sigma(g0, z))) < 1e-9):
            bad.append(f'decode scalars {sched} timesteps={timesteps} precision={prec}')
    return n


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    t0, bad, chains, worst, glue = time.time(), [], 0, 0.0, 0
    n_sched = schedules_and_step_scalars(200, bad)
    print(f'{n_sched} noise schedules (polynomial_1 / 2 / 2.5 / 3, cosine; 2..1500 steps; precisions) and the per-step scalars of chains of 1..40 steps: '
          f'{len(bad)} differences from the reference', flush=True)
    for k in range(a.cases):
        seed = a.seed * 100000 + k
        rng = np.random.default_rng(seed)
        g = torch.Generator().manual_seed(seed)
        pockets = bool(rng.random() < 0.4)
        nf = int(rng.choice([8, 9, 10]))
        mols = molecules(rng, g, pockets, nf)
        tag = f'seed {seed} ({"pockets" if pockets else "fc"}, nf {nf}, atoms {[m["num_atoms"] for m in mols]})'
        n0 = len(bad)
        ref = ref_datasets.collate([dict(m) for m in mols])
        same(our_datasets.collate([dict(m) for m in mols]), ref, f'collate {tag}', bad)
        same(our_datasets.collate_with_fragment_edges([dict(m) for m in mols]), ref_datasets.collate_with_fragment_edges([dict(m) for m in mols]),
             f'collate_with_fragment_edges {tag}', bad)
        if pockets:
            same(our_datasets.collate_with_fragment_without_pocket_edges([dict(m) for m in mols]),
                 ref_datasets.collate_with_fragment_without_pocket_edges([dict(m) for m in mols]), f'collate_with_fragment_without_pocket_edges {tag}', bad)
        sizes = torch.tensor([int(rng.integers(1, 9)) for _ in mols])
        t_ref = ref_datasets.create_templates_for_linker_generation(ref, sizes)
        same(our_datasets.create_templates_for_linker_generation(our_datasets.collate([dict(m) for m in mols]), sizes), t_ref, f'templates {tag}', bad)
        t_orc = ddpm_oracle.create_templates(ddpm_oracle.collate([dict(m) for m in mols]), sizes.tolist())
        same(dict(t_orc, num_atoms=[int(v) for v in t_orc['num_atoms']]), dict(t_ref, num_atoms=[int(v) for v in t_ref['num_atoms']]),
             f'oracle templates {tag}', bad)              # (the reference's num_atoms: 0-dim tensors when linker_sizes is a tensor)
        anchors_context = bool(rng.random() < 0.5)
        com = str(rng.choice(['fragments', 'anchors']))
        ctx = (2 if pockets else 1) + int(anchors_context)
        L, T = 1, int(rng.integers(1, 4))
        hp = dict(GLUE_HPARAMS, in_node_nf=nf, context_node_nf=ctx, n_layers=L, anchors_context=anchors_context, center_of_mass=com)
        if pockets:                                     # (what makes the reference's DDPM take its pockets branch: lightning.py:431)
            hp.update(train_data_prefix='MOAD_train.full', val_data_prefix='MOAD_val.full', graph_type='FC-10A-4A')
        # the PRODUCT's DDPM.sample_chain up to the sampler call, against the reference's: what each hands to edm.sample_chain
        # (templates or - inpainting - the molecules as given, the context, the coordinates with the chosen centre removed)
        handed = []
        for cls, dsets in ((OurDDPM, our_datasets), (RefDDPM, ref_datasets)):
            m = cls(**dict(hp, inpainting=bool(rng.random() < 0.2) if cls is OurDDPM else handed[0][0])).eval()
            if cls is OurDDPM:
                handed.append((m.inpainting,))
            if pockets:
                m.val_dataset = dsets.MOADDataset(data=mols)
            got_kw = {}

            def record(**kw):
                got_kw.update(kw)
                return torch.zeros(1)
            m.edm.sample_chain = record
            _, nmask = m.sample_chain(dsets.collate([dict(x_) for x_ in mols]), sample_fn=lambda d: sizes, keep_frames=1)
            assert {'x', 'h', 'node_mask', 'edge_mask', 'fragment_mask', 'linker_mask', 'context', 'keep_frames'} <= set(got_kw)
            handed.append(dict(got_kw, returned_node_mask=nmask))
        same(handed[1], handed[2], f'DDPM.sample_chain -> edm.sample_chain arguments (inpainting={handed[0][0]}, anchors_context={anchors_context}, '
             f'center_of_mass={com}) {tag}', bad)
        glue += 1
        if k % 4 == 0:                                  # the sampler glue end to end: a reference DDPM on the CPU (its denoiser loops in Python)
            ddpm = RefDDPM(**hp)
            sd = seeded_state_dict(nf + ctx + 1, 128, L, seed, coord_gain=0.02)
            ddpm.edm.dynamics.load_state_dict(sd, strict=True)
            ddpm.eval()
            ddpm.edm.T = T
            if pockets:
                ddpm.val_dataset = ref_datasets.MOADDataset(data=mols)
            B, N = t_ref['positions'].shape[:2]
            bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=seed + 3)
            pos = [0]

            def banked(size, device, node_mask):
                d = bank.draws[pos[0]]
                assert tuple(d.shape) == tuple(size)
                pos[0] += 1
                return d * node_mask
            orig = ref_utils.sample_gaussian_with_mask
            ref_utils.sample_gaussian_with_mask = banked
            try:
                with torch.no_grad():
                    want, want_mask = ddpm.sample_chain(ref, sample_fn=lambda d: sizes, keep_frames=min(2, T))
            finally:
                ref_utils.sample_gaussian_with_mask = orig
            cfg = EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=L, graph_type='FC-10A-4A' if pockets else 'FC')
            orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=hp['diffusion_steps'],
                                       noise_schedule=hp['diffusion_noise_schedule'], noise_precision=hp['diffusion_noise_precision'],
                                       norm_values=hp['normalize_factors'])
            orc.T = T
            bank.reset()
            got, got_mask, _ = ddpm_oracle.sample_chain(orc, ddpm_oracle.collate([dict(m) for m in mols]), sizes.tolist(), bank, min(2, T),
                                                        anchors_context, pockets, pockets, center_of_mass=com)
            chains += 1
            e = max_abs(got, want)
            worst = max(worst, e)
            if e > 0.0 or not torch.equal(got_mask, want_mask):
                bad.append(f'DDPM.sample_chain {tag} anchors_context={anchors_context} center_of_mass={com}: max-abs {e:.2e}')
        print(('ok  ' if len(bad) == n0 else 'FAIL'), tag, *(bad[n0:]), flush=True)
    print(f'{a.cases} random batches in {time.time() - t0:.0f} s: collate / collate_with_fragment_edges / templates of the product and the oracle equal the '
          f"reference's bit for bit, and so is what the product's DDPM.sample_chain hands to its sampler ({glue} runs); {chains} DDPM.sample_chain runs of the "
          f'oracle end to end, largest difference {worst:.1e}; {len(bad)} failures')
    for b in bad:
        print('FAILED:', b)
